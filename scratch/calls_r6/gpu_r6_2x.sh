#!/bin/bash
# call 2x: the fused kernel's launch geometry re-swept after the trace got cheaper (grid multiplier, tail split, shading pools, region sharing) on cornell / veach 4K
O=gpurun_out/r6_2x; mkdir -p $O
timeout 2400 python scratch/ab_rates.py --repeat 2 --lib 'default=args:--pipeline auto' --lib 'grid16=tuning:{"grid_mult":16}' --lib 'grid24=tuning:{"grid_mult":24}' --lib 'grid48=tuning:{"grid_mult":48}' --lib 'grid64=tuning:{"grid_mult":64}' \
  --lib 'notail=tuning:{"tail_split":0}' --lib 'tail_q=tuning:{"tail_split":1,"tail_share":0.25,"tail_divide":4}' --lib 'tail_8=tuning:{"tail_split":1,"tail_share":0.5,"tail_divide":8}' --lib 'repack=tuning:{"repack":1}' --lib 'share=tuning:{"bounce_share":1}' --lib 'noshare=tuning:{"bounce_share":0}' \
  cornell veach4k > $O/ab_geometry.md 2> $O/ab.err; cat $O/ab_geometry.md
