#!/bin/bash
# call 2v: the accumulate kernel's other two forms at cornell 1024^2 (the tile count picks TILED there; WIDE and PIPED were built for few tiles per CU)
O=gpurun_out/r6_2v; mkdir -p $O
timeout 1500 python scratch/ab_rates.py --repeat 3 --lib 'tiled=args:--pipeline auto' --lib 'wide=tuning:{"accumulate":2}' --lib 'piped=tuning:{"accumulate":3}' cornell veach4k > $O/ab_accumulate.md 2> $O/ab.err; cat $O/ab_accumulate.md
