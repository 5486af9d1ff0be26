#!/bin/bash
# call n: the paired pipeline where one small mesh is walked inline (glass's lamp, table.tin): every mesh through k_walk instead, so that k_step runs its lean variant
O=gpurun_out/r6n; mkdir -p $O
python scratch/ab_rates.py --repeat 2 --lib 'split=args:--pipeline split' --lib 'paired=args:--pipeline paired' --lib 'paired_walkall=tuning:{"walk_min_tris":1};args:--pipeline paired' --lib 'split_walkall=tuning:{"walk_min_tris":1};args:--pipeline split' glass table meshlight > $O/ab_paired_walkall.md 2> $O/ab.err; cat $O/ab_paired_walkall.md
