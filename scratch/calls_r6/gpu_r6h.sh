#!/bin/bash
# call h: the PAIRED pipeline against the split one, fresh processes
O=gpurun_out/r6h; mkdir -p $O
python scratch/ab_rates.py --repeat 3 --lib 'split=args:--pipeline split' --lib 'paired=args:--pipeline paired' glass cfg3 aphrodite table transmission motionblur meshlight > $O/ab_paired.md 2> $O/ab.err; cat $O/ab_paired.md
python scratch/ab_rates.py --repeat 1 --lib 'split=args:--pipeline split' --lib 'paired=args:--pipeline paired' --lib 'auto=args:--pipeline auto' cornell veach4k > $O/ab_paired_fused_scenes.md 2>> $O/ab.err; cat $O/ab_paired_fused_scenes.md
