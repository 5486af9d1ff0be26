#!/bin/bash
# call l: AUTO now picks the paired pipeline where it was measured to win; the whole GPU suite; auto against split against k_step at five waves
O=gpurun_out/r6l; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
python scratch/ab_rates.py --repeat 2 --lib 'split=args:--pipeline split' --lib 'auto=args:--pipeline auto' --lib 'paired=args:--pipeline paired' --lib 'paired5=scratch/ab/libtinsel_hip_step5.so' cfg3 aphrodite transmission glass table motionblur meshlight > $O/ab_paired_auto.md 2> $O/ab.err; cat $O/ab_paired_auto.md
