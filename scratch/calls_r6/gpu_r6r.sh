#!/bin/bash
# call r: k_step's region groups longest first (-DTN_STEP_ORDER=1)
O=gpurun_out/r6r; mkdir -p $O
timeout 600 python scratch/ab_rates.py --repeat 3 --lib base=tinsel_amd/libtinsel_hip.so --lib order=scratch/ab/libtinsel_hip_steporder.so cfg3 aphrodite transmission > $O/ab_step_order.md 2> $O/ab.err; cat $O/ab_step_order.md
