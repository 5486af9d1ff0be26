#!/bin/bash
# call 2t: the tolerance arm after this round's changes to the scan and the light loops (bench line z3: fast / exact 0.98 on cornell, 1.25 before)
O=gpurun_out/r6_2t; mkdir -p $O
timeout 2400 python scratch/ab_rates.py --repeat 3 --lib 'exact=tinsel_amd/libtinsel_hip.so' --lib 'fast=tinsel_amd/libtinsel_hip.so;args:--arith fast' --lib 'fast_norecs=scratch/ab/libtinsel_hip_norecs.so;args:--arith fast' --lib 'fast_noquadrec=scratch/ab/libtinsel_hip_noquadrec.so;args:--arith fast' --lib 'fast_base=scratch/ab/libtinsel_hip_base.so;args:--arith fast' --lib 'exact_norecs=scratch/ab/libtinsel_hip_norecs.so' \
    cornell veach4k > $O/ab_fast_arm.md 2> $O/ab.err; cat $O/ab_fast_arm.md
