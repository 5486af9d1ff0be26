#!/bin/bash
# call 2m: the compiler's other knobs on the final tree: -O2, and the scheduling strategies round 5 did not try (iterative-minreg / -maxocc / -ilp)
O=gpurun_out/r6_2m; mkdir -p $O
timeout 3000 python scratch/ab_rates.py --repeat 3 --lib now=tinsel_amd/libtinsel_hip.so --lib o2=scratch/ab/libtinsel_hip_o2.so --lib minreg=scratch/ab/libtinsel_hip_minreg.so --lib maxocc=scratch/ab/libtinsel_hip_maxocc.so --lib iterilp=scratch/ab/libtinsel_hip_iterilp.so \
    cornell veach4k glass cfg3 > $O/ab_flags.md 2> $O/ab.err; cat $O/ab_flags.md
