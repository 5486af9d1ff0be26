#!/bin/bash
# call 2w: the lean k_extend (with light sampling) held to six / seven waves per SIMD by its launch bounds (it takes 83-89 VGPRs at the quad level: five waves)
O=gpurun_out/r6_2w; mkdir -p $O
timeout 1500 python scratch/ab_rates.py --repeat 3 --lib now=tinsel_amd/libtinsel_hip.so --lib ext6=scratch/ab/libtinsel_hip_ext6.so --lib ext7=scratch/ab/libtinsel_hip_ext7.so glass 'cfg3' > $O/ab_extend_waves.md 2> $O/ab.err; cat $O/ab_extend_waves.md
