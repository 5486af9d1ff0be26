#!/bin/bash
# call 2y: the path kernels as 128-thread workgroups (two waves: finer dispatch, less waiting for a workgroup's slowest wave) instead of 256
O=gpurun_out/r6_2y; mkdir -p $O
TINSEL_HIP_LIB=scratch/ab/libtinsel_hip_b128.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q > $O/pytest_b128.log 2>&1; tail -3 $O/pytest_b128.log
timeout 1800 python scratch/ab_rates.py --repeat 3 --lib b256=tinsel_amd/libtinsel_hip.so --lib b128=scratch/ab/libtinsel_hip_b128.so cornell veach4k cfg1 glass cfg3 > $O/ab_block128.md 2> $O/ab.err; cat $O/ab_block128.md
