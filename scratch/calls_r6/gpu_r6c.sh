#!/bin/bash
# call c: the path state by 64-position blocks (tn_layout.h) against the arrays of rounds 2-5 (scratch/ab/libtinsel_hip_arrays.so, -DTN_STATE_BLOCKS=0):
# the GPU suite on the blocks build, then rates over fresh processes (the spread is the point)
mkdir -p gpurun_out/r6c
python -m pytest tests -m gpu -q -x > gpurun_out/r6c/pytest_gpu.log 2>&1; tail -4 gpurun_out/r6c/pytest_gpu.log
B=tinsel_amd/libtinsel_hip.so; A=scratch/ab/libtinsel_hip_arrays.so
python scratch/ab_rates.py --lib blocks=$B --lib arrays=$A --repeat 5 glass cfg3 > gpurun_out/r6c/ab_blocks_split.md 2> gpurun_out/r6c/ab.err
python scratch/ab_rates.py --lib blocks=$B --lib arrays=$A --repeat 2 aphrodite many_spheres motionblur cornell veach4k cfg1 > gpurun_out/r6c/ab_blocks_other.md 2>> gpurun_out/r6c/ab.err
cat gpurun_out/r6c/ab_blocks_split.md gpurun_out/r6c/ab_blocks_other.md
