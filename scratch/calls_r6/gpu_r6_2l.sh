#!/bin/bash
# call 2l: the next round's records touched into the L2 (global_load_lds_dword into an LDS sink) by the streaming kernels k_shade and k_step
O=gpurun_out/r6_2l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_paired.py tests/test_gpu_walk.py tests/test_gpu_split.py -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 2400 python scratch/ab_rates.py --repeat 3 --lib notouch=scratch/ab/libtinsel_hip_notouch.so --lib touch=tinsel_amd/libtinsel_hip.so \
    glass cfg3 aphrodite transmission many_spheres motionblur > $O/ab_touch.md 2> $O/ab.err; cat $O/ab_touch.md
