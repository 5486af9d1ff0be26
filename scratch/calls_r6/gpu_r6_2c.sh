#!/bin/bash
# call 2c: the flat scan's loop over the primitives that are NOT in the plane table (scanMask), and a visited primitive's record requested with its box
O=gpurun_out/r6_2c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_paired.py tests/test_gpu_refit.py tests/test_gpu_display.py -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 1800 python scratch/ab_rates.py --repeat 3 --lib base=scratch/ab/libtinsel_hip_base.so --lib mask=scratch/ab/libtinsel_hip_mask.so --lib ahead=tinsel_amd/libtinsel_hip.so \
    cornell veach4k cfg1 glass cfg3 motionblur > $O/ab_scan.md 2> $O/ab.err; cat $O/ab_scan.md
