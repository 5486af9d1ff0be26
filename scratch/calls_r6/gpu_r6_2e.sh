#!/bin/bash
# call 2e: the light loops through the scalar path (light table behind the primitive records, the light's record in SGPRs, a quad light's arrays
# through the offsets in its record), against the arm without (nolt) and the arm without the quad offsets (noqr); base = this morning's library
O=gpurun_out/r6_2e; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_paired.py tests/test_gpu_refit.py tests/test_gpu_leaf.py tests/test_gpu_configs.py tests/test_gpu_walk.py tests/test_gpu_probe.py tests/test_fuzz.py -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 2400 python scratch/ab_rates.py --repeat 3 --lib base=scratch/ab/libtinsel_hip_base.so --lib ahead=scratch/ab/libtinsel_hip_ahead.so --lib noqr=scratch/ab/libtinsel_hip_noqr.so --lib nolt=scratch/ab/libtinsel_hip_nolt.so --lib all=tinsel_amd/libtinsel_hip.so \
    cornell veach4k cfg1 glass > $O/ab_lights.md 2> $O/ab.err; cat $O/ab_lights.md
