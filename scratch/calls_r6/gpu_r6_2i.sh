#!/bin/bash
# call 2i: one 16-B record per light for the light loops (LDS), and the next round's path state requested into a corner of LDS while this round shades
O=gpurun_out/r6_2i; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_paired.py tests/test_gpu_configs.py tests/test_gpu_walk.py tests/test_gpu_probe.py tests/test_fuzz.py tests/test_gpu_distributed.py -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 2400 python scratch/ab_rates.py --repeat 3 --lib norecs=scratch/ab/libtinsel_hip_norecs.so --lib recs=scratch/ab/libtinsel_hip_nopf.so --lib prefetch=tinsel_amd/libtinsel_hip.so \
    cornell veach4k cfg1 glass > $O/ab_prefetch.md 2> $O/ab.err; cat $O/ab_prefetch.md
