#!/bin/bash
# call 2a: quads (two-triangle meshes) tested in the scan by the lean kernels' second level (SceneT<.., 2, ..>): glass.tin through the paired
# pipeline with k_step<1,2,1>; the split pipeline's k_extend / k_shadow at that level; and the path buffers sized to the Infinity Cache
O=gpurun_out/r6_2a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_paired.py tests/test_gpu_walk.py tests/test_gpu_switches.py -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 1200 python scratch/ab_rates.py --repeat 3 --lib 'auto=args:--pipeline auto' --lib 'general=tuning:{"quads_in_scan":0}' --lib 'split2=args:--pipeline split' \
    glass table transmission motionblur > $O/ab_quads.md 2> $O/ab.err; cat $O/ab_quads.md
timeout 900 python scratch/ab_rates.py --repeat 2 --lib 'b8m=args:--pipeline auto' --lib 'b512k=tuning:{"batch_paths":524288}' --lib 'b1m=tuning:{"batch_paths":1048576}' --lib 'b2m=tuning:{"batch_paths":2097152}' --lib 'b4m=tuning:{"batch_paths":4194304}' \
    glass cfg3 > $O/ab_batch_paths.md 2>> $O/ab.err; cat $O/ab_batch_paths.md
