#!/bin/bash
# call 2r: k_walk2 as two 768-thread workgroups per CU (six waves per SIMD, twelve rays per lane slot)
O=gpurun_out/r6_2r; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_walk.py -x -q -k "two_rays and ajax" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
timeout 1800 python scratch/ab_rates.py --repeat 3 --lib 'one=args:--pipeline auto' --lib 'two6=tuning:{"walk_two_rays":2}' --lib 'two6_r32=tuning:{"walk_two_rays":2,"walk_refill_min":32}' --lib 'two6_s6=tuning:{"walk_two_rays":2,"walk_lds_stack":6}' --lib 'two6_r48=tuning:{"walk_two_rays":2,"walk_refill_min":48}' \
    cfg3 aphrodite > $O/ab_two_rays6.md 2> $O/ab.err; cat $O/ab_two_rays6.md
