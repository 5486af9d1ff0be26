#!/bin/bash
# call 2q: k_walk2 -- two rays per lane (the walk is bound by the texture addresser's 16 cycles per wave instruction: fuller instructions)
O=gpurun_out/r6_2q; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_walk.py -x -q -k two_rays > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 1800 python scratch/ab_rates.py --repeat 3 --lib 'one=args:--pipeline auto' --lib 'two=tuning:{"walk_two_rays":1}' --lib 'two_r32=tuning:{"walk_two_rays":1,"walk_refill_min":32}' --lib 'two_r16_l16=tuning:{"walk_two_rays":1,"walk_refill_min":16,"walk_leaf_min":16}' --lib 'two_l24=tuning:{"walk_two_rays":1,"walk_leaf_min":24}' \
    cfg3 aphrodite > $O/ab_two_rays.md 2> $O/ab.err; cat $O/ab_two_rays.md
