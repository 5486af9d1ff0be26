#!/bin/bash
# call m: the paired pipeline's pending records packed (sample 0's cosine in the throughput record, the light index recomputed); its grid and tail
O=gpurun_out/r6m; mkdir -p $O
python -m pytest tests/test_gpu_paired.py -q > $O/pytest_paired.log 2>&1; tail -2 $O/pytest_paired.log
python scratch/ab_rates.py --repeat 2 --lib 'auto=tuning:{}' --lib 'g16=tuning:{"grid_mult":16}' --lib 'g64=tuning:{"grid_mult":64}' --lib 'tail=tuning:{"tail_split":1,"tail_share":0.125,"tail_divide":4}' cfg3 aphrodite transmission > $O/ab_paired_grid.md 2> $O/ab.err; cat $O/ab_paired_grid.md
