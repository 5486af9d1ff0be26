#!/bin/bash
# call 2n: the paired pipeline with a batch's passes as two overlapped chunks on two streams (k_walk is latency-bound, k_step bandwidth-bound: do they share a chip?)
O=gpurun_out/r6_2n; mkdir -p $O
timeout 2400 python scratch/ab_rates.py --repeat 3 --lib 'one=args:--pipeline auto' --lib 'overlap=tuning:{"overlap":1}' --lib 'overlap_w256=tuning:{"overlap":1,"walk_block":256}' --lib 'w256=tuning:{"walk_block":256}' \
    cfg3 aphrodite transmission glass > $O/ab_overlap.md 2> $O/ab.err; cat $O/ab_overlap.md
