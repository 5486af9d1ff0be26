#!/bin/bash
# call 2f: of the light table, only the cursor and nee_sum through the scalar path (the light's primitive record stays a per-lane LDS read)
O=gpurun_out/r6_2f; mkdir -p $O
timeout 2400 python scratch/ab_rates.py --repeat 3 --lib nolt=scratch/ab/libtinsel_hip_nolt.so --lib cursor=tinsel_amd/libtinsel_hip.so --lib all=scratch/ab/libtinsel_hip_all.so \
    cornell veach4k cfg1 > $O/ab_cursor.md 2> $O/ab.err; cat $O/ab_cursor.md
