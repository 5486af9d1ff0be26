#!/bin/bash
# call 2o: overlapped chunks with the walk kernels at full size: bit-identical? the slow processes of call 2n?
O=gpurun_out/r6_2o; mkdir -p $O
for s in large/transmission large/ajax_standin glass; do for p in auto split; do timeout 300 python scratch/overlap_check.py $s $p 8 2>&1 | grep -v amdgpu.ids; done; done > $O/overlap_check.txt; cat $O/overlap_check.txt
