#!/bin/bash
# round 5, call v: the two speeds of the streaming kernels -- a property of the allocation (physical placement) or of the process?
# six consecutive processes, each streaming over ten 2-GiB allocations in turn
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5v; mkdir -p $O
for i in 1 2 3 4 5 6; do timeout 120 python scratch/stream_regions.py 10 2 2>&1 | grep -v amdgpu.ids; done > $O/stream_regions.txt; cat $O/stream_regions.txt
