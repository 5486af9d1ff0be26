#!/bin/bash
# round 5, call n: what a rank's tile numbering costs -- one shard of 8 alone on the device (8 x the passes over its 1/8 of the pixels)
# against the one-shard frame, per tile edge
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5n; mkdir -p $O
{
echo "| workload | numbering | 20 steps | paths/s vs one shard | kernel ms (20 steps) |"; echo "|---|---|---|---|---|"
timeout 300 python scratch/shard_emul.py cornell 1024 1024 8
timeout 300 python scratch/shard_emul.py veach 3840 2160 8
timeout 300 python scratch/shard_emul.py large/ajax_standin 1920 1080 8 4
} > $O/shard_tile.md 2>&1; cat $O/shard_tile.md
