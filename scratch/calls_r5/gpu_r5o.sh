#!/bin/bash
# round 5, call o: k_accumulate_tiled's halo tiles (a shard's accumulate tiles that only TOUCH an owned shard tile) -- pixels with no candidate
# of this shard left alone, the tile dealt by rows or columns so the strip is one wave's, heaviest tiles first -- scratch build "halo" against
# the in-tree library: the shard tests, then one shard of 8 alone on the device
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5o; mkdir -p $O
V=TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/libtinsel_hip_halo.so
( time env $V timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_group.py tests/test_gpu_multirank.py -m gpu -q -k "shard or group or rank or filter or golden" 2>&1 | tail -12 ) > $O/pytest_halo.log 2>&1; grep -a "passed\|failed\|rror" $O/pytest_halo.log | tail -8
P=TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/libtinsel_hip_prehalo.so
for L in "" "$V"; do
{
echo "library: ${L:-in tree}"; echo
echo "| workload | numbering | 20 steps | paths/s vs one shard | kernel ms (20 steps) |"; echo "|---|---|---|---|---|"
env $L timeout 300 python scratch/shard_emul.py cornell 1024 1024 8
env $L timeout 300 python scratch/shard_emul.py veach 3840 2160 8
env $L timeout 300 python scratch/shard_emul.py large/ajax_standin 1920 1080 8 4
echo
} 2>&1 | grep -v amdgpu.ids
done > $O/shard_halo.md; cat $O/shard_halo.md | cut -c1-300
