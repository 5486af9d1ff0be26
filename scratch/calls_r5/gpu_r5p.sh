#!/bin/bash
# round 5, call p: is a shard's accumulate launch as long as its chain of radiance loads?  Scratch build whose pass loop reads the radiance of
# the first two passes over and over (cache hits; wrong image, timing only) against the in-tree library
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5p; mkdir -p $O
V=TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/libtinsel_hip_halo.so
for L in "" "$V"; do
{
echo "library: ${L:-in tree}"; echo
echo "| workload | numbering | 20 steps | paths/s vs one shard | kernel ms (20 steps) |"; echo "|---|---|---|---|---|"
env $L timeout 300 python scratch/shard_emul.py cornell 1024 1024 8
echo
} 2>&1 | grep -v amdgpu.ids
done > $O/acc_samepass.md; cat $O/acc_samepass.md | cut -c1-300
