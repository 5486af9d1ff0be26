#!/bin/bash
# round 5, call x: the path state's arrays carved out of one allocation, array k starting k x SKEW bytes past a 2-MiB boundary: which skews run
# the streaming kernels at their fast speed?  (call w: skew 0 = every array on a 2-MiB boundary is always the SLOW speed)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5x; mkdir -p $O
V="TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/libtinsel_hip_slab.so TINSEL_HIP_STATE_SLAB_GB=40"
{
echo "| environment | workload | Msamples/s | kernel ms |"; echo "|---|---|---|---|"
W="--scene glass --width 1920 --height 1080 --maxdepth 12 --steps 20 --warmup 3"
for K in 256 1024 4352 8448 69888 1118464 33024 2359552; do
  bash scratch/gpu_envs.sh "$V TINSEL_HIP_STATE_SLAB_SKEW=$K" "$V TINSEL_HIP_STATE_SLAB_SKEW=$K" -- $W
done
} > $O/ab_slab_skew.md 2>&1; sed "s|TINSEL_HIP_LIB=[^ ]*libtinsel_hip_slab.so TINSEL_HIP_STATE_SLAB_GB=40 TINSEL_HIP_STATE_SLAB_||" $O/ab_slab_skew.md
