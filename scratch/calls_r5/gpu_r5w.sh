#!/bin/bash
# round 5, call w: the streaming kernels' two speeds alternate between consecutive processes -- do they when every array of the path state is
# carved out of ONE 40-GiB allocation instead of 25 allocations of their own?  Ten processes in a row, the slab on in every other pair
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5w; mkdir -p $O
V=TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/libtinsel_hip_slab.so
{
echo "| environment | workload | Msamples/s | kernel ms |"; echo "|---|---|---|---|"
W="--scene glass --width 1920 --height 1080 --maxdepth 12 --steps 20 --warmup 3"
bash scratch/gpu_envs.sh "$V" "$V" "$V TINSEL_HIP_STATE_SLAB_GB=40" "$V TINSEL_HIP_STATE_SLAB_GB=40" "$V" "$V" "$V TINSEL_HIP_STATE_SLAB_GB=40" "$V TINSEL_HIP_STATE_SLAB_GB=40" "$V TINSEL_HIP_STATE_SLAB_GB=40" "$V" -- $W
} > $O/ab_state_slab.md 2>&1; sed "s|TINSEL_HIP_LIB=[^ ]*libtinsel_hip_slab.so|slab build|" $O/ab_state_slab.md
