#!/bin/bash
# round 5, call z: placements of the path state sampled at allocation (TINSEL_HIP_PLACEMENTS=N: N allocations of the whole set, each timed with
# k_probe_state while the best so far is held, the fastest kept) -- twelve processes in a row on glass, then the 524k-triangle config
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5z; mkdir -p $O
V="TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/libtinsel_hip_place.so"
{
echo "| environment | workload | Msamples/s | kernel ms |"; echo "|---|---|---|---|"
W="--scene glass --width 1920 --height 1080 --maxdepth 12 --steps 20 --warmup 3"
bash scratch/gpu_envs.sh "$V" "$V TINSEL_HIP_PLACEMENTS=4" "$V" "$V TINSEL_HIP_PLACEMENTS=4" "$V" "$V TINSEL_HIP_PLACEMENTS=4" "$V TINSEL_HIP_PLACEMENTS=4" "$V" "$V TINSEL_HIP_PLACEMENTS=6" "$V TINSEL_HIP_PLACEMENTS=6" "$V" "$V TINSEL_HIP_PLACEMENTS=6" -- $W
W="--scene large/ajax_standin --width 1920 --height 1080 --maxdepth 4 --steps 20 --warmup 3"
bash scratch/gpu_envs.sh "$V" "$V TINSEL_HIP_PLACEMENTS=4" "$V" "$V TINSEL_HIP_PLACEMENTS=4" "$V" "$V TINSEL_HIP_PLACEMENTS=4" -- $W
} > $O/ab_placements.md 2>&1; sed "s|TINSEL_HIP_LIB=[^ ]*libtinsel_hip_place.so|placement build|" $O/ab_placements.md
