#!/bin/bash
# round 5, call y: the slab again with COARSE skews between the arrays (8 MiB ... 512 MiB per array): the speed of the streaming kernels changes
# with where the path state's arrays land (scratch/realloc_modes.py: 7.98 ... 9.88 ms for k_shade within ONE process), small skews do not
# matter (call x) -- do large ones?  Then the addresses of the library's own allocations, eight states in one process, beside their speeds
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5y; mkdir -p $O
V="TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/libtinsel_hip_slab.so TINSEL_HIP_STATE_SLAB_GB=60"
{
echo "| environment | workload | Msamples/s | kernel ms |"; echo "|---|---|---|---|"
W="--scene glass --width 1920 --height 1080 --maxdepth 12 --steps 20 --warmup 3"
for K in 8392704 33554432 100663296 167772160 270532608 536870912 1073741824; do
  bash scratch/gpu_envs.sh "$V TINSEL_HIP_STATE_SLAB_SKEW=$K" "$V TINSEL_HIP_STATE_SLAB_SKEW=$K" -- $W
done
} > $O/ab_slab_coarse.md 2>&1; sed "s|TINSEL_HIP_LIB=[^ ]*libtinsel_hip_slab.so TINSEL_HIP_STATE_SLAB_GB=60 TINSEL_HIP_STATE_SLAB_||" $O/ab_slab_coarse.md
TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/libtinsel_hip_slab.so TINSEL_HIP_STATE_SLAB_PRINT=1 timeout 250 python scratch/realloc_modes.py > $O/realloc_addresses.txt 2>&1; grep -c array $O/realloc_addresses.txt
