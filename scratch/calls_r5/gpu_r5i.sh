#!/bin/bash
# round 5, call i: the whole library under the compiler's other instruction-scheduling strategies (-mllvm -amdgpu-sched-strategy=max-ilp /
# max-memory-clause; scheduling only: results cannot change) against the default -- k_walk's 4-8 % swing with the scoping of its source
# (call g) says the schedule matters
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5i; mkdir -p $O
I=TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/libtinsel_hip_ilp.so
M=TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/libtinsel_hip_memc.so
{
echo "| library | workload | Msamples/s | kernel ms |"; echo "|---|---|---|---|"
for W in "--scene cornell --width 1024 --height 1024 --steps 20 --warmup 5" "--scene veach --width 3840 --height 2160 --steps 20 --warmup 5" \
         "--scene large/ajax_standin --width 1920 --height 1080 --maxdepth 4 --steps 20 --warmup 5" "--scene glass --width 1920 --height 1080 --maxdepth 12 --steps 20 --warmup 5" \
         "--scene cornell --width 256 --height 256 --steps 16 --warmup 4" "--scene large/table --width 1024 --height 1024 --steps 20 --warmup 5"; do
  bash scratch/gpu_envs.sh "-" "$I" "$M" "-" "$I" "$M" -- $W
done
} > $O/ab_sched_strategy.md 2>&1; cat $O/ab_sched_strategy.md
