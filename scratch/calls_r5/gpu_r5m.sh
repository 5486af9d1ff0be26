#!/bin/bash
# round 5, call m: SCATTERED regions at one shard (scratch build: the 64-pixel chunks of a pass transposed as a 16 x C matrix, so the 16 waves
# of a 1024-slot region come from 16 evenly spaced places of the frame while a wave keeps its 64 x 1 row piece) -- the opposite direction of
# call l, where compact regions lost 2-10 %
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5m; mkdir -p $O
V=TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/libtinsel_hip_scatter.so
( time env $V TINSEL_HIP_SCATTER=1 timeout 900 python -m pytest tests/test_gpu_reference_scenes.py tests/test_gpu_split.py tests/test_gpu_walk.py tests/test_gpu_configs.py -m gpu -q 2>&1 | tail -12 ) > $O/pytest_scatter.log 2>&1; grep -a "passed\|failed\|rror" $O/pytest_scatter.log | tail -8
{
echo "| environment | workload | Msamples/s | kernel ms |"; echo "|---|---|---|---|"
for W in "--scene large/env_loft --width 1920 --height 1080 --steps 20 --warmup 5" "--scene cornell --width 1024 --height 1024 --steps 20 --warmup 5" \
         "--scene veach --width 3840 --height 2160 --steps 20 --warmup 5" "--scene large/ajax_standin --width 1920 --height 1080 --maxdepth 4 --steps 20 --warmup 5" \
         "--scene glass --width 1920 --height 1080 --maxdepth 12 --steps 20 --warmup 5" "--scene gloss --width 1024 --height 1024 --steps 20 --warmup 5"; do
  bash scratch/gpu_envs.sh "-" "$V" "$V TINSEL_HIP_SCATTER=1" "-" "$V TINSEL_HIP_SCATTER=1" -- $W
done
} > $O/ab_scatter.md 2>&1; sed "s|$GRAFT_REPO_ROOT/scratch/ab/||" $O/ab_scatter.md
