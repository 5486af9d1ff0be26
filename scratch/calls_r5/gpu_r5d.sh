#!/bin/bash
# round 5, call d: this tree's kernels against call a's library (the split pipeline's kernels after the scalar-register pass: slower, or the box?);
# the always-hit planes four at a time in the fused kernel too (-DTN_PLANE_TABLE_ALL=1: lost 1-3 % at three waves in round 4); the tail
# split at four waves; parity of the tree after the two k_walk modes were removed
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5d; mkdir -p $O
( time timeout 600 python -m pytest tests/test_gpu_walk.py tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_gpu_refit.py tests/test_gpu_lbvh.py tests/test_gpu_rebuild.py -m gpu -q -x 2>&1 | tail -8 ) > $O/pytest_subset.log 2>&1; grep -a "passed\|failed\|rror" $O/pytest_subset.log | tail -5
A=TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/libtinsel_hip_r5a.so
P=TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/libtinsel_hip_pt.so
{
echo "| library | workload | Msamples/s | kernel ms |"; echo "|---|---|---|---|"
for W in "--scene glass --width 1920 --height 1080 --maxdepth 12 --steps 20 --warmup 5" "--scene many_spheres --width 1024 --height 768 --steps 64 --warmup 8" \
         "--scene large/ajax_standin --width 1920 --height 1080 --maxdepth 4 --steps 20 --warmup 5" "--scene motionblur --width 1920 --height 1080 --steps 20 --warmup 5"; do
  bash scratch/gpu_envs.sh "$A" "-" "$A" "-" "$A" "-" -- $W
done
} > $O/ab_split_vs_call_a.md 2>&1; cat $O/ab_split_vs_call_a.md
{
echo "| library | workload | Msamples/s | kernel ms |"; echo "|---|---|---|---|"
for W in "--scene cornell --width 1024 --height 1024 --steps 20 --warmup 5" "--scene veach --width 3840 --height 2160 --steps 20 --warmup 5" \
         "--scene cornell --width 256 --height 256 --steps 16 --warmup 4" "--scene gloss --width 1024 --height 1024 --steps 20 --warmup 5" \
         "--scene features --width 1024 --height 1024 --steps 20 --warmup 5" "--scene large/env_loft --width 1024 --height 1024 --steps 20 --warmup 5"; do
  bash scratch/gpu_envs.sh "-" "$P" "-" "$P" -- $W
done
} > $O/ab_plane_table_fused.md 2>&1; cat $O/ab_plane_table_fused.md
{
echo "| environment | workload | Msamples/s | kernel ms |"; echo "|---|---|---|---|"
for W in "--scene cornell --width 1024 --height 1024 --steps 20 --warmup 5" "--scene veach --width 3840 --height 2160 --steps 20 --warmup 5" "--scene cornell --width 512 --height 512 --steps 16 --warmup 4"; do
  bash scratch/gpu_envs.sh "-" "TINSEL_HIP_TAIL_SPLIT=-0.25,4" "TINSEL_HIP_TAIL_SPLIT=-1.0,4" "TINSEL_HIP_TAIL_SPLIT=-0.5,2" "TINSEL_HIP_TAIL_SPLIT=-0.5,8" "TINSEL_HIP_TAIL_SPLIT=0" "TINSEL_HIP_GRID_MULT=16" "TINSEL_HIP_GRID_MULT=48" -- $W
done
} > $O/ab_tail_split_4waves.md 2>&1; cat $O/ab_tail_split_4waves.md
