#!/bin/bash
# round 5, call g: the one-primitive walk three ways -- the previous library (prev), one refill source for both cases (v1: call f), the old
# refill source under `if constexpr (SINGLE)` (this tree)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5g; mkdir -p $O
P=TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/libtinsel_hip_prev.so
V=TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/libtinsel_hip_v1.so
{
echo "| library | workload | Msamples/s | kernel ms |"; echo "|---|---|---|---|"
for W in "--scene large/ajax_standin --width 1920 --height 1080 --maxdepth 4 --steps 20 --warmup 5" "--scene ajax_standin_96 --width 1920 --height 1080 --maxdepth 4 --steps 20 --warmup 5" \
         "--scene motionblur --width 1920 --height 1080 --steps 20 --warmup 5"; do
  bash scratch/gpu_envs.sh "$P" "$V" "-" "$P" "$V" "-" "$P" "$V" "-" -- $W
done
} > $O/ab_walk_single_refill.md 2>&1; cat $O/ab_walk_single_refill.md
( time timeout 300 python -m pytest tests/test_gpu_walk.py tests/test_gpu_configs.py -m gpu -q -x 2>&1 | tail -5 ) > $O/pytest_subset.log 2>&1; grep -a "passed\|failed\|rror" $O/pytest_subset.log | tail -3
