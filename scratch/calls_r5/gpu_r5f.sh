#!/bin/bash
# round 5, call f: the one-primitive walk with its two per-lane LDS rows back (call e: config 3's k_walk 10.7 -> 11.4 ms with three) against the
# previous library; walk tests
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5f; mkdir -p $O
( time timeout 600 python -m pytest tests/test_gpu_walk.py tests/test_gpu_configs.py tests/test_gpu_reference_scenes.py -m gpu -q -x 2>&1 | tail -8 ) > $O/pytest_subset.log 2>&1; grep -a "passed\|failed\|rror" $O/pytest_subset.log | tail -5
P=TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/libtinsel_hip_prev.so
{
echo "| library | workload | Msamples/s | kernel ms |"; echo "|---|---|---|---|"
for W in "--scene large/ajax_standin --width 1920 --height 1080 --maxdepth 4 --steps 20 --warmup 5" "--scene ajax_standin_96 --width 1920 --height 1080 --maxdepth 4 --steps 20 --warmup 5" \
         "--scene motionblur --width 1920 --height 1080 --steps 20 --warmup 5" "--scene glass --width 1920 --height 1080 --maxdepth 12 --steps 20 --warmup 5"; do
  bash scratch/gpu_envs.sh "$P" "-" "$P" "-" "$P" "-" -- $W
done
} > $O/ab_walk_single_rows.md 2>&1; cat $O/ab_walk_single_rows.md
