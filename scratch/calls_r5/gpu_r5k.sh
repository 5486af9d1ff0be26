#!/bin/bash
# round 5, call k: the library's own sort and scan against numpy (tests/test_gpu_sort.py); a batch's passes as two overlapped chunks on two
# streams (TINSEL_HIP_OVERLAP=1) for the fused scenes again, now that k_bounce runs four waves per SIMD (round 4: -1..2 % at three)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5k; mkdir -p $O
( time timeout 300 python -m pytest tests/test_gpu_sort.py tests/test_gpu_lbvh.py -m gpu -q -x 2>&1 | tail -6 ) > $O/pytest_sort.log 2>&1; grep -a "passed\|failed\|rror" $O/pytest_sort.log | tail -4
{
echo "| environment | workload | Msamples/s | kernel ms |"; echo "|---|---|---|---|"
for W in "--scene cornell --width 1024 --height 1024 --steps 20 --warmup 5" "--scene veach --width 3840 --height 2160 --steps 20 --warmup 5" "--scene cornell --width 1024 --height 1024 --steps 64 --warmup 8" \
         "--scene gloss --width 1024 --height 1024 --steps 20 --warmup 5" "--scene glass --width 1920 --height 1080 --maxdepth 12 --steps 20 --warmup 5" "--scene large/ajax_standin --width 1920 --height 1080 --maxdepth 4 --steps 20 --warmup 5"; do
  bash scratch/gpu_envs.sh "-" "TINSEL_HIP_OVERLAP=1" "-" "TINSEL_HIP_OVERLAP=1" -- $W
done
} > $O/ab_overlap_4waves.md 2>&1; cat $O/ab_overlap_4waves.md
