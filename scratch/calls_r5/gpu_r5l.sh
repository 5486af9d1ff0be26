#!/bin/bash
# round 5, call l: one shard with TILED path slots (TINSEL_HIP_SLOT_TILE=T: a wave's 64 slots are a block of pixels instead of a piece of a
# frame row, so its camera rays and first shadow rays visit the same BVH nodes) -- parity first, then every workload against row-major
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5l; mkdir -p $O
( time TINSEL_HIP_SLOT_TILE=8 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_walk.py tests/test_gpu_reference_scenes.py tests/test_gpu_split.py -m gpu -q -x 2>&1 | tail -6 ) > $O/pytest_tile8.log 2>&1; grep -a "passed\|failed\|rror" $O/pytest_tile8.log | tail -4
{
echo "| environment | workload | Msamples/s | kernel ms |"; echo "|---|---|---|---|"
for W in "--scene large/ajax_standin --width 1920 --height 1080 --maxdepth 4 --steps 20 --warmup 5" "--scene glass --width 1920 --height 1080 --maxdepth 12 --steps 20 --warmup 5" \
         "--scene table --width 1280 --height 720 --steps 20 --warmup 5" "--scene large/env_loft --width 1920 --height 1080 --steps 20 --warmup 5" \
         "--scene cornell --width 1024 --height 1024 --steps 20 --warmup 5" "--scene veach --width 3840 --height 2160 --steps 20 --warmup 5"; do
  bash scratch/gpu_envs.sh "-" "TINSEL_HIP_SLOT_TILE=8" "TINSEL_HIP_SLOT_TILE=16" "TINSEL_HIP_SLOT_TILE=4" "-" "TINSEL_HIP_SLOT_TILE=8" -- $W
done
} > $O/ab_slot_tile.md 2>&1; cat $O/ab_slot_tile.md
