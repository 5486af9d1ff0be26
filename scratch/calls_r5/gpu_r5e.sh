#!/bin/bash
# round 5, call e: k_walk's work items by RAY (several walked primitives: a ray is fetched once and its primitives' leaf boxes tested from
# wave-uniform records; glass 2, table.tin 7 walked meshes) and the plane table in fused scenes with four planes or more -- whole suite, then
# this tree against the previous one (scratch/ab/libtinsel_hip_prev.so = commit 2f51fb7)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5e; mkdir -p $O
( time timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 ) > $O/pytest_gpu.log 2>&1; grep -a "passed\|failed\|rror" $O/pytest_gpu.log | tail -8
P=TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/libtinsel_hip_prev.so
{
echo "| library | workload | Msamples/s | kernel ms |"; echo "|---|---|---|---|"
for W in "--scene glass --width 1920 --height 1080 --maxdepth 12 --steps 20 --warmup 5" "--scene large/table --width 1024 --height 1024 --steps 20 --warmup 5" \
         "--scene large/transmission --width 1024 --height 1024 --steps 20 --warmup 5" "--scene large/meshlight --width 1024 --height 1024 --steps 20 --warmup 5" \
         "--scene large/ajax_standin --width 1920 --height 1080 --maxdepth 4 --steps 20 --warmup 5" "--scene motionblur --width 1024 --height 1024 --steps 20 --warmup 5" \
         "--scene cornell --width 1024 --height 1024 --steps 20 --warmup 5" "--scene gloss --width 1024 --height 1024 --steps 20 --warmup 5" "--scene large/env_loft --width 1024 --height 1024 --steps 20 --warmup 5"; do
  bash scratch/gpu_envs.sh "$P" "-" "$P" "-" -- $W
done
} > $O/ab_walk_by_ray.md 2>&1; cat $O/ab_walk_by_ray.md
bash scratch/gpu_pmc_kernels.sh $O byray "glass 1920 1080 12 20" "large/table 1024 1024 8 20" > /dev/null 2>&1; cat $O/pmc_byray.md
