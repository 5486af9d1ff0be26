#!/bin/bash
# round 5, call b: the pruned library (no rocprim, no pair records, no shadow-in-shade, 22 switches) under the whole GPU suite; k_bounce with its
# wave-uniform bookkeeping in scalar registers at three / four waves per SIMD as the host plans them (shading pools of 25 fields); k_walk with
# the whole mesh in LDS (glass) and its two thresholds swept; the driver's bench line
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5b; mkdir -p $O
( time timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 ) > $O/pytest_gpu.log 2>&1; grep -a "passed\|failed\|rror" $O/pytest_gpu.log | tail -8
( time timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real
cp bench_detail.json $O/bench_detail_default.json; wc -c $O/bench_default.json; python - <<'PY'
import json
d=json.load(open('gpurun_out/r5b/bench_default.json'))
print('headline', d['value'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['waves_per_simd'], d['roofline']['avg_launch_ms'])
for c in d.get('configs', []): print(c)
PY
{
echo "| environment | workload | Msamples/s | kernel ms |"; echo "|---|---|---|---|"
for W in "--scene cornell --width 1024 --height 1024 --steps 20 --warmup 5" "--scene veach --width 3840 --height 2160 --steps 20 --warmup 5" \
         "--scene cornell --width 256 --height 256 --steps 16 --warmup 4" "--scene gloss --width 1024 --height 1024 --steps 20 --warmup 5" \
         "--scene features --width 1024 --height 1024 --steps 20 --warmup 5" "--scene large/env_loft --width 1024 --height 1024 --steps 20 --warmup 5"; do
  bash scratch/gpu_envs.sh "-" "TINSEL_HIP_BOUNCE_WAVES=3" "TINSEL_HIP_BOUNCE_WAVES=4" -- $W
done
} > $O/ab_bounce_waves.md 2>&1; cat $O/ab_bounce_waves.md
{
echo "| environment | workload | Msamples/s | kernel ms |"; echo "|---|---|---|---|"
bash scratch/gpu_envs.sh "-" "TINSEL_HIP_WALK_LDS_MESH=0" -- --scene glass --width 1920 --height 1080 --maxdepth 12 --steps 20 --warmup 5
bash scratch/gpu_envs.sh "-" "TINSEL_HIP_WALK_LDS_MESH=0" -- --scene motionblur --width 1024 --height 1024 --steps 20 --warmup 5
bash scratch/gpu_envs.sh "-" "TINSEL_HIP_WALK_LDS_MESH=0" -- --scene large/table --width 1024 --height 1024 --steps 20 --warmup 5
bash scratch/gpu_envs.sh "-" "TINSEL_HIP_WALK_LDS_MESH=0" -- --scene large/transmission --width 1024 --height 1024 --steps 20 --warmup 5
export TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/libtinsel_hip_tune.so
bash scratch/gpu_envs.sh "TN_TUNE_WALK_REFILL=8 TN_TUNE_WALK_LEAFMIN=1" "TN_TUNE_WALK_REFILL=16 TN_TUNE_WALK_LEAFMIN=1" "TN_TUNE_WALK_REFILL=24 TN_TUNE_WALK_LEAFMIN=1" "TN_TUNE_WALK_REFILL=32 TN_TUNE_WALK_LEAFMIN=1" \
     "TN_TUNE_WALK_REFILL=16 TN_TUNE_WALK_LEAFMIN=4" "TN_TUNE_WALK_REFILL=24 TN_TUNE_WALK_LEAFMIN=4" "TN_TUNE_WALK_REFILL=16 TN_TUNE_WALK_LEAFMIN=16" "TN_TUNE_WALK_REFILL=40 TN_TUNE_WALK_LEAFMIN=2" \
     -- --scene glass --width 1920 --height 1080 --maxdepth 12 --steps 20 --warmup 5
unset TINSEL_HIP_LIB
} > $O/ab_glass_lds_mesh.md 2>&1; cat $O/ab_glass_lds_mesh.md
( export TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/libtinsel_hip_walkprof.so
  python scratch/walk_prof.py glass 1920 1080 12 20; TINSEL_HIP_WALK_LDS_MESH=0 python scratch/walk_prof.py glass 1920 1080 12 20 ) > $O/walk_profile_glass.txt 2>&1; cat $O/walk_profile_glass.txt
