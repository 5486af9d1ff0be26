#!/bin/bash
# round 5, call a: the new bench line (size, the driver's flags), the N-rank legs on one device, and k_bounce at four waves per SIMD
# (-DTN_WAVES_BOUNCE=4: 128 VGPRs, the compiler's own spills) with and without the frame parameters fetched late (-DTN_LATE_FRAME=1)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5a; mkdir -p $O
( time timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real
cp bench_detail.json $O/bench_detail_default.json; wc -c $O/bench_default.json; tail -c 600 $O/bench_default.json
( time timeout 600 python -m pytest tests/test_gpu_multirank.py -m gpu -q -x 2>&1 | tail -15 ) > $O/pytest_multirank.log 2>&1; grep -a "passed\|failed\|rror" $O/pytest_multirank.log | tail -5
{
echo "| library | workload | Msamples/s | kernel ms |"; echo "|---|---|---|---|"
for W in "--scene cornell --width 1024 --height 1024 --steps 20 --warmup 5" "--scene veach --width 3840 --height 2160 --steps 20 --warmup 5" \
         "--scene cornell --width 256 --height 256 --steps 16 --warmup 4" "--scene gloss --width 1024 --height 1024 --steps 20 --warmup 5" \
         "--scene features --width 1024 --height 1024 --steps 20 --warmup 5" "--scene large/env_loft --width 1024 --height 1024 --steps 20 --warmup 5"; do
  bash scratch/gpu_envs.sh "-" "TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/libtinsel_hip_w4.so" "TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/libtinsel_hip_w4lf.so" \
       "TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/libtinsel_hip_lf.so" -- $W
done
} > $O/ab_waves4.md 2>&1; cat $O/ab_waves4.md
# counters of the headline under the four-wave library
( export TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/libtinsel_hip_w4lf.so; timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-second-config --no-more-configs --no-api --no-fast > $O/bench_w4lf.json 2> $O/bench_w4lf.err; cp bench_detail.json $O/bench_detail_w4lf.json; cat $O/bench_w4lf.json | head -c 1500 )
# bit-identity of the variants
( export TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/libtinsel_hip_w4lf.so; time timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x 2>&1 | tail -3 ) > $O/pytest_w4lf.log 2>&1; grep -a "passed\|failed" $O/pytest_w4lf.log
# glass: k_walk's existing switches on the configuration as benched (whole trees in LDS with one workgroup per CU = TINSEL_HIP_WALK_LDS_STACK=0)
{
echo "| environment | workload | Msamples/s | kernel ms |"; echo "|---|---|---|---|"
bash scratch/gpu_envs.sh "-" "TINSEL_HIP_WALK_LDS_STACK=0" "TINSEL_HIP_WALK_REFILL=16" "TINSEL_HIP_WALK_REFILL=32" "TINSEL_HIP_WALK_REFILL=44" "TINSEL_HIP_WALK_LEAFMIN=4" "TINSEL_HIP_WALK_LEAFMIN=16" \
     "TINSEL_HIP_WALK_LDS_STACK=0 TINSEL_HIP_WALK_REFILL=32" "TINSEL_HIP_WALK_GRID_MULT=2" -- --scene glass --width 1920 --height 1080 --maxdepth 12 --steps 20 --warmup 5
} > $O/ab_glass_walk.md 2>&1; cat $O/ab_glass_walk.md
( export TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/libtinsel_hip_walkprof.so
  python scratch/walk_prof.py glass 1920 1080 12 20; TINSEL_HIP_WALK_LDS_STACK=0 python scratch/walk_prof.py glass 1920 1080 12 20 ) > $O/walk_profile_glass.txt 2>&1; cat $O/walk_profile_glass.txt
