#!/bin/bash
# round 3, call B: suite on the new defaults (-fno-slp-vectorize, all bounces in one launch, pools on open scenes), k_walk with a
# short LDS stack + overflow at two workgroups per CU, launch-bound variants, per-bounce vs one launch
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/r3b; mkdir -p $OUT
( time timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 2>&1 | tail -40 ) > $OUT/pytest_gpu.log 2>&1
tail -12 $OUT/pytest_gpu.log
echo "=== walk tests with a 4-entry LDS stack (overflow in HBM)"
( TINSEL_HIP_WALK_LDS_STACK=4 timeout 900 python -m pytest tests/test_gpu_walk.py tests/test_gpu_configs.py tests/test_gpu_refit.py -m gpu -q --maxfail=5 -k "not full" 2>&1 | tail -5 ) | tee $OUT/pytest_walkstack.log
run() { timeout 600 python bench.py "$@" --no-cpu-baseline --no-pmc --no-second-config --no-api --no-fast --no-ubench 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('| %s | %s | %.1f | %s |' % ("$TAG", d['config']['workload'].split(',')[0], d['value'], d['roofline']['kernel_ms']))
PY
}
echo "=== per-bounce launches vs one launch (fused pipeline)"
for MODE in all per; do
  export TAG="bounce launches: $MODE"; [ $MODE = per ] && export TINSEL_HIP_BOUNCE_LAUNCHES=per || unset TINSEL_HIP_BOUNCE_LAUNCHES
  run --scene cornell --steps 20 --warmup 5
  run --scene cornell --steps 64 --warmup 8
  run --scene cornell --width 256 --height 256 --steps 16 --warmup 4
  run --scene veach --width 3840 --height 2160 --steps 8 --warmup 1
  run --scene large/env_loft --width 1024 --height 512 --steps 64 --warmup 2
  run --scene features --width 1920 --height 1080 --maxdepth 6 --steps 16 --warmup 1
  run --scene gloss --steps 64 --warmup 8
done 2>&1 | tee $OUT/ab_bounce_launches.txt
unset TINSEL_HIP_BOUNCE_LAUNCHES
echo "=== pools forced on / off"
for R in 0 1; do export TAG="TINSEL_HIP_REPACK=$R"; export TINSEL_HIP_REPACK=$R
  run --scene cornell --steps 64 --warmup 8
  run --scene veach --width 3840 --height 2160 --steps 8 --warmup 1
done 2>&1 | tee $OUT/ab_repack.txt
unset TINSEL_HIP_REPACK
echo "=== k_walk LDS stack / two workgroups per CU"
for S in 0 6 8 12; do export TAG="WALK_LDS_STACK=$S"; [ $S != 0 ] && export TINSEL_HIP_WALK_LDS_STACK=$S || unset TINSEL_HIP_WALK_LDS_STACK
  run --scene large/ajax_standin --width 1920 --height 1080 --steps 32 --warmup 2
  run --scene glass --width 1920 --height 1080 --maxdepth 12 --steps 32 --warmup 2
done 2>&1 | tee $OUT/ab_walk_stack.txt
unset TINSEL_HIP_WALK_LDS_STACK
echo "=== launch bounds"
for L in tinsel_amd/libtinsel_hip.so scratch/ab/libtinsel_hip_w5.so scratch/ab/libtinsel_hip_w6.so; do export TAG=$L; export TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/$L
  run --scene large/ajax_standin --width 1920 --height 1080 --steps 32 --warmup 2
  run --scene glass --width 1920 --height 1080 --maxdepth 12 --steps 32 --warmup 2
  run --scene many_spheres --width 1024 --height 768 --steps 64 --warmup 2
done 2>&1 | tee $OUT/ab_bounds.txt
unset TINSEL_HIP_LIB
echo "=== default bench line"
( time timeout 1200 python bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2>&1 | tail -3
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3b/bench_default.json'))
print('headline', d['value'], d['roofline']['frac'], d.get('yard_sticks'))
for c in d.get('configs', []):
    r=c.get('roofline') or {}
    print(c['config']['workload'][:40], c.get('value'), r.get('kernel'), r.get('frac'), r.get('l2_hit_rate'), c.get('unavailable'))
print('api', d.get('pcie_inclusive_msamples_s'), d.get('api_1pass_plain_msamples_s'), d.get('api_1pass_msamples_s'), d.get('api_1pass_pinned_output_msamples_s'))
PY
