#!/bin/bash
# round 3, call Z3: k_shade with the first shadow ray's records requested a round ahead (-DTN_SHADE_NEE_AHEAD=1) against the default build
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3z3; mkdir -p $OUT
( TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/libtinsel_hip_preload.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_split.py tests/test_gpu_walk.py -m gpu -q -x 2>&1 | grep -aE "passed|failed" | tail -2 ) 2>&1 | tee $OUT/pytest_subset.log
run() { timeout 600 python bench.py "$@" --no-cpu-baseline --no-pmc --no-second-config --no-api --no-fast --no-ubench 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('| %s | %s | %.1f | %s |' % ("$TAG", d['config']['workload'].split(',')[0], d['value'], d['roofline']['kernel_ms']))
PY
}
for R in base preload base preload; do
  if [ $R = base ]; then unset TINSEL_HIP_LIB; else export TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/libtinsel_hip_$R.so; fi
  export TAG="$R"
  run --scene large/ajax_standin --width 1920 --height 1080 --steps 32 --warmup 2
  run --scene glass --width 1920 --height 1080 --maxdepth 12 --steps 32 --warmup 2
  run --scene many_spheres --width 1024 --height 768 --steps 64 --warmup 2
done 2>&1 | tee $OUT/ab_walk_preload.txt
