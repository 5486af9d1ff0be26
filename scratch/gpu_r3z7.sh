#!/bin/bash
# round 3, call Z7: k_walk's shared pool at the end of the work list (TINSEL_HIP_WALK_POOL_SHIFT: 0 none, 3 an eighth, 2 a quarter, 4 a sixteenth)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3z7; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_walk.py tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x 2>&1 | grep -aE "passed|failed" | tail -2 ) 2>&1 | tee $OUT/pytest_subset.log
run() { timeout 600 python bench.py "$@" --no-cpu-baseline --no-pmc --no-second-config --no-api --no-fast --no-ubench 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('| %s | %s | %.1f | %s |' % ("$TAG", d['config']['workload'].split(',')[0] + " x$STEPS", d['value'], d['roofline']['kernel_ms']))
PY
}
for rep in 1 2; do
for T in 0 3 2 4; do
  export TINSEL_HIP_WALK_POOL_SHIFT=$T
  export TAG="WALK_POOL_SHIFT=$T"
  STEPS=32 run --scene large/ajax_standin --width 1920 --height 1080 --steps 32 --warmup 2
  STEPS=20 run --scene large/ajax_standin --width 1920 --height 1080 --steps 20 --warmup 2
  STEPS=32 run --scene glass --width 1920 --height 1080 --maxdepth 12 --steps 32 --warmup 2
done; done 2>&1 | tee $OUT/ab_walk_pool.txt
