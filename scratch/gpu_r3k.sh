#!/bin/bash
# round 3, call K: light sampling inside the (non-lean) k_extend for scenes with a staged arena and meshes in HBM (glass)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3k; mkdir -p $OUT
( TINSEL_HIP_LIGHTS_IN_EXTEND=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_walk.py tests/test_fuzz.py tests/test_gpu_split.py -m gpu -q --maxfail=10 2>&1 | tail -4 ) | tee $OUT/pytest_lights.log
run() { timeout 600 python bench.py "$@" --no-cpu-baseline --no-pmc --no-second-config --no-api --no-fast --no-ubench 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('| %s | %s | %.1f | %s |' % ("$TAG", d['config']['workload'].split(',')[0], d['value'], d['roofline']['kernel_ms']))
PY
}
for S in 0 1 0 1; do export TINSEL_HIP_LIGHTS_IN_EXTEND=$S; export TAG="LIGHTS_IN_EXTEND=$S"
  run --scene glass --width 1920 --height 1080 --maxdepth 12 --steps 32 --warmup 2
done 2>&1 | tee $OUT/ab_lights_in_extend.txt
