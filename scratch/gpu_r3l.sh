#!/bin/bash
# round 3, call L: k_seg_prefix with the counts staged in LDS
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3l; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_walk.py tests/test_gpu_swalk.py tests/test_gpu_configs.py tests/test_gpu_parity.py -m gpu -q --maxfail=10 -k "not full" 2>&1 | tail -4 ) | tee $OUT/pytest.log
run() { timeout 600 python bench.py "$@" --no-cpu-baseline --no-pmc --no-second-config --no-api --no-fast --no-ubench 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('| %s | %s | %.1f | %s |' % ("$TAG", d['config']['workload'].split(',')[0], d['value'], d['roofline']['kernel_ms']))
PY
}
export TAG="k_seg_prefix in LDS"
( run --scene glass --width 1920 --height 1080 --maxdepth 12 --steps 32 --warmup 2
  run --scene large/ajax_standin --width 1920 --height 1080 --steps 32 --warmup 2
  run --scene many_spheres --width 1024 --height 768 --steps 64 --warmup 2 ) 2>&1 | tee $OUT/ab_seg.txt
