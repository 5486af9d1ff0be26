#!/bin/bash
# round 3, call P: device-built trees -- LBVH and PLOC, both with a breadth-first top -- against the reference's SAH trees
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3p; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_lbvh.py tests/test_gpu_refit.py -m gpu -q -s --maxfail=10 2>&1 | grep -E "passed|failed|built|Error|error" ) | tee $OUT/pytest.log
run() { timeout 600 python bench.py "$@" --no-cpu-baseline --no-pmc --no-second-config --no-api --no-fast --no-ubench 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('| %s | %s | %.1f | build %s ms | %s |' % ("$TAG", d['config']['workload'].split(',')[0], d['value'], d['config'].get('mesh_bvh_build_ms'), d['roofline']['kernel_ms']))
PY
}
for B in reference lbvh ploc; do export TAG="mesh BVH: $B"
  run --scene large/ajax_standin --width 1920 --height 1080 --steps 32 --warmup 2 --bvh $B
  run --scene glass --width 1920 --height 1080 --maxdepth 12 --steps 32 --warmup 2 --bvh $B
done 2>&1 | tee $OUT/ab_device_bvh.txt
