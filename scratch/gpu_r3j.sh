#!/bin/bash
# round 3, call J: minimum grid of the streaming kernels at three waves per SIMD (small batches)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3j; mkdir -p $OUT
run() { timeout 600 python bench.py "$@" --no-cpu-baseline --no-pmc --no-second-config --no-api --no-fast --no-ubench 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('| %s | %s | %.1f | %s |' % ("$TAG", d['config']['workload'].split(',')[0], d['value'], d['roofline']['kernel_ms']))
PY
}
for GM in 2 3 4 6 8; do export TINSEL_HIP_GRID_MIN=$GM; export TAG="grid min $GM per CU"
  run --scene cornell --width 256 --height 256 --steps 16 --warmup 4
  run --scene cornell --width 512 --height 512 --steps 8 --warmup 4
  run --scene many_spheres --width 512 --height 384 --steps 8 --warmup 2
done 2>&1 | tee $OUT/ab_grid_min.txt
