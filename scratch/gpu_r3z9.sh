#!/bin/bash
# round 3, call Z9: one-set batches (cfg1: 256^2 x 16 passes): short regions at the end there too?
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3z9; mkdir -p $OUT
run() { timeout 300 python bench.py "$@" --no-cpu-baseline --no-pmc --no-second-config --no-api --no-fast --no-ubench 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('| %s | %s | %.1f | %s |' % ("$TAG", d['config']['workload'].split(',')[0], d['value'], d['roofline']['kernel_ms']))
PY
}
small() { run --scene cornell --width 256 --height 256 --steps 16 --warmup 4; }
for rep in 1 2; do
TAG="default" small
TINSEL_HIP_TAIL_SPLIT=0.125,2 TAG="0.125,2" small
TINSEL_HIP_TAIL_SPLIT=0.25,2 TAG="0.25,2" small
TINSEL_HIP_TAIL_SPLIT=0.25,3 TAG="0.25,3" small
TINSEL_HIP_GRID_MIN=4 TINSEL_HIP_TAIL_SPLIT=0.25,2 TAG="grid_min=4 0.25,2" small
TINSEL_HIP_GRID_MIN=6 TINSEL_HIP_TAIL_SPLIT=0.25,2 TAG="grid_min=6 0.25,2" small
TINSEL_HIP_GRID_MIN=2 TINSEL_HIP_TAIL_SPLIT=0.3,4 TAG="grid_min=2 0.3,4" small
done 2>&1 | tee $OUT/ab_small_tail.txt
