#!/bin/bash
# round 3, call X: small batches (cfg1: 256^2 x 16 passes = one 1 M-path batch) -- waves sharing their workgroup's regions, grid per CU, region length
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3x; mkdir -p $OUT
run() { timeout 600 python bench.py "$@" --no-cpu-baseline --no-pmc --no-second-config --no-api --no-fast --no-ubench 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('| %s | %s | %.1f | %s |' % ("$TAG", d['config']['workload'].split(',')[0], d['value'], d['roofline']['kernel_ms']))
PY
}
small() { run --scene cornell --width 256 --height 256 --steps 16 --warmup 4; }
for rep in 1 2; do
TAG="default" small
TINSEL_HIP_BOUNCE_SHARE=1 TAG="share" small
TINSEL_HIP_GRID_MIN=2 TAG="grid_min=2" small
TINSEL_HIP_GRID_MIN=2 TINSEL_HIP_BOUNCE_SHARE=1 TAG="grid_min=2 share" small
TINSEL_HIP_GRID_MIN=4 TAG="grid_min=4" small
TINSEL_HIP_GRID_MIN=4 TINSEL_HIP_BOUNCE_SHARE=1 TAG="grid_min=4 share" small
TINSEL_HIP_GRID_MIN=6 TAG="grid_min=6" small
TINSEL_HIP_GRID_MIN=6 TINSEL_HIP_BOUNCE_SHARE=1 TAG="grid_min=6 share" small
done 2>&1 | tee $OUT/ab_small_share.txt
for rep in 1 2; do
TAG="default" run --scene cornell --steps 20 --warmup 5
TINSEL_HIP_BOUNCE_SHARE=1 TAG="share" run --scene cornell --steps 20 --warmup 5
TAG="default" run --scene cornell --width 512 --height 512 --steps 16 --warmup 4
TINSEL_HIP_BOUNCE_SHARE=1 TAG="share" run --scene cornell --width 512 --height 512 --steps 16 --warmup 4
done 2>&1 | tee $OUT/ab_share_large.txt
