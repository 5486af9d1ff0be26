#!/bin/bash
# round 3, call S: grid rounded so that the last resident set of a launch is full
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3s; mkdir -p $OUT
run() { timeout 600 python bench.py "$@" --no-cpu-baseline --no-pmc --no-second-config --no-api --no-fast --no-ubench 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('| %s | %s | %.1f | %s |' % ("$TAG", d['config']['workload'].split(',')[0], d['value'], d['roofline']['kernel_ms']))
PY
}
for R in 0 1 0 1; do export TINSEL_HIP_GRID_ROUND=$R; export TAG="GRID_ROUND=$R"
  run --scene cornell --steps 20 --warmup 5
  run --scene cornell --steps 64 --warmup 5
  run --scene veach --width 3840 --height 2160 --steps 8 --warmup 1
  run --scene large/ajax_standin --width 1920 --height 1080 --steps 20 --warmup 2
  run --scene glass --width 1920 --height 1080 --maxdepth 12 --steps 20 --warmup 2
done 2>&1 | tee $OUT/ab_grid_round.txt
