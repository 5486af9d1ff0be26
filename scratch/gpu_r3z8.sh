#!/bin/bash
# round 3, call Z8: the short regions' share of a batch as a multiple of one resident set's part (TINSEL_HIP_TAIL_SPLIT=-m,4) against the fixed eighth
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3z8; mkdir -p $OUT
run() { timeout 300 python bench.py "$@" --no-cpu-baseline --no-pmc --no-second-config --no-api --no-fast --no-ubench 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('| %s | %s | %.1f | %s |' % ("$TAG", d['config']['workload'].split(',')[0] + " x$STEPS", d['value'], d['roofline']['kernel_ms']))
PY
}
for T in 0.125,4 -0.5,4 -0.35,4 -0.75,4 0.125,4 -0.5,4; do
  export TINSEL_HIP_TAIL_SPLIT=$T
  export TAG="TAIL_SPLIT=$T"
  STEPS=20 run --scene cornell --steps 20 --warmup 5
  STEPS=64 run --scene cornell --steps 64 --warmup 5
  STEPS=16 run --scene features --width 1920 --height 1080 --maxdepth 6 --steps 16 --warmup 1
  STEPS=20 run --scene veach --width 1920 --height 1080 --steps 20 --warmup 1
  STEPS=64 run --scene gloss --steps 64 --warmup 8
done 2>&1 | tee $OUT/ab_tail_share.txt
