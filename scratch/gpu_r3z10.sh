#!/bin/bash
# round 3, call Z10: one-set batches: which share / length of short regions
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3z10; mkdir -p $OUT
run() { timeout 300 python bench.py "$@" --no-cpu-baseline --no-pmc --no-second-config --no-api --no-fast --no-ubench 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('| %s | %s | %.1f | %s |' % ("$TAG", d['config']['workload'].split(',')[0] + " x$STEPS", d['value'], d['roofline']['kernel_ms']))
PY
}
for T in default 0.125,3 0.25,3 0.375,3 0.5,3 0.25,6 0.5,6 0.25,3; do
  if [ $T = default ]; then unset TINSEL_HIP_TAIL_SPLIT; else export TINSEL_HIP_TAIL_SPLIT=$T; fi
  export TAG="TAIL_SPLIT=$T"
  STEPS=16 run --scene cornell --width 256 --height 256 --steps 16 --warmup 4
  STEPS=4 run --scene cornell --width 512 --height 512 --steps 4 --warmup 4
  STEPS=16 run --scene veach --width 256 --height 256 --steps 16 --warmup 4
  STEPS=2 run --scene cornell --steps 2 --warmup 4
done 2>&1 | tee $OUT/ab_small_tail2.txt
