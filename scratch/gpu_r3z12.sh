#!/bin/bash
# round 3, call Z12: batches of a few resident sets cut like the one-set batches (TINSEL_HIP_FEW_SETS=sets,perCU,share,divide)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3z12; mkdir -p $OUT
run() { timeout 60 python bench.py "$@" --no-cpu-baseline --no-pmc --no-second-config --no-api --no-fast --no-ubench 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('| %s | %s | %.1f | %s |' % ("$TAG", d['config']['workload'].split(',')[0] + " x$STEPS", d['value'], d['roofline']['kernel_ms']))
PY
}
for T in off 6,2,0.7,4 6,3,0.75,3 6,2,0.6,4; do
  if [ $T = off ]; then unset TINSEL_HIP_FEW_SETS; else export TINSEL_HIP_FEW_SETS=$T; fi
  export TAG="FEW_SETS=$T"
  STEPS=16 run --scene cornell --width 512 --height 512 --steps 16 --warmup 3
  STEPS=8 run --scene cornell --steps 8 --warmup 3
done 2>&1 | tee $OUT/ab_few_sets.txt
