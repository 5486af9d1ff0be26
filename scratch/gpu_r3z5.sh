#!/bin/bash
# round 3, call Z5: tail split parameters over more scenes and batch sizes
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3z5; mkdir -p $OUT
run() { timeout 600 python bench.py "$@" --no-cpu-baseline --no-pmc --no-second-config --no-api --no-fast --no-ubench 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('| %s | %s | %.1f | %s |' % ("$TAG", d['config']['workload'].split(',')[0] + " x$STEPS", d['value'], d['roofline']['kernel_ms']))
PY
}
for T in off 0.09,4 0.125,4 0.18,4 0.125,3 0.125,2 0.18,2; do
  if [ $T = off ]; then unset TINSEL_HIP_TAIL_SPLIT; else export TINSEL_HIP_TAIL_SPLIT=$T; fi
  export TAG="TAIL_SPLIT=$T"
  STEPS=20 run --scene cornell --steps 20 --warmup 5
  STEPS=8 run --scene cornell --steps 8 --warmup 5
  STEPS=16 run --scene cornell --width 512 --height 512 --steps 16 --warmup 4
  STEPS=16 run --scene features --width 1920 --height 1080 --maxdepth 6 --steps 16 --warmup 1
  STEPS=64 run --scene gloss --steps 64 --warmup 8
  STEPS=64 run --scene large/env_loft --width 1024 --height 512 --steps 64 --warmup 2
  STEPS=20 run --scene veach --width 1920 --height 1080 --steps 20 --warmup 1
done 2>&1 | tee $OUT/ab_tail_split_params.txt
