#!/bin/bash
# round 3, call Z4: k_bounce's last regions short (TINSEL_HIP_TAIL_SPLIT=share,divide): the launch's tail is a short region's time
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3z4; mkdir -p $OUT
( TINSEL_HIP_TAIL_SPLIT=0.125,4 timeout 600 python tests/switch_probe.py cornell,veach,features,gloss 2>&1 | grep -v amdgpu | tr '\n' ' ' ) | tee $OUT/parity.txt; echo
( TINSEL_HIP_TAIL_SPLIT=0.3,8 timeout 600 python tests/switch_probe.py cornell,veach,features,gloss 2>&1 | grep -v amdgpu | tr '\n' ' ' ) | tee -a $OUT/parity.txt; echo
run() { timeout 600 python bench.py "$@" --no-cpu-baseline --no-pmc --no-second-config --no-api --no-fast --no-ubench 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('| %s | %s | %.1f | %s |' % ("$TAG", d['config']['workload'].split(',')[0], d['value'], d['roofline']['kernel_ms']))
PY
}
for rep in 1 2; do
for T in off 0.125,4 0.25,4 0.125,8 0.06,4; do
  if [ $T = off ]; then unset TINSEL_HIP_TAIL_SPLIT; else export TINSEL_HIP_TAIL_SPLIT=$T; fi
  export TAG="TAIL_SPLIT=$T"
  run --scene cornell --steps 20 --warmup 5
  run --scene cornell --steps 64 --warmup 5
  run --scene veach --width 3840 --height 2160 --steps 8 --warmup 1
  run --scene features --width 1920 --height 1080 --maxdepth 6 --steps 16 --warmup 1
done; done 2>&1 | tee $OUT/ab_tail_split.txt
