#!/bin/bash
# round 3, call V: which part of the short sequences costs -- rcp only / sqrt only / one guard per direction (group3) against the compiler's expansions
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3v; mkdir -p $OUT
run() { timeout 600 python bench.py "$@" --no-cpu-baseline --no-pmc --no-second-config --no-api --no-fast --no-ubench 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('| %s | %s | %.1f | %s |' % ("$TAG", d['config']['workload'].split(',')[0], d['value'], d['roofline']['kernel_ms']))
PY
}
for R in ieee rcponly sqrtonly group3 ieee rcponly sqrtonly group3; do
  export TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/libtinsel_hip_$R.so
  export TAG="$R"
  run --scene cornell --steps 20 --warmup 5
  run --scene veach --width 3840 --height 2160 --steps 8 --warmup 1
  run --scene large/ajax_standin --width 1920 --height 1080 --steps 32 --warmup 2
  run --scene glass --width 1920 --height 1080 --maxdepth 12 --steps 32 --warmup 2
done 2>&1 | tee $OUT/ab_short_arith_parts.txt
