#!/bin/bash
# round 3, call M: compiler flag variants (-O2, -fno-unroll-loops, -fno-vectorize) against the default build
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3m; mkdir -p $OUT
run() { timeout 600 python bench.py "$@" --no-cpu-baseline --no-pmc --no-second-config --no-api --no-fast --no-ubench 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('| %s | %s | %.1f | %s |' % ("$TAG", d['config']['workload'].split(',')[0], d['value'], d['roofline']['kernel_ms']))
PY
}
for L in tinsel_amd/libtinsel_hip.so scratch/ab/libtinsel_hip_o2.so scratch/ab/libtinsel_hip_nounroll.so scratch/ab/libtinsel_hip_novec.so; do export TAG=$(basename $L); export TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/$L
  run --scene cornell --steps 20 --warmup 5
  run --scene veach --width 3840 --height 2160 --steps 8 --warmup 1
  run --scene large/ajax_standin --width 1920 --height 1080 --steps 32 --warmup 2
  run --scene glass --width 1920 --height 1080 --maxdepth 12 --steps 32 --warmup 2
done 2>&1 | tee $OUT/ab_flags.txt
