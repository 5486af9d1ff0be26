#!/bin/bash
# round 3, call H: exact-arm k_shade / k_bounce squeezed to three waves per SIMD (spills) under -fno-slp-vectorize
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3h; mkdir -p $OUT
run() { timeout 600 python bench.py "$@" --no-cpu-baseline --no-pmc --no-second-config --no-api --no-fast --no-ubench 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('| %s | %s | %.1f | %s |' % ("$TAG", d['config']['workload'].split(',')[0], d['value'], d['roofline']['kernel_ms']))
PY
}
for L in tinsel_amd/libtinsel_hip.so scratch/ab/libtinsel_hip_shade3.so; do export TAG=$L; export TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/$L
  run --scene glass --width 1920 --height 1080 --maxdepth 12 --steps 32 --warmup 2
  run --scene large/ajax_standin --width 1920 --height 1080 --steps 32 --warmup 2
  run --scene many_spheres --width 1024 --height 768 --steps 64 --warmup 2
done 2>&1 | tee $OUT/ab_waves3.txt
for L in tinsel_amd/libtinsel_hip.so scratch/ab/libtinsel_hip_bounce3.so; do export TAG=$L; export TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/$L
  run --scene cornell --steps 20 --warmup 5
  run --scene veach --width 3840 --height 2160 --steps 8 --warmup 1
  run --scene gloss --steps 64 --warmup 8
done 2>&1 | tee -a $OUT/ab_waves3.txt
