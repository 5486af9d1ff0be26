#!/bin/bash
# round 3, call R: launch bounds of the k_extend variants that draw the light samples
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3r; mkdir -p $OUT
run() { timeout 600 python bench.py "$@" --no-cpu-baseline --no-pmc --no-second-config --no-api --no-fast --no-ubench 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('| %s | %s | %.1f | %s |' % ("$TAG", d['config']['workload'].split(',')[0], d['value'], d['roofline']['kernel_ms']))
PY
}
for L in tinsel_amd/libtinsel_hip.so scratch/ab/libtinsel_hip_extl5.so scratch/ab/libtinsel_hip_scan5.so scratch/ab/libtinsel_hip_scan3.so; do export TAG=$(basename $L); export TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/$L
  run --scene glass --width 1920 --height 1080 --maxdepth 12 --steps 32 --warmup 2
  run --scene large/ajax_standin --width 1920 --height 1080 --steps 32 --warmup 2
done 2>&1 | tee $OUT/ab_extend_bounds.txt
