#!/bin/bash
# round 3, call I: the suite on the new defaults (k_bounce / k_shade at three waves per SIMD), four waves, pools at three waves
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3i; mkdir -p $OUT
( time timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 2>&1 | tail -15 ) > $OUT/pytest_gpu.log 2>&1
grep -v "^Host\|^Librccl\|^RCCL\|^HIP ver\|^ROCm" $OUT/pytest_gpu.log | tail -8
run() { timeout 600 python bench.py "$@" --no-cpu-baseline --no-pmc --no-second-config --no-api --no-fast --no-ubench 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('| %s | %s | %.1f | %s |' % ("$TAG", d['config']['workload'].split(',')[0], d['value'], d['roofline']['kernel_ms']))
PY
}
for L in tinsel_amd/libtinsel_hip.so scratch/ab/libtinsel_hip_bounce4.so; do export TAG=$L; export TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/$L
  run --scene cornell --steps 20 --warmup 5
  run --scene cornell --steps 64 --warmup 5
  run --scene cornell --width 256 --height 256 --steps 16 --warmup 4
  run --scene veach --width 3840 --height 2160 --steps 8 --warmup 1
  run --scene gloss --steps 64 --warmup 8
  run --scene large/env_loft --width 1024 --height 512 --steps 64 --warmup 2
  run --scene features --width 1920 --height 1080 --maxdepth 6 --steps 16 --warmup 1
done 2>&1 | tee $OUT/ab_waves4.txt
for L in tinsel_amd/libtinsel_hip.so scratch/ab/libtinsel_hip_shade4.so; do export TAG=$L; export TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/$L
  run --scene glass --width 1920 --height 1080 --maxdepth 12 --steps 32 --warmup 2
  run --scene large/ajax_standin --width 1920 --height 1080 --steps 32 --warmup 2
  run --scene many_spheres --width 1024 --height 768 --steps 64 --warmup 2
done 2>&1 | tee -a $OUT/ab_waves4.txt
unset TINSEL_HIP_LIB
echo "=== pools at three waves per SIMD: default heuristic / forced on / forced off"
for R in default 1 0; do export TAG="TINSEL_HIP_REPACK=$R"; [ $R = default ] && unset TINSEL_HIP_REPACK || export TINSEL_HIP_REPACK=$R
  run --scene veach --width 3840 --height 2160 --steps 8 --warmup 1
  run --scene features --width 1920 --height 1080 --maxdepth 6 --steps 16 --warmup 1
  run --scene large/env_loft --width 1024 --height 512 --steps 64 --warmup 2
  run --scene gloss --steps 64 --warmup 8
  run --scene cornell --steps 20 --warmup 5
done 2>&1 | tee $OUT/ab_repack3.txt
unset TINSEL_HIP_REPACK
echo "=== the tolerance arm"
export TAG="arith fast"
( run --scene cornell --steps 20 --warmup 5 --arith fast; run --scene veach --width 3840 --height 2160 --steps 8 --warmup 1 --arith fast; run --scene glass --width 1920 --height 1080 --maxdepth 12 --steps 32 --warmup 2 --arith fast; run --scene large/ajax_standin --width 1920 --height 1080 --steps 32 --warmup 2 --arith fast ) 2>&1 | tee $OUT/fast_arm.txt
