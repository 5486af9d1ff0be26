#!/bin/bash
# round 3, call U: the proven short reciprocal / sqrt sequences (TN_RCP_VARIANT=11, TN_SQRT_VARIANT=11) against the compiler's
# expansions (scratch/ab/libtinsel_hip_ieee.so): exhaustive self-test, parity subset, A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3u; mkdir -p $OUT
python - 2>&1 <<'PY' | grep -v amdgpu.ids | tee $OUT/selftest.txt
import tinsel_amd
for op, name in ((0, "rcp"), (1, "sqrt")):
    c, first = tinsel_amd.selftest_arith(op)
    print("%s as built: mismatches over 2^32 inputs: %d (first bad 0x%08x)" % (name, c[0], first))
PY
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_leaf.py -m gpu -q -x 2>&1 | tail -3 ) 2>&1 | tee $OUT/pytest_subset.log
run() { timeout 600 python bench.py "$@" --no-cpu-baseline --no-pmc --no-second-config --no-api --no-fast --no-ubench 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('| %s | %s | %.1f | %s |' % ("$TAG", d['config']['workload'].split(',')[0], d['value'], d['roofline']['kernel_ms']))
PY
}
for R in short ieee short ieee; do
  if [ $R = ieee ]; then export TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/libtinsel_hip_ieee.so; else unset TINSEL_HIP_LIB; fi
  export TAG="$R"
  run --scene cornell --steps 20 --warmup 5
  run --scene cornell --width 256 --height 256 --steps 16 --warmup 2
  run --scene veach --width 3840 --height 2160 --steps 8 --warmup 1
  run --scene large/ajax_standin --width 1920 --height 1080 --steps 32 --warmup 2
  run --scene glass --width 1920 --height 1080 --maxdepth 12 --steps 32 --warmup 2
  run --scene many_spheres --width 1024 --height 768 --steps 64 --warmup 2
done 2>&1 | tee $OUT/ab_short_arith.txt
