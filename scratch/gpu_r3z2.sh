#!/bin/bash
# round 3, call Z2: 1/sqrt(x) as the reference rounds it, RN(1/RN(sqrt x)), in one proven sequence (TN_RSQRT_VARIANT=1, the build here) vs the two sequences one after the other (rsq0)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3z2; mkdir -p $OUT
python - 2>&1 <<'PY' | grep -v amdgpu.ids | tee $OUT/selftest.txt
import tinsel_amd
for op, name in ((0, "1/x"), (1, "sqrt"), (2, "1/sqrt")):
    for v in ((-1,) if op < 2 else (0, 1, 2, 3, -1)):
        c, first = tinsel_amd.selftest_arith(op, v)
        print("%s variant %d: mismatches over 2^32 operands: %d (denormal %d, negative/big %d, other %d; first bad 0x%08x) %s" % (name, v, c[0], c[1], c[2], c[3], first, {e: n for e, n in enumerate(c[4:]) if n}))
PY
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_leaf.py tests/test_gpu_configs.py -m gpu -q -x 2>&1 | grep -aE "passed|failed" | tail -2 ) 2>&1 | tee $OUT/pytest_subset.log
run() { timeout 600 python bench.py "$@" --no-cpu-baseline --no-pmc --no-second-config --no-api --no-fast --no-ubench 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('| %s | %s | %.1f | %s |' % ("$TAG", d['config']['workload'].split(',')[0], d['value'], d['roofline']['kernel_ms']))
PY
}
for R in rsq0 fused3 fused rsq0 fused3 fused; do
  if [ $R = fused ]; then unset TINSEL_HIP_LIB; else export TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/libtinsel_hip_$R.so; fi
  export TAG="$R"
  run --scene cornell --steps 20 --warmup 5
  run --scene veach --width 3840 --height 2160 --steps 8 --warmup 1
  run --scene large/ajax_standin --width 1920 --height 1080 --steps 32 --warmup 2
  run --scene glass --width 1920 --height 1080 --maxdepth 12 --steps 32 --warmup 2
done 2>&1 | tee $OUT/ab_rsqrt.txt
