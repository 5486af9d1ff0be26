#!/bin/bash
# round 3, call W: branch-free short sqrt (variant 21, the default build here) vs guarded (11) vs the compiler's expansion
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3w; mkdir -p $OUT
python - 2>&1 <<'PY' | grep -v amdgpu.ids | tee $OUT/selftest.txt
import tinsel_amd
for v in (21, -1):
    c, first = tinsel_amd.selftest_arith(1, v)
    print("sqrt variant %d: mismatches over 2^32 inputs: %d (first bad 0x%08x) %s" % (v, c[0], first, {e: n for e, n in enumerate(c[4:]) if n}))
c, first = tinsel_amd.selftest_arith(0)
print("rcp as built: mismatches %d" % c[0])
PY
run() { timeout 600 python bench.py "$@" --no-cpu-baseline --no-pmc --no-second-config --no-api --no-fast --no-ubench 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('| %s | %s | %.1f | %s |' % ("$TAG", d['config']['workload'].split(',')[0], d['value'], d['roofline']['kernel_ms']))
PY
}
for R in ieee sqrtonly free21 ieee sqrtonly free21; do
  if [ $R = free21 ]; then unset TINSEL_HIP_LIB; else export TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/libtinsel_hip_$R.so; fi
  export TAG="$R"
  run --scene cornell --steps 20 --warmup 5
  run --scene veach --width 3840 --height 2160 --steps 8 --warmup 1
  run --scene large/ajax_standin --width 1920 --height 1080 --steps 32 --warmup 2
  run --scene glass --width 1920 --height 1080 --maxdepth 12 --steps 32 --warmup 2
done 2>&1 | tee $OUT/ab_sqrt.txt
