#!/bin/bash
# A/B of two builds of the library within one gpurun call: parity tests on the new one, then one bench line per config for both
# usage: gpu_ab_shade.sh [lib_a lib_b ...]   (paths relative to the repo; default: scratch/ab/libtinsel_hip_base.so and the in-tree build)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/ab; mkdir -p $O
export TMPDIR=/tmp
LIBS="${@:-scratch/ab/libtinsel_hip_base.so tinsel_amd/libtinsel_hip.so}"
( timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_walk.py tests/test_gpu_probe.py tests/test_gpu_roulette.py tests/test_gpu_configs.py tests/test_fuzz.py -m gpu -q -x 2>&1 | tail -5 ) | tee $O/pytest.log
run() { timeout 600 python bench.py "$@" --no-cpu-baseline --no-pmc --no-second-config --no-api $FASTFLAG 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
r=d['roofline']; f=d.get('fast') or {}
print('| %s | %.1f | %.1f | %s | %s |' % (d['config']['workload'].split(',')[0], d['value'], d['mrays_per_s'], r['kernel_ms'], ('%.1f' % f['msamples_s']) if f.get('msamples_s') else '-'))
PY
}
for L in $LIBS; do
export TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/$L
echo "== $L"
run --scene cornell --steps 64 --warmup 8
run --scene large/ajax_standin --width 1920 --height 1080 --steps 32 --warmup 2
run --scene glass --width 1920 --height 1080 --maxdepth 12 --steps 32 --warmup 2
run --scene veach --width 3840 --height 2160 --steps 8 --warmup 1
run --scene large/env_loft --width 1024 --height 512 --steps 64 --warmup 2
run --scene many_spheres --width 1024 --height 768 --steps 64 --warmup 2
done 2>&1 | tee $O/ab.md
