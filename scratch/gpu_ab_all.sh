#!/bin/bash
# A/B of library builds on every bench config (exact arm only): gpu_ab_all.sh [pytest] lib_a lib_b ...
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
if [ "$1" = "pytest" ]; then shift; ( timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_walk.py tests/test_gpu_probe.py tests/test_gpu_configs.py tests/test_fuzz.py tests/test_gpu_leaf.py -m gpu -q -x 2>&1 | tail -3 ); fi
run() { timeout 600 python bench.py "$@" --no-cpu-baseline --no-pmc --no-second-config --no-api --no-fast 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('| %s | %.1f | %s |' % (d['config']['workload'].split(',')[0], d['value'], d['roofline']['kernel_ms']))
PY
}
for L in "$@"; do
export TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/$L
echo "== $L"
run --scene cornell --steps 64 --warmup 8
run --scene large/ajax_standin --width 1920 --height 1080 --steps 32 --warmup 2
run --scene glass --width 1920 --height 1080 --maxdepth 12 --steps 32 --warmup 2
run --scene veach --width 3840 --height 2160 --steps 8 --warmup 1
run --scene large/env_loft --width 1024 --height 512 --steps 64 --warmup 2
run --scene many_spheres --width 1024 --height 768 --steps 64 --warmup 2
run --scene features --width 1920 --height 1080 --maxdepth 6 --steps 16 --warmup 1
run --scene gloss --steps 64 --warmup 8
done
