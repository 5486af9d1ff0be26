#!/bin/bash
# A/B of library builds on the fused-pipeline configs: gpu_fused_ab.sh lib_a lib_b ...
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
run() { timeout 600 python bench.py "$@" --no-cpu-baseline --no-pmc --no-second-config --no-api 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json')); f=d.get('fast') or {}
print('| %s | %.1f | %s | fast %s |' % (d['config']['workload'].split(',')[0], d['value'], d['roofline']['kernel_ms'], ('%.1f' % f['msamples_s']) if f.get('msamples_s') else '-'))
PY
}
for L in "$@"; do
export TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/$L
echo "== $L"
run --scene cornell --steps 64 --warmup 8
run --scene cornell --width 256 --height 256 --steps 16 --warmup 2
run --scene veach --width 3840 --height 2160 --steps 8 --warmup 1
run --scene large/env_loft --width 1024 --height 512 --steps 64 --warmup 2
run --scene gloss --steps 64 --warmup 8
run --scene features --width 1920 --height 1080 --maxdepth 6 --steps 16 --warmup 1
done
