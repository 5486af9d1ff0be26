#!/bin/bash
# fused vs split pipeline on the configs near the crossover
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
run() { timeout 600 python bench.py "$@" --no-cpu-baseline --no-pmc --no-second-config --no-api --no-fast 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
r=d['roofline']
print('| %s | %.1f | %s |' % (d['config']['workload'].split(',')[0], d['value'], r['kernel_ms']))
PY
}
for P in wavefront split; do
echo "== $P"
run --scene veach --width 3840 --height 2160 --steps 8 --warmup 1 --pipeline $P
run --scene features --width 1920 --height 1080 --maxdepth 6 --steps 16 --warmup 1 --pipeline $P
run --scene cornell --steps 64 --warmup 8 --pipeline $P
run --scene gloss --steps 64 --warmup 8 --pipeline $P
run --scene large/env_loft --width 1024 --height 512 --steps 64 --warmup 2 --pipeline $P
run --scene features_probe --width 1920 --height 1080 --maxdepth 6 --steps 16 --warmup 1 --pipeline $P
done
