#!/bin/bash
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for env in "" "TINSEL_HIP_NO_FLAT_SCAN=1"; do
for p in wavefront mega split; do
  env $env timeout 300 python bench.py --steps 64 --warmup 4 --pipeline $p --no-cpu-baseline 2>/dev/null > /tmp/b.json
  python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('%-26s %-10s Msamples/s %7.1f Mrays/s %7.1f' % ('$env', '$p', d['value'], d['mrays_per_s']), d['roofline']['kernel_ms'])
PY
done; done
python scratch/bench_normals.py cornell 2>&1 | tail -1
TINSEL_HIP_NO_FLAT_SCAN=1 python scratch/bench_normals.py cornell 2>&1 | tail -1
