#!/bin/bash
for rep in 1 2; do
for m in 4 8 16 32 64; do
  TINSEL_HIP_GRID_MULT=$m timeout 300 python bench.py --steps 128 --warmup 8 --no-cpu-baseline 2>/dev/null > /tmp/b.json
  python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('grid mult %3d Msamples/s %7.1f' % ($m, d['value']), d['roofline']['kernel_ms'])
PY
done
done
