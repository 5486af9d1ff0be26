#!/bin/bash
for b in 2097152 4194304 8388608 16777216 33554432; do
  TINSEL_HIP_BATCH_PATHS=$b timeout 300 python bench.py --steps 128 --warmup 8 --no-cpu-baseline 2>/dev/null > /tmp/b.json
  python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('batch %9d Msamples/s %7.1f' % ($b, d['value']), d['roofline']['kernel_ms'], 'launches', d['roofline']['launches'])
PY
done
