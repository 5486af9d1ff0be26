#!/bin/bash
for sc in glass veach features; do
 for p in wavefront split; do
  timeout 300 python bench.py --scene $sc --steps 32 --warmup 2 --pipeline $p --no-cpu-baseline 2>/dev/null > /tmp/b.json
  python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('%-14s %-10s Msamples/s %7.1f Mrays/s %8.1f' % ('$sc', '$p', d['value'], d['mrays_per_s']), d['roofline']['kernel_ms'])
PY
 done
done
