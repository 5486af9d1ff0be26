#!/bin/bash
for sc in large/env_loft features_probe; do
 for p in wavefront split; do
  timeout 300 python bench.py --scene $sc --width 1024 --height 512 --steps 32 --warmup 2 --pipeline $p --no-cpu-baseline 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
  python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('%-16s %-10s Msamples/s %7.1f Mrays/s %8.1f rays/sample %.2f B_ray %.0f' % ('$sc', '$p', d['value'], d['mrays_per_s'], d['config']['rays_per_sample'], d['roofline']['B_ray']), d['roofline']['kernel_ms'])
PY
 done
done
