#!/bin/bash
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for p in wavefront mega split; do
  timeout 600 python bench.py --scene large/ajax_standin --width 1920 --height 1080 --steps 16 --warmup 2 --pipeline $p --no-cpu-baseline 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
  python - <<PY
import json
d=json.load(open('/tmp/b.json'))
r=d['roofline']
print('ajax %-10s Msamples/s %7.1f Mrays/s %8.1f rays/sample %.2f B_ray %.0f I %.1f T %.2f P %.2f achieved %.0f GB/s' % ('$p', d['value'], d['mrays_per_s'], d['config']['rays_per_sample'], r['B_ray'], r['I'], r['T'], r['P'], r['achieved']), r['kernel_ms'])
PY
done
