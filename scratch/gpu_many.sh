#!/bin/bash
run() { timeout 600 python bench.py "$@" --no-cpu-baseline 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
r=d['roofline']
print('%-60s Msamples/s %7.1f Mrays/s %8.1f I %.1f T %.2f P %.2f' % (d['config']['workload'][:60], d['value'], d['mrays_per_s'], r['I'], r['T'], r['P']), r['kernel_ms'])
PY
}
for p in wavefront split mega; do run --scene many_spheres --width 1024 --height 768 --steps 16 --warmup 2 --pipeline $p; done
