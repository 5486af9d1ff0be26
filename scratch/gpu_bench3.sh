#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
for p in wavefront mega split; do
  timeout 300 python bench.py --steps 64 --warmup 4 --pipeline $p --no-cpu-baseline 2>/dev/null > gpurun_out/bench_$p.json
  python - <<PY
import json
d=json.load(open('gpurun_out/bench_$p.json'))
print('$p', 'Msamples/s %.1f Mrays/s %.1f' % (d['value'], d['mrays_per_s']), d['roofline']['kernel_ms'])
PY
done
