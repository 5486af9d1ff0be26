#!/bin/bash
for lib in "$@"; do
  for p in wavefront mega; do
    TINSEL_HIP_LIB=$PWD/scratch/libs/$lib timeout 300 python bench.py --steps 64 --warmup 4 --pipeline $p --no-cpu-baseline 2>/dev/null > /tmp/b.json
    python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('%-12s %-10s Msamples/s %7.1f Mrays/s %7.1f' % ('$lib', '$p', d['value'], d['mrays_per_s']), d['roofline']['kernel_ms'])
PY
  done
done
