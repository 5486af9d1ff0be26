#!/bin/bash
# A/B: scratch/libs/lib_*.so vs the in-tree library, interleaved
for rep in 1 2; do
for lib in "" $(ls scratch/libs/lib_[a-z].so 2>/dev/null); do
  if [ -n "$lib" ]; then export TINSEL_HIP_LIB=$PWD/$lib; else unset TINSEL_HIP_LIB; fi
  timeout 300 python bench.py --steps 128 --warmup 8 --no-cpu-baseline 2>/dev/null > /tmp/b.json
  python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('%-24s Msamples/s %7.1f' % ('${lib:-in-tree}', d['value']), d['roofline']['kernel_ms'])
PY
done
done
