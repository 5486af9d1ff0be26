#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -5
for b in 8388608 16777216; do
  TINSEL_HIP_BATCH_PATHS=$b timeout 300 python bench.py --steps 128 --warmup 8 --no-cpu-baseline 2>/dev/null > /tmp/b.json
  python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('batch %9d Msamples/s %7.1f' % ($b, d['value']), d['roofline']['kernel_ms'], 'launches', d['roofline']['launches'])
PY
done
