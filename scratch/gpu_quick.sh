#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
for i in 1 2; do
timeout 300 python bench.py --steps 128 --warmup 8 --no-cpu-baseline 2>/dev/null > /tmp/b.json
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('Msamples/s %7.1f' % d['value'], d['roofline']['kernel_ms'])
PY
done
