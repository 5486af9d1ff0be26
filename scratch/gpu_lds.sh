#!/bin/bash
run() { timeout 600 python bench.py "$@" --no-cpu-baseline 2>/tmp/err.txt > /tmp/b.json || { tail -3 /tmp/err.txt; return; }
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('limit ${TINSEL_HIP_ARENA_LDS_LIMIT:-32768} %-58s Msamples/s %7.1f' % (d['config']['workload'][-58:], d['value']), d['roofline']['kernel_ms'])
PY
}
for lim in 32768 65536; do
export TINSEL_HIP_ARENA_LDS_LIMIT=$lim
for p in wavefront split auto; do
run --scene many_spheres --width 1024 --height 768 --steps 32 --warmup 2 --pipeline $p
done
done
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "many" 2>&1 | tail -2
