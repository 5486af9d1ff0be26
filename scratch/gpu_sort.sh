#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_fuzz.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -3
run() { timeout 600 python bench.py "$@" --no-cpu-baseline 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('nosort=${TINSEL_HIP_NO_SORT_SHADE:-0} %-36s Msamples/s %7.1f' % (d['config']['workload'][:36], d['value']), d['roofline']['kernel_ms'])
PY
}
for rep in 1 2; do
for ns in "" 1; do
if [ -n "$ns" ]; then export TINSEL_HIP_NO_SORT_SHADE=1; else unset TINSEL_HIP_NO_SORT_SHADE; fi
run --scene veach --width 3840 --height 2160 --steps 8 --warmup 1
run --scene features --steps 32 --warmup 1
run --scene glass --width 1920 --height 1080 --maxdepth 12 --steps 32 --warmup 1
run --scene large/ajax_standin --width 1920 --height 1080 --steps 32 --warmup 1
done
done
