#!/bin/bash
run() { timeout 600 python bench.py "$@" --no-cpu-baseline 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('trace mult $TINSEL_HIP_GRID_MULT_TRACE %-34s Msamples/s %7.1f' % (d['config']['workload'][:34], d['value']), d['roofline']['kernel_ms'])
PY
}
for m in 16 32 64 128 256; do
export TINSEL_HIP_GRID_MULT_TRACE=$m
run --scene large/ajax_standin --width 1920 --height 1080 --steps 64 --warmup 2
run --scene glass --width 1920 --height 1080 --maxdepth 12 --steps 32 --warmup 1
done
