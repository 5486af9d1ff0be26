#!/bin/bash
run() { timeout 600 python bench.py "$@" --no-cpu-baseline 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('batch $TINSEL_HIP_BATCH_PATHS %-40s Msamples/s %7.1f' % (d['config']['workload'][:40], d['value']), d['roofline']['kernel_ms'])
PY
}
for b in 4194304 8388608 16777216 33554432; do
export TINSEL_HIP_BATCH_PATHS=$b
run --scene glass --width 1920 --height 1080 --maxdepth 12 --steps 16 --warmup 1
run --scene veach --width 3840 --height 2160 --steps 8 --warmup 1
run --scene large/ajax_standin --width 1920 --height 1080 --steps 16 --warmup 1
done
