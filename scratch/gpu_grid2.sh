#!/bin/bash
run() { timeout 600 python bench.py "$@" --no-cpu-baseline 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('mult $TINSEL_HIP_GRID_MULT %-40s Msamples/s %7.1f' % (d['config']['workload'][:40], d['value']))
PY
}
for m in 32 64 128 256; do
export TINSEL_HIP_GRID_MULT=$m
run --scene glass --width 1920 --height 1080 --maxdepth 12 --steps 8 --warmup 1
run --scene veach --width 3840 --height 2160 --steps 4 --warmup 1
run --scene large/ajax_standin --width 1920 --height 1080 --steps 8 --warmup 1
run --scene features --steps 16 --warmup 1
done
