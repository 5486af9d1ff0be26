#!/bin/bash
run() { timeout 600 python bench.py "$@" --no-cpu-baseline 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('%-64s Msamples/s %7.1f' % (d['config']['workload'][:64], d['value']), d['roofline']['kernel_ms'])
PY
}
for p in wavefront split; do
run --scene veach --width 3840 --height 2160 --steps 8 --warmup 1 --pipeline $p
run --scene veach --width 1024 --height 1024 --steps 32 --warmup 1 --pipeline $p
run --scene features --steps 32 --warmup 1 --pipeline $p
run --scene features_probe --steps 32 --warmup 1 --pipeline $p
run --scene large/env_loft --width 1024 --height 512 --steps 64 --warmup 2 --pipeline $p
run --scene gloss --steps 32 --warmup 1 --pipeline $p
run --scene cornell_probe --steps 32 --warmup 1 --pipeline $p
done
