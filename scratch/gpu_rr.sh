#!/bin/bash
run() { timeout 600 python bench.py "$@" --no-cpu-baseline 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('rr %d %-40s Msamples/s %7.1f rays/sample %.2f' % (d['config']['russian_roulette_from_bounce'], d['config']['workload'][:40], d['value'], d['config']['rays_per_sample']))
PY
}
for rr in 0 3; do
run --scene glass --width 1920 --height 1080 --maxdepth 12 --steps 32 --warmup 1 --roulette $rr
run --steps 64 --warmup 4 --roulette $rr
run --scene large/ajax_standin --width 1920 --height 1080 --steps 32 --warmup 1 --roulette $rr
done
