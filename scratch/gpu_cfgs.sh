#!/bin/bash
run() { timeout 600 python bench.py "$@" --no-cpu-baseline 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
r=d['roofline']
print('%-44s Msamples/s %7.1f Mrays/s %8.1f I %.1f T %.2f P %.2f' % (d['config']['workload'][:44], d['value'], d['mrays_per_s'], r['I'], r['T'], r['P']), r['kernel_ms'], 'achieved GB/s %.0f frac %.2f (%s)' % (r['achieved'], r['frac'], r['kernel']))
PY
}
run --scene cornell --width 256 --height 256 --steps 16 --warmup 2
run --scene cornell --steps 256 --warmup 8
run --scene large/ajax_standin --width 1920 --height 1080 --steps 64 --warmup 2
run --scene glass --width 1920 --height 1080 --maxdepth 12 --steps 64 --warmup 2
run --scene veach --width 3840 --height 2160 --steps 16 --warmup 1
run --scene large/env_loft --width 1024 --height 512 --steps 64 --warmup 2
run --scene many_spheres --width 1024 --height 768 --steps 64 --warmup 2
