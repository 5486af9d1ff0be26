#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_fuzz.py -m gpu -x -q 2>&1 | tail -3
run() { timeout 600 python bench.py "$@" --no-cpu-baseline 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('%-8s %-40s Msamples/s %7.1f' % ('${TINSEL_HIP_LIB:+HEAD}', d['config']['workload'][:40], d['value']), d['roofline']['kernel_ms'])
PY
}
for rep in 1 2; do
for lib in "" scratch/libs/lib_h.so; do
if [ -n "$lib" ]; then export TINSEL_HIP_LIB=$PWD/$lib; else unset TINSEL_HIP_LIB; fi
run --steps 128 --warmup 8
run --scene gloss --steps 64 --warmup 2
run --scene cornell_probe --steps 64 --warmup 2
run --scene large/env_loft --width 1024 --height 512 --steps 64 --warmup 2
done
done
