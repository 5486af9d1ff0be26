#!/bin/bash
# round 3, call N: k_accumulate_tiled with the window span as a compile-time constant
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3n; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_fuzz.py tests/test_gpu_group.py -m gpu -q --maxfail=10 -k "not full" 2>&1 | tail -4 ) | tee $OUT/pytest.log
run() { timeout 600 python bench.py "$@" --no-cpu-baseline --no-pmc --no-second-config --no-api --no-fast --no-ubench 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('| %s | %s | %.1f | %s |' % ("$TAG", d['config']['workload'].split(',')[0], d['value'], d['roofline']['kernel_ms']))
PY
}
for S in span nospan span nospan; do export TAG="accumulate: $S"; [ $S = nospan ] && export TINSEL_HIP_ACC_NO_SPAN=1 || unset TINSEL_HIP_ACC_NO_SPAN
  run --scene cornell --steps 20 --warmup 5
  run --scene cornell --width 256 --height 256 --steps 16 --warmup 4
  run --scene veach --width 3840 --height 2160 --steps 8 --warmup 1
  run --scene large/ajax_standin --width 1920 --height 1080 --steps 32 --warmup 2
done 2>&1 | tee $OUT/ab_acc_span.txt
