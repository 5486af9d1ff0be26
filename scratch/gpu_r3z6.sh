#!/bin/bash
# round 3, call Z6: the short regions at the end of a batch in EVERY streaming kernel (split pipeline too) and at the widest grids
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3z6; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_split.py tests/test_gpu_walk.py tests/test_gpu_swalk.py tests/test_fuzz.py tests/test_gpu_configs.py -m gpu -q -x 2>&1 | grep -aE "passed|failed" | tail -2 ) 2>&1 | tee $OUT/pytest_subset.log
run() { timeout 600 python bench.py "$@" --no-cpu-baseline --no-pmc --no-second-config --no-api --no-fast --no-ubench 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('| %s | %s | %.1f | %s |' % ("$TAG", d['config']['workload'].split(',')[0] + " x$STEPS", d['value'], d['roofline']['kernel_ms']))
PY
}
for rep in 1 2; do
for T in 0 on; do
  if [ $T = on ]; then unset TINSEL_HIP_TAIL_SPLIT; else export TINSEL_HIP_TAIL_SPLIT=$T; fi
  export TAG="TAIL_SPLIT=$T"
  STEPS=32 run --scene large/ajax_standin --width 1920 --height 1080 --steps 32 --warmup 2
  STEPS=20 run --scene large/ajax_standin --width 1920 --height 1080 --steps 20 --warmup 2
  STEPS=32 run --scene glass --width 1920 --height 1080 --maxdepth 12 --steps 32 --warmup 2
  STEPS=64 run --scene many_spheres --width 1024 --height 768 --steps 64 --warmup 2
  STEPS=64 run --scene cornell --steps 64 --warmup 5
  STEPS=8 run --scene veach --width 3840 --height 2160 --steps 8 --warmup 1
  STEPS=64 run --scene gloss --steps 64 --warmup 8
done; done 2>&1 | tee $OUT/ab_tail_split_all.txt
