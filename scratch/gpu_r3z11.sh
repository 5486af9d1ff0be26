#!/bin/bash
# round 3, call Z11: one-set batches cut so that every CU gets the same work (split_one_set, the default now) against the uniform cut
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3z11; mkdir -p $OUT
( timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x 2>&1 | grep -aE "passed|failed" | tail -2 ) 2>&1 | tee $OUT/pytest_subset.log
run() { timeout 300 python bench.py "$@" --no-cpu-baseline --no-pmc --no-second-config --no-api --no-fast --no-ubench 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('| %s | %s | %.1f | %s |' % ("$TAG", d['config']['workload'].split(',')[0] + " x$STEPS", d['value'], d['roofline']['kernel_ms']))
PY
}
for T in 0 default 0 default; do
  if [ $T = default ]; then unset TINSEL_HIP_TAIL_SPLIT; else export TINSEL_HIP_TAIL_SPLIT=$T; fi
  export TAG="TAIL_SPLIT=$T"
  STEPS=16 run --scene cornell --width 256 --height 256 --steps 16 --warmup 4
  STEPS=4 run --scene cornell --width 512 --height 512 --steps 4 --warmup 4
  STEPS=16 run --scene veach --width 256 --height 256 --steps 16 --warmup 4
  STEPS=16 run --scene gloss --width 256 --height 256 --steps 16 --warmup 4
  STEPS=6 run --scene cornell --width 512 --height 512 --steps 6 --warmup 4
  STEPS=20 run --scene cornell --steps 20 --warmup 5
done 2>&1 | tee $OUT/ab_one_set.txt
