#!/bin/bash
# round 3, call O: k_walk with 1 / 2 / 3 / 4 node visits per turn of its loop
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3o; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_walk.py tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_gpu_refit.py -m gpu -q --maxfail=10 -k "not full" 2>&1 | grep -E "passed|failed" ) | tee $OUT/pytest.log
run() { timeout 600 python bench.py "$@" --no-cpu-baseline --no-pmc --no-second-config --no-api --no-fast --no-ubench 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('| %s | %s | %.1f | %s |' % ("$TAG", d['config']['workload'].split(',')[0], d['value'], d['roofline']['kernel_ms']))
PY
}
for L in scratch/ab/libtinsel_hip_reps1.so tinsel_amd/libtinsel_hip.so scratch/ab/libtinsel_hip_reps3.so scratch/ab/libtinsel_hip_reps4.so; do export TAG=$(basename $L); export TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/$L
  run --scene large/ajax_standin --width 1920 --height 1080 --steps 32 --warmup 2
  run --scene glass --width 1920 --height 1080 --maxdepth 12 --steps 32 --warmup 2
done 2>&1 | tee $OUT/ab_node_reps.txt
