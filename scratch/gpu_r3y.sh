#!/bin/bash
# round 3, call Y: small frames -- 512-thread accumulate workgroups (TINSEL_HIP_ACC_WIDE), the pass-seed table (no k_pass_seeds launch per call), share for short regions (default now)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3y; mkdir -p $OUT
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_group.py tests/test_gpu_display.py tests/test_gpu_arith.py -m gpu -q -x 2>&1 | tail -3 ) 2>&1 | tee $OUT/pytest_subset.log
run() { timeout 600 python bench.py "$@" --no-cpu-baseline --no-pmc --no-second-config --no-fast --no-ubench 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('| %s | %s | %.1f | %s | api %s / %s / %s |' % ("$TAG", d['config']['workload'].split(',')[0], d['value'], d['roofline']['kernel_ms'], d.get('api_1pass_plain_msamples_s'), d.get('api_1pass_msamples_s'), d.get('api_1pass_pinned_output_msamples_s')))
PY
}
for rep in 1 2; do
for W in 0 1; do export TINSEL_HIP_ACC_WIDE=$W; export TAG="ACC_WIDE=$W"
run --scene cornell --width 256 --height 256 --steps 16 --warmup 4 --no-api
run --scene cornell --width 512 --height 512 --steps 16 --warmup 4 --no-api
run --scene cornell --steps 20 --warmup 5 --no-api
run --scene gloss --width 256 --height 256 --steps 16 --warmup 4 --no-api
done; done 2>&1 | tee $OUT/ab_acc_wide.txt
unset TINSEL_HIP_ACC_WIDE; TAG=default run --scene cornell --steps 20 --warmup 5 2>&1 | tee $OUT/api.txt
