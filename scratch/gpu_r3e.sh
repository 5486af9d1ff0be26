#!/bin/bash
# round 3, call E: k_swalk with 1024-thread workgroups and the arena in LDS vs the 256-thread generic variant vs k_extend / k_shadow
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/r3e; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_swalk.py tests/test_gpu_parity.py -m gpu -q --maxfail=10 2>&1 | tail -12 ) | tee $OUT/pytest_swalk.log
( TINSEL_HIP_SWALK_NO_LDS=1 timeout 900 python -m pytest tests/test_gpu_swalk.py -m gpu -q --maxfail=10 2>&1 | tail -4 ) | tee -a $OUT/pytest_swalk.log
run() { timeout 600 python bench.py "$@" --no-cpu-baseline --no-pmc --no-second-config --no-api --no-fast --no-ubench 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('| %s | %s | %.1f | %s |' % ("$TAG", d['config']['workload'].split(',')[0], d['value'], d['roofline']['kernel_ms']))
PY
}
MS="--scene many_spheres --width 1024 --height 768 --steps 64 --warmup 2"
export TAG="k_extend / k_shadow (TINSEL_HIP_NO_SCENE_WALK)"; TINSEL_HIP_NO_SCENE_WALK=1 run $MS | tee $OUT/ab_swalk.txt
export TAG="k_swalk 256 threads, generic pointers, 32 per CU, golden list"; TINSEL_HIP_SWALK_NO_LDS=1 run $MS | tee -a $OUT/ab_swalk.txt
export TAG="k_swalk 256 threads, generic pointers, 32 per CU, index-order list"; TINSEL_HIP_SWALK_NO_LDS=1 TINSEL_HIP_SWALK_LIST_STEP=1 run $MS | tee -a $OUT/ab_swalk.txt
export TAG="k_swalk 1024 threads, arena in LDS (default)"; run $MS | tee -a $OUT/ab_swalk.txt
for RF in 16 32 48; do for LM in 8 16 32; do export TAG="k_swalk LDS refill $RF leafmin $LM"; export TINSEL_HIP_SWALK_REFILL=$RF TINSEL_HIP_SWALK_LEAFMIN=$LM
  run $MS
done; done 2>&1 | tee -a $OUT/ab_swalk.txt
unset TINSEL_HIP_SWALK_REFILL TINSEL_HIP_SWALK_LEAFMIN
export TAG="k_swalk LDS, 2 workgroup ranges per CU"; TINSEL_HIP_SWALK_GRID_MULT=2 run $MS | tee -a $OUT/ab_swalk.txt
export TAG="k_swalk LDS, 4 workgroup ranges per CU"; TINSEL_HIP_SWALK_GRID_MULT=4 run $MS | tee -a $OUT/ab_swalk.txt
