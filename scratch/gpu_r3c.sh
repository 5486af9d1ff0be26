#!/bin/bash
# round 3, call C: suite after the memset-ordering fix; sweeps: k_walk refill / leaf thresholds at 8 waves per SIMD, region length for
# 1 M-path batches, golden-step group order vs index order in the one-launch k_bounce
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/r3c; mkdir -p $OUT
( time timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 2>&1 | tail -40 ) > $OUT/pytest_gpu.log 2>&1
tail -8 $OUT/pytest_gpu.log
for i in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_group.py -m gpu -q 2>&1 | tail -2; done | tee $OUT/pytest_group_x3.log
run() { timeout 600 python bench.py "$@" --no-cpu-baseline --no-pmc --no-second-config --no-api --no-fast --no-ubench 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('| %s | %s | %.1f | %s |' % ("$TAG", d['config']['workload'].split(',')[0], d['value'], d['roofline']['kernel_ms']))
PY
}
echo "=== k_walk thresholds"
for RF in 8 16 24 32 48; do for LM in 4 8 16; do export TAG="refill $RF leafmin $LM"; export TINSEL_HIP_WALK_REFILL=$RF TINSEL_HIP_WALK_LEAFMIN=$LM
  run --scene large/ajax_standin --width 1920 --height 1080 --steps 32 --warmup 2
done; done 2>&1 | tee $OUT/ab_walk_thresholds.txt
unset TINSEL_HIP_WALK_REFILL TINSEL_HIP_WALK_LEAFMIN
for GM in 2 3; do export TAG="walk grid mult $GM"; export TINSEL_HIP_WALK_GRID_MULT=$GM
  run --scene large/ajax_standin --width 1920 --height 1080 --steps 32 --warmup 2
done 2>&1 | tee -a $OUT/ab_walk_thresholds.txt
unset TINSEL_HIP_WALK_GRID_MULT
echo "=== region length, 1 M-path batches (cfg1) and 20 M (the driver's cornell)"
for RL in 128 256 512 1024 2048; do export TAG="region len $RL"; export TINSEL_HIP_REGION_LEN=$RL
  run --scene cornell --width 256 --height 256 --steps 16 --warmup 4
  run --scene cornell --steps 20 --warmup 5
done 2>&1 | tee $OUT/ab_region_len.txt
unset TINSEL_HIP_REGION_LEN
echo "=== group order in the one-launch k_bounce: golden step (default) vs index order"
for ST in 0 1; do export TAG="group step $ST"; [ $ST = 1 ] && export TINSEL_HIP_BOUNCE_GROUP_STEP=1 || unset TINSEL_HIP_BOUNCE_GROUP_STEP
  run --scene large/env_loft --width 1024 --height 512 --steps 64 --warmup 2
  run --scene cornell --steps 20 --warmup 5
  run --scene veach --width 3840 --height 2160 --steps 8 --warmup 1
  run --scene gloss --steps 64 --warmup 8
  run --scene features --width 1920 --height 1080 --maxdepth 6 --steps 16 --warmup 1
done 2>&1 | tee $OUT/ab_group_step.txt
unset TINSEL_HIP_BOUNCE_GROUP_STEP
echo "=== all configs, defaults"
export TAG=default
bash scratch/gpu_ab_all.sh tinsel_amd/libtinsel_hip.so 2>&1 | tee $OUT/all_configs.txt
python - <<'PY'
import tinsel_amd
from tinsel_amd import renderer as R
ms, u = tinsel_amd.ubench(R.UBENCH_COPY, 1 << 30)
print("stream copy best shape: %.1f GB/s" % (u/(ms*1e-3)/1e9))
PY
