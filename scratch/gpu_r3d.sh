#!/bin/bash
# round 3, call D: k_swalk (scene-level walk with ray replacement): parity with the flat scan off, A/B on many_spheres, thresholds
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/r3d; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_swalk.py tests/test_gpu_group.py -m gpu -q --maxfail=10 2>&1 | tail -30 ) | tee $OUT/pytest_swalk.log
run() { timeout 600 python bench.py "$@" --no-cpu-baseline --no-pmc --no-second-config --no-api --no-fast --no-ubench 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('| %s | %s | %.1f | %s |' % ("$TAG", d['config']['workload'].split(',')[0], d['value'], d['roofline']['kernel_ms']))
PY
}
export TAG="k_extend / k_shadow (TINSEL_HIP_NO_SCENE_WALK)"; TINSEL_HIP_NO_SCENE_WALK=1 run --scene many_spheres --width 1024 --height 768 --steps 64 --warmup 2 | tee $OUT/ab_swalk.txt
export TAG="k_swalk defaults"; run --scene many_spheres --width 1024 --height 768 --steps 64 --warmup 2 | tee -a $OUT/ab_swalk.txt
for RF in 8 16 32; do for LM in 4 16 32; do export TAG="k_swalk refill $RF leafmin $LM"; export TINSEL_HIP_SWALK_REFILL=$RF TINSEL_HIP_SWALK_LEAFMIN=$LM
  run --scene many_spheres --width 1024 --height 768 --steps 64 --warmup 2
done; done 2>&1 | tee -a $OUT/ab_swalk.txt
unset TINSEL_HIP_SWALK_REFILL TINSEL_HIP_SWALK_LEAFMIN
for GM in 4 16 32; do export TAG="k_swalk grid mult $GM"; export TINSEL_HIP_SWALK_GRID_MULT=$GM
  run --scene many_spheres --width 1024 --height 768 --steps 64 --warmup 2
done 2>&1 | tee -a $OUT/ab_swalk.txt
unset TINSEL_HIP_SWALK_GRID_MULT
export TAG="k_swalk list golden step"; TINSEL_HIP_SWALK_LIST_STEP=5011 run --scene many_spheres --width 1024 --height 768 --steps 64 --warmup 2 | tee -a $OUT/ab_swalk.txt
python - <<'PY'
import tinsel_amd
from tinsel_amd import renderer as R
ms, u = tinsel_amd.ubench(R.UBENCH_COPY, 1 << 30)
print("stream copy best shape: %.1f GB/s" % (u/(ms*1e-3)/1e9))
PY
