#!/bin/bash
# round 3, call G: k_shade_sorted (paths of a region shaded class by class): parity, A/B with per-kernel counters
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3g; mkdir -p $OUT
( TINSEL_HIP_SHADE_SORTED=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_walk.py tests/test_fuzz.py tests/test_gpu_swalk.py tests/test_gpu_split.py tests/test_gpu_roulette.py -m gpu -q --maxfail=10 2>&1 | tail -6 ) | tee $OUT/pytest_sorted.log
run() { timeout 600 python bench.py "$@" --no-cpu-baseline --no-pmc --no-second-config --no-api --no-fast --no-ubench 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('| %s | %s | %.1f | %s |' % ("$TAG", d['config']['workload'].split(',')[0], d['value'], d['roofline']['kernel_ms']))
PY
}
for S in 0 1; do export TINSEL_HIP_SHADE_SORTED=$S; export TAG="SHADE_SORTED=$S"
  run --scene glass --width 1920 --height 1080 --maxdepth 12 --steps 32 --warmup 2
  run --scene large/ajax_standin --width 1920 --height 1080 --steps 32 --warmup 2
  run --scene many_spheres --width 1024 --height 768 --steps 64 --warmup 2
  run --scene features --pipeline split --width 1920 --height 1080 --maxdepth 6 --steps 16 --warmup 1
  run --scene veach --pipeline split --width 3840 --height 2160 --steps 8 --warmup 1
done 2>&1 | tee $OUT/ab_shade_sorted.txt
for S in 0 1; do export TINSEL_HIP_SHADE_SORTED=$S
  bash scratch/gpu_pmc_kernels.sh $OUT sorted$S "glass 1920 1080 12 32" "large/ajax_standin 1920 1080 4 32" "many_spheres 1024 768 4 64" > /dev/null
done
cat $OUT/pmc_sorted0.md $OUT/pmc_sorted1.md | grep "###\|k_shade"
