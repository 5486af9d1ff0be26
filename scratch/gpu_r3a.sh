#!/bin/bash
# round 3, call A: the GPU suite on the new code (refit at scene level, group look-ahead, eNormals shards, k_bounce's shading pools,
# whole-frame parity), A/B of the SLP-vectoriser-off build and of the pools, the default bench line with live calibration
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/r3a; mkdir -p $OUT
( time timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 -s 2>&1 | tail -60 ) > $OUT/pytest_gpu.log 2>&1
tail -25 $OUT/pytest_gpu.log
echo "=== parity subset on the noslp build"
( TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/libtinsel_hip_noslp.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_walk.py tests/test_fuzz.py tests/test_gpu_leaf.py tests/test_gpu_fast.py tests/test_gpu_split.py -m gpu -q --maxfail=5 2>&1 | tail -8 ) | tee $OUT/pytest_noslp.log
echo "=== A/B"
( bash scratch/gpu_ab_all.sh tinsel_amd/libtinsel_hip.so scratch/ab/libtinsel_hip_noslp.so
  echo "== no repack (default build, then noslp)"
  TINSEL_HIP_NO_REPACK=1 bash scratch/gpu_ab_all.sh tinsel_amd/libtinsel_hip.so scratch/ab/libtinsel_hip_noslp.so ) 2>&1 | tee $OUT/ab.txt
echo "=== default bench line"
( time timeout 1200 python bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2>&1 | tail -3
tail -5 $OUT/bench_default.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3a/bench_default.json'))
print('headline', d['value'], d['roofline']['frac'], d.get('yard_sticks'))
for c in d.get('configs', []):
    r=c.get('roofline') or {}
    print(c['config']['workload'], c.get('value'), r.get('kernel'), r.get('frac'), r.get('l2_hit_rate'), r.get('counter_calibration'), r.get('node_visits_G_s'), c.get('unavailable'))
print('api', d.get('pcie_inclusive_msamples_s'), d.get('api_1pass_plain_msamples_s'), d.get('api_1pass_msamples_s'), d.get('api_1pass_pinned_output_msamples_s'))
PY
