#!/bin/bash
# round 4, last call: the driver's bench line, rocprofv3 --stats of the same command and the parity files on the final tree (k_extend's
# light-sampling variant at five waves per SIMD is the only change since the full evidence run, profiles/r04_z_*)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4x; mkdir -p $O
( time timeout 200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_reference_scenes.py -m gpu -q -x 2>&1 | tail -3 ) > $O/pytest_parity.log 2>&1; grep -a "passed\|failed" $O/pytest_parity.log
( time timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $O -o stats -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-fast --no-api --no-ubench > $O/bench_under_stats.json 2> $O/stats.err
cd $GRAFT_REPO_ROOT
python scratch/rocprof_summary.py $(ls $O/*stats*.db 2>/dev/null | head -1) > $O/kernel_stats.md 2>&1
find $O -name "*.csv" -size +4M -delete; find $O -name "*.db" -size +20M -delete
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4x/bench_default.json'))
print('headline', d['value'], d['roofline']['kernel'], d['roofline']['frac'])
for c in d.get('configs', []):
    r=c.get('roofline') or {}
    print(c['config']['workload'][:40], c.get('value'), r.get('kernel'), r.get('frac'), r.get('kernel_ms'))
PY
