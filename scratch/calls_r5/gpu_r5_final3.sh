#!/bin/bash
# round 5, last call: the GPU suite and the driver's bench command on the final tree (after the accumulate-kernel test was added)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5final3; mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -q -s 2>&1 | grep -aE "passed|failed|per-pixel L2|rebuilt on the device" ) > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
( time timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real
cp bench_detail.json $O/bench_detail_default.json; wc -c $O/bench_default.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5final3/bench_default.json'))
print('headline', d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['cpu_baseline']['value'])
for c in d.get('configs', []): print(c['workload'][:40], c.get('value'))
PY
