#!/bin/bash
# round 5, closing call: rocprofv3 --kernel-trace --stats of the driver's bench command on the final tree (the accumulate kernels changed after
# the table of gpu_r5_final2.sh), beside the same command's own line
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5final4; mkdir -p $O
( time timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real
cp bench_detail.json $O/bench_detail_default.json
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O -o stats -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-fast --no-api --no-ubench > $O/bench_under_stats.json 2> $O/stats.err
cd $GRAFT_REPO_ROOT
python scratch/rocprof_summary.py $(ls $O/*stats*.db 2>/dev/null | head -1) > $O/kernel_stats.md 2>&1
head -16 $O/kernel_stats.md
find $O -name "*.csv" -size +4M -delete; find $O -name "*.db" -size +20M -delete
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5final4/bench_default.json'))
print('headline', d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])
for c in d.get('configs', []): print(c['workload'][:40], c.get('value'))
PY
