#!/bin/bash
# round 5, call h: the driver's bench line with the kernel events in the LAST timed block, rocprofv3 --stats of the same command, smoke()
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5h; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
( time timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real
cp bench_detail.json $O/bench_detail_default.json; wc -c $O/bench_default.json; cat $O/bench_default.json
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O -o stats -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-fast --no-api --no-ubench > $O/bench_under_stats.json 2> $O/stats.err
cd $GRAFT_REPO_ROOT
python scratch/rocprof_summary.py $(ls $O/*stats*.db 2>/dev/null | head -1) > $O/kernel_stats.md 2>&1; head -12 $O/kernel_stats.md
find $O -name "*.csv" -size +4M -delete; find $O -name "*.db" -size +20M -delete
