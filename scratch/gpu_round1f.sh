#!/bin/bash
mkdir -p gpurun_out/r1f
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r1f
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -2 $O/bench_default.err; cat $O/bench_default.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O -o stats -- python $GRAFT_REPO_ROOT/bench.py --steps 64 --warmup 4 --no-cpu-baseline > $O/bench_under_stats.json 2> $O/stats.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d $O -o pmc_$c --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 16 --warmup 0 --no-cpu-baseline > /dev/null 2> $O/pmc_$c.err
done
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU -d $O -o pmc_sq --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 16 --warmup 0 --no-cpu-baseline > /dev/null 2> $O/pmc_sq.err
cd $GRAFT_REPO_ROOT; ls $O
