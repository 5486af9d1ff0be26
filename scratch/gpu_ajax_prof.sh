#!/bin/bash
mkdir -p gpurun_out/r1f_ajax
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r1f_ajax
A="--scene large/ajax_standin --width 1920 --height 1080"
timeout 900 python bench.py $A --steps 64 --warmup 2 > $O/bench_ajax.json 2> $O/bench_ajax.err; tail -2 $O/bench_ajax.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O -o stats -- python $GRAFT_REPO_ROOT/bench.py $A --steps 32 --warmup 2 --no-cpu-baseline > $O/bench_under_stats.json 2> $O/stats.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d $O -o pmc_$c --output-format csv -- python $GRAFT_REPO_ROOT/bench.py $A --steps 16 --warmup 0 --no-cpu-baseline > /dev/null 2> $O/pmc_$c.err
done
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU -d $O -o pmc_sq --output-format csv -- python $GRAFT_REPO_ROOT/bench.py $A --steps 16 --warmup 0 --no-cpu-baseline > /dev/null 2> $O/pmc_sq.err
cd $GRAFT_REPO_ROOT; ls $O | wc -l
