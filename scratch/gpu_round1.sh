#!/bin/bash
# first measured run: parity tests, bench, rocprof kernel trace
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/pytest_gpu.log
cat gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 64 --warmup 4 > gpurun_out/bench_64.json 2> gpurun_out/bench_64.err
tail -3 gpurun_out/bench_64.err; cat gpurun_out/bench_64.json
timeout 600 python bench.py --steps 64 --warmup 4 --pipeline mega --no-cpu-baseline > gpurun_out/bench_mega_64.json 2>> gpurun_out/bench_64.err
cat gpurun_out/bench_mega_64.json
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 32 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof/bench_under_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof/rocprof.err
cd $GRAFT_REPO_ROOT; ls -R gpurun_out/prof | head -30
