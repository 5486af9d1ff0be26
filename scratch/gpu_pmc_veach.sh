#!/bin/bash
mkdir -p gpurun_out/pmc_veach; export TMPDIR=/tmp; cd /tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_veach
for sc in veach features; do
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU -d $O -o $sc --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --scene $sc --steps 8 --warmup 0 --pipeline split --no-cpu-baseline > /dev/null 2> $O/err_$sc.txt
done
ls $O | head
