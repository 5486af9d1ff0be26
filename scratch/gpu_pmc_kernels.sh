#!/bin/bash
# gpu_pmc_kernels.sh OUTDIR TAG "scene W H depth passes" ...   per-kernel SQ counters of one workload (lanes active, waiting, VALU issue,
# resident waves) as a markdown table; environment variables (A/B switches) are inherited; PMC_EXTRA = further bench.py arguments
OUTD=$1; TAG=$2; shift 2
cd /tmp; export TMPDIR=/tmp
mkdir -p $OUTD
for sc in "$@"; do set -- $sc
B=$(basename $1)
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY GRBM_GUI_ACTIVE -d $OUTD/raw_${TAG}_$B -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --inner-pmc --no-ubench --scene $1 --width $2 --height $3 --maxdepth $4 --steps $5 $PMC_EXTRA > /dev/null 2> $OUTD/pmc_${TAG}_$B.err
CC=$(find $OUTD/raw_${TAG}_$B -name "*counter_collection.csv" | head -1); KT=$(find $OUTD/raw_${TAG}_$B -name "*kernel_trace.csv" | head -1)
( echo "### $TAG: $1 ${2}x$3 maxDepth $4, $5 passes"; python $GRAFT_REPO_ROOT/scratch/pmc_table.py $CC $KT ) | tee -a $OUTD/pmc_$TAG.md
rm -rf $OUTD/raw_${TAG}_$B
done
