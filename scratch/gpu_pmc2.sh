#!/bin/bash
mkdir -p gpurun_out/pmc2; export TMPDIR=/tmp; cd /tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pmc2
PIPE=${1:-wavefront}
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM" \
           "SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_LEVEL_WAVES SQ_ACCUM_PREV_HIRES"; do
  tag=$(echo $set | md5sum | cut -c1-6)
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $O -o pmc_${PIPE}_$tag --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 0 --pipeline $PIPE --no-cpu-baseline > /dev/null 2> $O/err_$tag.txt
done
ls $O | head -20
