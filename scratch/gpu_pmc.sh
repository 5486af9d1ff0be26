#!/bin/bash
# PMC pass (counters only: no sys/hip traces) over a short split-pipeline run
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/pmc/counters_list.txt 2>&1
PIPE=${1:-split}
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD" \
           "SQ_INSTS_VALU_MFMA_F64 SQ_IFETCH SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM"; do
  tag=$(echo $set | md5sum | cut -c1-6)
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $GRAFT_REPO_ROOT/gpurun_out/pmc -o pmc_${PIPE}_$tag --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 0 --pipeline $PIPE --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/pmc/err_$tag.txt
  tail -2 $GRAFT_REPO_ROOT/gpurun_out/pmc/err_$tag.txt
done
ls -la $GRAFT_REPO_ROOT/gpurun_out/pmc
