#!/bin/bash
mkdir -p gpurun_out/pmc_ajax; export TMPDIR=/tmp; cd /tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_ajax
export TINSEL_HIP_NO_BVH4=1
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH"; do
  tag=$(echo $set | md5sum | cut -c1-6)
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $O -o aj_$tag --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --scene large/ajax_standin --width 1920 --height 1080 --steps 4 --warmup 0 --no-cpu-baseline > /dev/null 2> $O/err_$tag.txt
done
ls $O | head
