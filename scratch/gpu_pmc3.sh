#!/bin/bash
mkdir -p gpurun_out/pmc3; export TMPDIR=/tmp; cd /tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pmc3
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH"; do
  tag=$(echo $set | md5sum | cut -c1-6)
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $O -o wf_$tag --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 0 --no-cpu-baseline > /dev/null 2> $O/err_wf_$tag.txt
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $O -o nrm_$tag --output-format csv -- python $GRAFT_REPO_ROOT/scratch/bench_normals.py cornell > /dev/null 2> $O/err_nrm_$tag.txt
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $O -o sp_$tag --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 0 --pipeline split --no-cpu-baseline > /dev/null 2> $O/err_sp_$tag.txt
done
ls $O | wc -l
