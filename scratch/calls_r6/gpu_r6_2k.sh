#!/bin/bash
# call 2k: what k_bounce's waiting wave cycles wait for -- instruction mix and wait counters by kind (cornell 1024^2, 20 passes), in passes of <= 8 counters
O=gpurun_out/r6_2k; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "SQ_[A-Z0-9_]+" | sort -u > $GRAFT_REPO_ROOT/$O/sq_counters_available.txt
run() { tag=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $GRAFT_REPO_ROOT/$O/raw_$tag -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --inner-pmc --no-ubench --scene cornell --width 1024 --height 1024 --maxdepth 4 --steps 20 > /dev/null 2> $GRAFT_REPO_ROOT/$O/err_$tag.txt
  CC=$(find $GRAFT_REPO_ROOT/$O/raw_$tag -name "*counter_collection.csv" | head -1)
  [ -n "$CC" ] && python $GRAFT_REPO_ROOT/scratch/pmc_raw.py $CC k_bounce >> $GRAFT_REPO_ROOT/$O/k_bounce_counters.txt
  rm -rf $GRAFT_REPO_ROOT/$O/raw_$tag; }
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run b SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
run c SQ_WAVE_CYCLES SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_BRANCH SQ_INSTS_FLAT
run d SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_IFETCH
cat $GRAFT_REPO_ROOT/$O/k_bounce_counters.txt; wc -l $GRAFT_REPO_ROOT/$O/sq_counters_available.txt; tail -3 $GRAFT_REPO_ROOT/$O/err_d.txt
