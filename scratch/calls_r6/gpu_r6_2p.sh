#!/bin/bash
# call 2p (second attempt: the first asked for more TA / TCP counters per pass than the hardware collects and sat out three timeouts): is k_walk bound by the vector L1's lookup rate (four 16-B pieces per 64-B node, each its own tag lookup)?  TA / TCP busy counters, config 3
O=gpurun_out/r6_2p; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(TA|TCP|TD|TCC)_[A-Z0-9_a-z\[\]]+" | sort -u > $GRAFT_REPO_ROOT/$O/mem_counters_available.txt
run() { tag=$1; shift
  timeout 75 rocprofv3 --kernel-trace --pmc "$@" -d $GRAFT_REPO_ROOT/$O/raw_$tag -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --inner-pmc --no-ubench --scene large/ajax_standin --width 1920 --height 1080 --maxdepth 4 --steps 20 > /dev/null 2> $GRAFT_REPO_ROOT/$O/err_$tag.txt
  CC=$(find $GRAFT_REPO_ROOT/$O/raw_$tag -name "*counter_collection.csv" | head -1)
  [ -n "$CC" ] && { python $GRAFT_REPO_ROOT/scratch/pmc_raw.py $CC k_walk; python $GRAFT_REPO_ROOT/scratch/pmc_raw.py $CC k_step; } >> $GRAFT_REPO_ROOT/$O/k_walk_counters.txt
  rm -rf $GRAFT_REPO_ROOT/$O/raw_$tag; }
run a GRBM_GUI_ACTIVE TA_TA_BUSY_sum
run b GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
run c GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
run d GRBM_GUI_ACTIVE TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum
cat $GRAFT_REPO_ROOT/$O/k_walk_counters.txt; wc -l $GRAFT_REPO_ROOT/$O/mem_counters_available.txt; grep -i "error\|invalid\|not" $GRAFT_REPO_ROOT/$O/err_a.txt | head -5
