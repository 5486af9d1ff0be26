#!/bin/bash
# round 2, call A: GPU tests with k_walk, cfg3/cfg4 A/B sweep of k_walk's knobs, VALU counters on cornell
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2a; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > $O/pytest.log
tail -3 $O/pytest.log
A="--scene large/ajax_standin --width 1920 --height 1080 --steps 32 --warmup 2 --no-cpu-baseline"
run() { # tag, env...
  tag=$1; shift
  ( env "$@" timeout 300 python bench.py $A > $O/ajax_$tag.json 2> $O/ajax_$tag.err ) ; python - <<PY
import json
try:
    d = json.load(open("$O/ajax_$tag.json"))
    print("%-14s %8.1f Msamples/s  %s" % ("$tag", d["value"], d["roofline"]["kernel_ms"]))
except Exception as e:
    print("$tag FAILED", e)
PY
}
run base TINSEL_HIP_NO_WALK=1
run walk X=1
run waves4 TINSEL_HIP_WALK_WAVES=4
run waves8 TINSEL_HIP_WALK_WAVES=8
run refill1 TINSEL_HIP_WALK_REFILL=1
run refill8 TINSEL_HIP_WALK_REFILL=8
run refill32 TINSEL_HIP_WALK_REFILL=32
run refill64 TINSEL_HIP_WALK_REFILL=64
run leaf8 TINSEL_HIP_WALK_LEAFMIN=8
run leaf16 TINSEL_HIP_WALK_LEAFMIN=16
run leaf32 TINSEL_HIP_WALK_LEAFMIN=32
run grid8 TINSEL_HIP_WALK_GRID_MULT=8
run grid64 TINSEL_HIP_WALK_GRID_MULT=64
run grid128 TINSEL_HIP_WALK_GRID_MULT=128
A="--scene glass --width 1920 --height 1080 --steps 32 --warmup 2 --no-cpu-baseline"
run glass_base TINSEL_HIP_NO_WALK=1
run glass_walk X=1
cd /tmp
rocprofv3 -L > $O/counters.txt 2>&1
B="$GRAFT_REPO_ROOT/bench.py --steps 32 --warmup 2 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $O -o pmc_cornell_a --output-format csv -- python $B > /dev/null 2> $O/pmc_cornell_a.err
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VALU SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE -d $O -o pmc_cornell_b --output-format csv -- python $B > /dev/null 2> $O/pmc_cornell_b.err
C="$GRAFT_REPO_ROOT/bench.py --scene large/ajax_standin --width 1920 --height 1080 --steps 8 --warmup 0 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $O -o pmc_ajax_a --output-format csv -- python $C > /dev/null 2> $O/pmc_ajax_a.err
cd $GRAFT_REPO_ROOT
for f in $O/*counter_collection.csv; do echo "== $f"; python scratch/pmc_summary.py $f | head -80; done > $O/pmc_summary.txt 2>&1
find $O -name "*.csv" -size +8M -delete
ls $O | head -50
