#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2b; mkdir -p $O
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
run chunks0 TINSEL_HIP_WALK_XCD_CHUNKS=0
run chunks1 TINSEL_HIP_WALK_XCD_CHUNKS=1
run chunks2 TINSEL_HIP_WALK_XCD_CHUNKS=2
run chunks4 TINSEL_HIP_WALK_XCD_CHUNKS=4
run chunks8 TINSEL_HIP_WALK_XCD_CHUNKS=8
run chunks16 TINSEL_HIP_WALK_XCD_CHUNKS=16
run c1_leaf8 TINSEL_HIP_WALK_LEAFMIN=8
run c1_leaf16 TINSEL_HIP_WALK_LEAFMIN=16
run c1_refill32 TINSEL_HIP_WALK_REFILL=32
run c1_refill64 TINSEL_HIP_WALK_REFILL=64
run c1_waves4 TINSEL_HIP_WALK_WAVES=4
run c1_grid64 TINSEL_HIP_WALK_GRID_MULT=64
run c1_grid16 TINSEL_HIP_WALK_GRID_MULT=16
A="--scene glass --width 1920 --height 1080 --steps 32 --warmup 2 --no-cpu-baseline"
run glass_base TINSEL_HIP_NO_WALK=1
run glass_walk X=1
