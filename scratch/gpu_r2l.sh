#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2l; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_walk.py tests/test_gpu_configs.py tests/test_gpu_parity.py -x -q 2>&1 | grep -E "passed|failed" ) 
A="--scene large/ajax_standin --width 1920 --height 1080 --steps 32 --warmup 2 --no-cpu-baseline --no-pmc --no-fast"
run() { # tag, env...
  tag=$1; shift
  ( env "$@" timeout 300 python bench.py $A > $O/b_$tag.json 2> $O/b_$tag.err ) ; python - <<PY
import json
try:
    d = json.load(open("$O/b_$tag.json"))
    print("%-14s %8.1f Msamples/s  %s" % ("$tag", d["value"], d["roofline"]["kernel_ms"]))
except Exception as e:
    print("$tag FAILED", e, open("$O/b_$tag.err").read()[-600:])
PY
}
run base TINSEL_HIP_NO_WALK=1
run walk X=1
run g2 TINSEL_HIP_WALK_GRID_MULT=2
run g4 TINSEL_HIP_WALK_GRID_MULT=4
run refill8 TINSEL_HIP_WALK_REFILL=8
run refill32 TINSEL_HIP_WALK_REFILL=32
run top0 TINSEL_HIP_WALK_TOP=0
run b256 TINSEL_HIP_WALK_BLOCK=256 TINSEL_HIP_WALK_GRID_MULT=8
A="--scene ajax_standin_96 --width 1920 --height 1080 --steps 32 --warmup 2 --no-cpu-baseline --no-pmc --no-fast"
run s96_base TINSEL_HIP_NO_WALK=1
run s96_walk X=1
A="--scene glass --width 1920 --height 1080 --steps 32 --warmup 2 --no-cpu-baseline --no-pmc --no-fast"
run glass X=1
run glass_walkall TINSEL_HIP_WALK_MIN_TRIS=0
