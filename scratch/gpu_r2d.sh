#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2d; mkdir -p $O
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
    print("$tag FAILED", e, open("$O/ajax_$tag.err").read()[-600:])
PY
}
run base TINSEL_HIP_NO_WALK=1
run walk X=1
run top0 TINSEL_HIP_WALK_TOP=0
run top256 TINSEL_HIP_WALK_TOP=256
run top512 TINSEL_HIP_WALK_TOP=512
run b256 TINSEL_HIP_WALK_BLOCK=256
run leaf1 TINSEL_HIP_WALK_LEAFMIN=1
run leaf16 TINSEL_HIP_WALK_LEAFMIN=16
run refill8 TINSEL_HIP_WALK_REFILL=8
run refill32 TINSEL_HIP_WALK_REFILL=32
run grid2 TINSEL_HIP_WALK_GRID_MULT=2
run grid4 TINSEL_HIP_WALK_GRID_MULT=4
run grid16 TINSEL_HIP_WALK_GRID_MULT=16
run grid32 TINSEL_HIP_WALK_GRID_MULT=32
A="--scene glass --width 1920 --height 1080 --steps 32 --warmup 2 --no-cpu-baseline"
run glass_base TINSEL_HIP_NO_WALK=1
run glass_walk X=1
export TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/libtinsel_hip_walkprof.so
python scratch/walk_prof.py large/ajax_standin 1920 1080 4 32 2>&1 | grep -v amdgpu.ids
python scratch/walk_prof.py glass 1920 1080 12 32 2>&1 | grep -v amdgpu.ids
