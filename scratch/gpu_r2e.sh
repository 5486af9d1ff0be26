#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2e; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q 2>&1 | tail -5 ) > $O/pytest.log
tail -2 $O/pytest.log
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
run walk X=1
run top0 TINSEL_HIP_WALK_TOP=0
run g2 TINSEL_HIP_WALK_GRID_MULT=2
run g2_leaf1 TINSEL_HIP_WALK_GRID_MULT=2 TINSEL_HIP_WALK_LEAFMIN=1
run g2_leaf16 TINSEL_HIP_WALK_GRID_MULT=2 TINSEL_HIP_WALK_LEAFMIN=16
run g2_refill8 TINSEL_HIP_WALK_GRID_MULT=2 TINSEL_HIP_WALK_REFILL=8
run g2_refill32 TINSEL_HIP_WALK_GRID_MULT=2 TINSEL_HIP_WALK_REFILL=32
run g1 TINSEL_HIP_WALK_GRID_MULT=1
run g3 TINSEL_HIP_WALK_GRID_MULT=3
run b256 TINSEL_HIP_WALK_BLOCK=256
run b256_g4 TINSEL_HIP_WALK_BLOCK=256 TINSEL_HIP_WALK_GRID_MULT=4
run b256_g16 TINSEL_HIP_WALK_BLOCK=256 TINSEL_HIP_WALK_GRID_MULT=16
A="--scene glass --width 1920 --height 1080 --steps 32 --warmup 2 --no-cpu-baseline"
run glass_walk X=1
run glass_g2 TINSEL_HIP_WALK_GRID_MULT=2
export TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/libtinsel_hip_walkprof.so
TINSEL_HIP_WALK_GRID_MULT=2 python scratch/walk_prof.py large/ajax_standin 1920 1080 4 32 2>&1 | grep -v amdgpu.ids
TINSEL_HIP_WALK_GRID_MULT=2 python scratch/walk_prof.py glass 1920 1080 12 32 2>&1 | grep -v amdgpu.ids
