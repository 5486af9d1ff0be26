#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2f; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/pytest.log
tail -4 $O/pytest.log
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
A="--scene ajax_standin_96 --width 1920 --height 1080 --steps 32 --warmup 2 --no-cpu-baseline"
run s96_base TINSEL_HIP_NO_WALK=1
run s96_walk X=1
A="--scene large/ajax_standin --width 1920 --height 1080 --steps 64 --warmup 2 --no-cpu-baseline"
run ajax_base TINSEL_HIP_NO_WALK=1
run ajax_walk X=1
A="--scene glass --width 1920 --height 1080 --steps 32 --warmup 2 --no-cpu-baseline"
run glass X=1
run glass_walkall TINSEL_HIP_WALK_MIN_TRIS=0
A="--scene veach --width 3840 --height 2160 --steps 8 --warmup 2 --no-cpu-baseline"
run veach X=1
A="--steps 64 --warmup 8 --no-cpu-baseline"
run cornell X=1
