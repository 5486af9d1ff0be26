#!/bin/bash
# round 4, call C: pair records, third version (no scratch: the hit normal is formed where the record is written); opt-in now
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4c; mkdir -p $O
( time timeout 600 python -m pytest tests/test_gpu_walk.py tests/test_gpu_refit.py tests/test_gpu_lbvh.py -x -q 2>&1 | tail -15 ) > $O/pytest_walk.log 2>&1; tail -4 $O/pytest_walk.log
( export TINSEL_HIP_WALK_PAIRS=1; time timeout 600 python -m pytest tests/test_gpu_walk.py tests/test_gpu_refit.py tests/test_gpu_lbvh.py tests/test_gpu_configs.py -x -q 2>&1 | tail -15 ) > $O/pytest_walk_pairs.log 2>&1; tail -4 $O/pytest_walk_pairs.log
( time timeout 600 python -m pytest tests/test_gpu_switches.py -q -k "WALK_ or defaults" 2>&1 | tail -15 ) > $O/pytest_switches.log 2>&1; tail -4 $O/pytest_switches.log
run() { timeout 120 python bench.py "$@" --no-cpu-baseline --no-pmc --no-second-config --no-api --no-fast --no-ubench 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('| $TAG | %s | %.1f | %s |' % (d['config']['workload'].split(',')[0], d['value'], d['roofline']['kernel_ms']), flush=True)
PY
}
ab() { local S="$1"; shift; ( [ "$S" != "-" ] && export $S; TAG="$S" run "$@" ); }
CFG3="--scene large/ajax_standin --width 1920 --height 1080 --steps 20 --warmup 2"
GLASS="--scene glass --width 1920 --height 1080 --maxdepth 12 --steps 20 --warmup 2"
( echo "| environment | config | Msamples/s | kernel ms of one timed block |"; echo "|---|---|---|---|"
for S in "-" "TINSEL_HIP_WALK_PAIRS=1" "-" "TINSEL_HIP_WALK_PAIRS=1" "TINSEL_HIP_WALK_SINGLE=0"; do ab "$S" $CFG3; done
for L in 12 16; do ab "TINSEL_HIP_WALK_PAIRS=1 TINSEL_HIP_WALK_LEAFMIN=$L" $CFG3; done
for S in "-" "TINSEL_HIP_WALK_PAIRS=1" "-" "TINSEL_HIP_WALK_PAIRS=1"; do ab "$S" $GLASS; done
) 2>&1 | tee $O/ab_walk_pairs3.md
for S in "-" "TINSEL_HIP_WALK_PAIRS=1"; do
  ( [ "$S" != "-" ] && export $S; echo "== $S"; TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/libtinsel_hip_walkprof.so timeout 120 python scratch/walk_prof.py large/ajax_standin 1920 1080 4 20 2>&1 | tail -2 )
done 2>&1 | tee $O/walk_profile.txt
