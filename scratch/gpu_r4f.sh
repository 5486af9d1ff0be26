#!/bin/bash
# round 4, call F: the medium index in the path state (no 16-B absorption record), k_shade_sorted chosen per scene, the reference's
# motionblur.tin as a fixture, SURVEY 8f rank 4 measured (scratch/bench_f4.py)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4f; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_fuzz.py tests/test_gpu_walk.py tests/test_gpu_swalk.py tests/test_gpu_roulette.py tests/test_gpu_configs.py tests/test_gpu_rebuild.py -x -q 2>&1 | tail -8 ) > $O/pytest_a.log 2>&1; tail -5 $O/pytest_a.log
( time timeout 900 python -m pytest tests/test_gpu_switches.py -x -q -k "SHADE_SORTED or REPACK or defaults or BATCH_PATHS or NO_LDS" 2>&1 | tail -8 ) > $O/pytest_b.log 2>&1; tail -4 $O/pytest_b.log
timeout 600 python scratch/bench_f4.py > $O/f4.md 2> $O/f4.err; cat $O/f4.md; tail -3 $O/f4.err
run() { timeout 120 python bench.py "$@" --no-cpu-baseline --no-pmc --no-second-config --no-api --no-fast --no-ubench 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('| $TAG | %s | %.1f | %s |' % (d['config']['workload'].split(',')[0], d['value'], d['roofline']['kernel_ms']), flush=True)
PY
}
ab() { local S="$1"; shift; ( [ "$S" != "-" ] && export $S; TAG="$S" run "$@" ); }
( echo "| environment | config | Msamples/s | kernel ms of one timed block |"; echo "|---|---|---|---|"
ab "-" --scene glass --width 1920 --height 1080 --maxdepth 12 --steps 20 --warmup 2
ab "-" --scene glass --width 1920 --height 1080 --maxdepth 12 --steps 20 --warmup 2
ab "-" --scene large/ajax_standin --width 1920 --height 1080 --steps 20 --warmup 2
ab "-" --scene many_spheres --width 1024 --height 768 --steps 64 --warmup 2
ab "TINSEL_HIP_SHADE_SORTED=0" --scene many_spheres --width 1024 --height 768 --steps 64 --warmup 2
ab "-" --scene features --width 1920 --height 1080 --maxdepth 6 --steps 16 --warmup 1
ab "-" --scene veach --width 3840 --height 2160 --steps 8 --warmup 1
ab "-" --scene veach --pipeline split --width 1920 --height 1080 --steps 16 --warmup 1
ab "TINSEL_HIP_SHADE_SORTED=0" --scene veach --pipeline split --width 1920 --height 1080 --steps 16 --warmup 1
ab "-" --scene motionblur --width 1920 --height 1080 --steps 16 --warmup 1
) 2>&1 | tee $O/rates.md
