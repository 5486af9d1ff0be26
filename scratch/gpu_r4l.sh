#!/bin/bash
# round 4, call L: k_shade tracing the shadow rays itself (TINSEL_HIP_SHADOW_IN_SHADE): bit-equality, then A/B rates
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4l; mkdir -p $O
( time timeout 600 python -m pytest tests/test_gpu_switches.py -k "SHADOW_IN_SHADE" -q -x 2>&1 | tail -8 ) > $O/pytest.log 2>&1; tail -5 $O/pytest.log
( time TINSEL_HIP_SHADOW_IN_SHADE=1 timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_walk.py tests/test_gpu_reference_scenes.py tests/test_gpu_parity.py -q -x 2>&1 | tail -8 ) > $O/pytest_env.log 2>&1; tail -5 $O/pytest_env.log
run() { timeout 120 python bench.py "$@" --no-cpu-baseline --no-pmc --no-second-config --no-api --no-fast --no-ubench 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('| $TAG | %s | %.1f | %s |' % (d['config']['workload'].split(',')[0], d['value'], d['roofline']['kernel_ms']), flush=True)
PY
}
ab() { local S="$1"; shift; ( [ "$S" != "-" ] && export $S; TAG="$S" run "$@" ); }
ON="TINSEL_HIP_SHADOW_IN_SHADE=1"
( echo "| environment | config | Msamples/s | kernel busy ms of one timed block |"; echo "|---|---|---|---|"
for S in "-" "$ON" "-" "$ON"; do ab "$S" --scene large/ajax_standin --width 1920 --height 1080 --steps 20 --warmup 2; done
for S in "-" "$ON" "-" "$ON"; do ab "$S" --scene glass --width 1920 --height 1080 --maxdepth 12 --steps 20 --warmup 2; done
for S in "-" "$ON"; do ab "$S" --scene motionblur --width 1920 --height 1080 --steps 16 --warmup 2; done
for S in "-" "$ON"; do ab "$S" --scene large/table --width 1920 --height 1080 --steps 8 --warmup 1; done
for S in "-" "$ON"; do ab "$S" --scene large/transmission --width 1920 --height 1080 --steps 8 --warmup 1; done
) 2>&1 | sed "s#$GRAFT_REPO_ROOT/##" | tee $O/ab_shadow_in_shade.md
