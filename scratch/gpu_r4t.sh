#!/bin/bash
# round 4, call T: k_shade's late kernel-argument loads (-DTN_LATE_SHADE=1): parity of the variant, A/B against the in-tree build
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4t; mkdir -p $O
NEW="TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/libtinsel_hip_lateshade.so"
( time env $NEW timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_probe.py tests/test_gpu_reference_scenes.py tests/test_gpu_split.py tests/test_gpu_walk.py tests/test_fuzz.py -m gpu -q -x 2>&1 | tail -6 ) > $O/pytest.log 2>&1; grep -a "passed\|failed" $O/pytest.log
run() { timeout 120 python bench.py "$@" --no-cpu-baseline --no-pmc --no-second-config --no-api --no-fast --no-ubench 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('| $TAG | %s | %.1f | %s |' % (d['config']['workload'].split(',')[0], d['value'], d['roofline']['kernel_ms']), flush=True)
PY
}
ab() { local S="$1"; shift; ( [ "$S" != "-" ] && export $S; TAG="$S" run "$@" ); }
( echo "| environment | config | Msamples/s | kernel busy ms of one timed block |"; echo "|---|---|---|---|"
for S in "-" "$NEW" "-" "$NEW"; do ab "$S" --scene glass --width 1920 --height 1080 --maxdepth 12 --steps 20 --warmup 2; done
for S in "-" "$NEW" "-" "$NEW"; do ab "$S" --scene large/ajax_standin --width 1920 --height 1080 --steps 20 --warmup 2; done
for S in "-" "$NEW"; do ab "$S" --scene motionblur --width 1920 --height 1080 --steps 16 --warmup 2; done
for S in "-" "$NEW"; do ab "$S" --scene large/table --width 1920 --height 1080 --steps 8 --warmup 1; done
for S in "-" "$NEW"; do ab "$S" --scene cornell --pipeline split --steps 20 --warmup 2; done
) 2>&1 | sed "s#$GRAFT_REPO_ROOT/##" | tee $O/ab_late_shade.md
