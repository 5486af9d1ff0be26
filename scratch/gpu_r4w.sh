#!/bin/bash
# round 4, call W (a lead for the next round, nothing adopted): k_extend's light-sampling variant at five waves per SIMD (-DTN_WAVES_EXTEND_LIGHTS=5)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4w; mkdir -p $O
NEW="TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/libtinsel_hip_extend5.so"
( env $NEW timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "glass or motionblur or ajax" 2>&1 | tail -2 ) > $O/pytest.log 2>&1; grep -a "passed\|failed" $O/pytest.log
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
for S in "-" "$NEW"; do ab "$S" --scene motionblur --width 1920 --height 1080 --steps 16 --warmup 2; done
for S in "-" "$NEW"; do ab "$S" --scene large/table --width 1920 --height 1080 --steps 8 --warmup 1; done
) 2>&1 | sed "s#$GRAFT_REPO_ROOT/##" | tee $O/ab_extend5.md
