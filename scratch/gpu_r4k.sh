#!/bin/bash
# round 4, call K: a batch's passes as two overlapped chunks on two streams (render_impl): bit-equality under the switch, then A/B rates
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4k; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_switches.py -k "OVERLAP or defaults" tests/test_gpu_split.py tests/test_gpu_parity.py -q -x 2>&1 | tail -8 ) > $O/pytest.log 2>&1; tail -5 $O/pytest.log
run() { timeout 120 python bench.py "$@" --no-cpu-baseline --no-pmc --no-second-config --no-api --no-fast --no-ubench 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('| $TAG | %s | %.1f | %s | %.2f |' % (d['config']['workload'].split(',')[0], d['value'], d['roofline']['kernel_ms'], d['roofline']['concurrent_launches']), flush=True)
PY
}
ab() { local S="$1"; shift; ( [ "$S" != "-" ] && export $S; TAG="$S" run "$@" ); }
OFF="TINSEL_HIP_OVERLAP=0"; ON="TINSEL_HIP_OVERLAP=1"
( echo "| environment | config | Msamples/s | kernel busy ms of one timed block | dominant kernel's launches at once |"; echo "|---|---|---|---|---|"
for S in "$OFF" "-" "$OFF" "-"; do ab "$S" --scene cornell --steps 20 --warmup 5; done
for S in "$OFF" "-"; do ab "$S" --scene cornell --steps 64 --warmup 5; done
for S in "$OFF" "-"; do ab "$S" --scene cornell --steps 256 --warmup 8; done
for S in "$OFF" "$ON"; do ab "$S" --scene cornell --steps 8 --warmup 5; done
for S in "$OFF" "$ON"; do ab "$S" --scene cornell --steps 4 --warmup 5; done
for S in "$OFF" "$ON"; do ab "$S" --scene cornell --width 256 --height 256 --steps 16 --warmup 4; done
for S in "$OFF" "-" "$OFF" "-"; do ab "$S" --scene veach --width 3840 --height 2160 --steps 8 --warmup 1; done
for S in "$OFF" "-"; do ab "$S" --scene features --width 1920 --height 1080 --maxdepth 6 --steps 16 --warmup 1; done
for S in "$OFF" "-"; do ab "$S" --scene gloss --steps 64 --warmup 8; done
for S in "$OFF" "$ON"; do ab "$S" --scene glass --width 1920 --height 1080 --maxdepth 12 --steps 20 --warmup 2; done
for S in "$OFF" "$ON"; do ab "$S" --scene large/ajax_standin --width 1920 --height 1080 --steps 20 --warmup 2; done
for S in "$OFF" "$ON"; do ab "$S" --scene many_spheres --width 1024 --height 768 --steps 64 --warmup 2; done
) 2>&1 | sed "s#$GRAFT_REPO_ROOT/##" | tee $O/ab_overlap.md
