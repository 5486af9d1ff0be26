#!/bin/bash
# round 4, call V: sinf / cosf / expf coefficients materialised where they are used (K64): the in-tree build (-) against the build that keeps
# them pinned in registers (pinned), and with it k_shade at four waves per SIMD (shade4); parity of both new builds
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4v; mkdir -p $O
OLD="TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/libtinsel_hip_pinned.so"
S4="TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/libtinsel_hip_shade4.so"
( timeout 600 python -m pytest tests/test_gpu_leaf.py tests/test_gpu_parity.py tests/test_gpu_probe.py -m gpu -q -x 2>&1 | tail -3 ) > $O/pytest.log 2>&1; grep -a "passed\|failed" $O/pytest.log
( env $S4 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_scenes.py tests/test_gpu_walk.py -m gpu -q -x 2>&1 | tail -3 ) > $O/pytest_shade4.log 2>&1; grep -a "passed\|failed" $O/pytest_shade4.log
run() { timeout 120 python bench.py "$@" --no-cpu-baseline --no-pmc --no-second-config --no-api --no-fast --no-ubench 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('| $TAG | %s | %.1f | %s |' % (d['config']['workload'].split(',')[0], d['value'], d['roofline']['kernel_ms']), flush=True)
PY
}
ab() { local S="$1"; shift; ( [ "$S" != "-" ] && export $S; TAG="$S" run "$@" ); }
( echo "| environment | config | Msamples/s | kernel busy ms of one timed block |"; echo "|---|---|---|---|"
for S in "$OLD" "-" "$OLD" "-"; do ab "$S" --scene cornell --steps 20 --warmup 5; done
for S in "$OLD" "-"; do ab "$S" --scene cornell --width 256 --height 256 --steps 16 --warmup 4; done
for S in "$OLD" "-"; do ab "$S" --scene veach --width 3840 --height 2160 --steps 8 --warmup 1; done
for S in "$OLD" "-"; do ab "$S" --scene features --width 1920 --height 1080 --maxdepth 6 --steps 16 --warmup 1; done
for S in "$OLD" "-"; do ab "$S" --scene gloss --steps 64 --warmup 8; done
for S in "$OLD" "-" "$S4" "$OLD" "-" "$S4"; do ab "$S" --scene glass --width 1920 --height 1080 --maxdepth 12 --steps 20 --warmup 2; done
for S in "$OLD" "-" "$S4"; do ab "$S" --scene large/ajax_standin --width 1920 --height 1080 --steps 20 --warmup 2; done
for S in "$OLD" "-" "$S4"; do ab "$S" --scene many_spheres --width 1024 --height 768 --steps 64 --warmup 2; done
for S in "$OLD" "-" "$S4"; do ab "$S" --scene motionblur --width 1920 --height 1080 --steps 16 --warmup 2; done
) 2>&1 | sed "s#$GRAFT_REPO_ROOT/##" | tee $O/ab_k64.md
