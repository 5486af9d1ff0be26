#!/bin/bash
# round 4, call U: atan2f's infinity cases in fp32 (no double-precision atan2 constants parked in scratch): the in-tree build (-) against the
# build before it (atan2d), and with it k_shade / k_bounce at four waves per SIMD (shade4 / bounce4)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4u; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_leaf.py tests/test_gpu_probe.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -6 ) > $O/pytest.log 2>&1; grep -a "passed\|failed" $O/pytest.log
run() { timeout 120 python bench.py "$@" --no-cpu-baseline --no-pmc --no-second-config --no-api --no-fast --no-ubench 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('| $TAG | %s | %.1f | %s |' % (d['config']['workload'].split(',')[0], d['value'], d['roofline']['kernel_ms']), flush=True)
PY
}
ab() { local S="$1"; shift; ( [ "$S" != "-" ] && export $S; TAG="$S" run "$@" ); }
OLD="TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/libtinsel_hip_atan2d.so"
S4="TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/libtinsel_hip_shade4.so"
B4="TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/libtinsel_hip_bounce4.so"
( echo "| environment | config | Msamples/s | kernel busy ms of one timed block |"; echo "|---|---|---|---|"
for S in "$OLD" "-" "$B4" "$OLD" "-" "$B4"; do ab "$S" --scene cornell --steps 20 --warmup 5; done
for S in "$OLD" "-" "$B4"; do ab "$S" --scene veach --width 3840 --height 2160 --steps 8 --warmup 1; done
for S in "$OLD" "-" "$B4"; do ab "$S" --scene large/env_loft --width 1024 --height 512 --steps 64 --warmup 2; done
for S in "$OLD" "-" "$S4" "$OLD" "-" "$S4"; do ab "$S" --scene glass --width 1920 --height 1080 --maxdepth 12 --steps 20 --warmup 2; done
for S in "$OLD" "-" "$S4"; do ab "$S" --scene large/ajax_standin --width 1920 --height 1080 --steps 20 --warmup 2; done
for S in "$OLD" "-" "$S4"; do ab "$S" --scene many_spheres --width 1024 --height 768 --steps 64 --warmup 2; done
) 2>&1 | sed "s#$GRAFT_REPO_ROOT/##" | tee $O/ab_atan2.md
