#!/bin/bash
# round 4, call S: k_bounce's late kernel-argument loads: camera + sky (latesky) against camera + sky + state pointers (latestate); the in-tree build has none
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4s; mkdir -p $O
A="TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/libtinsel_hip_latesky.so"
B="TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/libtinsel_hip_latestate.so"
( time env $B timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_probe.py -m gpu -q -x 2>&1 | tail -6 ) > $O/pytest.log 2>&1; grep -a "passed\|failed" $O/pytest.log
run() { timeout 120 python bench.py "$@" --no-cpu-baseline --no-pmc --no-second-config --no-api --no-fast --no-ubench 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('| $TAG | %s | %.1f | %s |' % (d['config']['workload'].split(',')[0], d['value'], d['roofline']['kernel_ms']), flush=True)
PY
}
ab() { local S="$1"; shift; ( [ "$S" != "-" ] && export $S; TAG="$S" run "$@" ); }
( echo "| environment | config | Msamples/s | kernel busy ms of one timed block |"; echo "|---|---|---|---|"
for S in "$A" "$B" "$A" "$B"; do ab "$S" --scene cornell --steps 20 --warmup 5; done
for S in "$A" "$B"; do ab "$S" --scene cornell --width 256 --height 256 --steps 16 --warmup 4; done
for S in "$A" "$B" "$A" "$B"; do ab "$S" --scene veach --width 3840 --height 2160 --steps 8 --warmup 1; done
for S in "$A" "$B"; do ab "$S" --scene features --width 1920 --height 1080 --maxdepth 6 --steps 16 --warmup 1; done
for S in "$A" "$B"; do ab "$S" --scene gloss --steps 64 --warmup 8; done
for S in "$A" "$B"; do ab "$S" --scene large/env_loft --width 1024 --height 512 --steps 64 --warmup 2; done
) 2>&1 | sed "s#$GRAFT_REPO_ROOT/##" | tee $O/ab_late_state.md
