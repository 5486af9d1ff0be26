#!/bin/bash
# round 4, call G: the whole GPU suite on the tree with the host-divided MIS constants and the plane-division pruning; A/B of the pruning
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4g; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -12 ) > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log
run() { timeout 120 python bench.py "$@" --no-cpu-baseline --no-pmc --no-second-config --no-api --no-fast --no-ubench 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('| $TAG | %s | %.1f | %s |' % (d['config']['workload'].split(',')[0], d['value'], d['roofline']['kernel_ms']), flush=True)
PY
}
ab() { local S="$1"; shift; ( [ "$S" != "-" ] && export $S; TAG="$S" run "$@" ); }
NOP="TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/libtinsel_hip_noprune.so"
( echo "| environment | config | Msamples/s | kernel ms of one timed block |"; echo "|---|---|---|---|"
for S in "$NOP" "-" "$NOP" "-"; do ab "$S" --scene cornell --steps 20 --warmup 5; done
for S in "$NOP" "-" "$NOP" "-"; do ab "$S" --scene veach --width 3840 --height 2160 --steps 8 --warmup 1; done
for S in "$NOP" "-"; do ab "$S" --scene glass --width 1920 --height 1080 --maxdepth 12 --steps 20 --warmup 2; done
for S in "$NOP" "-"; do ab "$S" --scene features --width 1920 --height 1080 --maxdepth 6 --steps 16 --warmup 1; done
for S in "$NOP" "-"; do ab "$S" --scene cornell --width 256 --height 256 --steps 16 --warmup 4; done
for S in "$NOP" "-"; do ab "$S" --scene large/ajax_standin --width 1920 --height 1080 --steps 20 --warmup 2; done
for S in "$NOP" "-"; do ab "$S" --scene gloss --steps 64 --warmup 8; done
) 2>&1 | sed "s#$GRAFT_REPO_ROOT/##" | tee $O/ab_plane_prune.md
