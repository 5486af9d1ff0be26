#!/bin/bash
# round 4, call E: scene-level rebuild (tests/test_gpu_rebuild.py), refit + shim + group (the arena grew: a Moving64 slot per primitive),
# many_spheres / cornell / features rates against call D's
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4e; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_rebuild.py -x -q -s 2>&1 | tail -25 ) > $O/pytest_rebuild.log 2>&1; tail -12 $O/pytest_rebuild.log
( time timeout 900 python -m pytest tests/test_gpu_refit.py tests/test_gpu_shim.py tests/test_gpu_group.py tests/test_gpu_leaf.py tests/test_gpu_parity.py -x -q 2>&1 | tail -8 ) > $O/pytest_more.log 2>&1; tail -5 $O/pytest_more.log
run() { timeout 120 python bench.py "$@" --no-cpu-baseline --no-pmc --no-second-config --no-api --no-fast --no-ubench 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('| $TAG | %s | %.1f | %s |' % (d['config']['workload'].split(',')[0], d['value'], d['roofline']['kernel_ms']), flush=True)
PY
}
ab() { local S="$1"; shift; ( [ "$S" != "-" ] && export $S; TAG="$S" run "$@" ); }
( echo "| environment | config | Msamples/s | kernel ms of one timed block |"; echo "|---|---|---|---|"
ab "-" --scene many_spheres --width 1024 --height 768 --steps 64 --warmup 2
ab "TINSEL_HIP_SHADE_SORTED=1" --scene many_spheres --width 1024 --height 768 --steps 64 --warmup 2
ab "-" --scene cornell --steps 20 --warmup 5
ab "-" --scene features --width 1920 --height 1080 --maxdepth 6 --steps 16 --warmup 1
ab "-" --scene gloss --steps 64 --warmup 8
ab "-" --scene large/env_loft --width 1024 --height 512 --steps 64 --warmup 2
) 2>&1 | tee $O/rates.md
