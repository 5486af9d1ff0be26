#!/bin/bash
# usage: gpu_sweep.sh VAR "v1 v2 ..." [pytest]  -- one bench line per split-pipeline config for every value of an environment variable
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/sweep; mkdir -p $O
export TMPDIR=/tmp
VAR=$1; VALS=$2
if [ "$3" = "pytest" ]; then ( timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_walk.py tests/test_gpu_probe.py tests/test_gpu_configs.py tests/test_fuzz.py -m gpu -q -x 2>&1 | tail -3 ) | tee $O/pytest.log; fi
run() { timeout 600 python bench.py "$@" --no-cpu-baseline --no-pmc --no-second-config --no-api --no-fast 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
r=d['roofline']
print('| %s | %.1f | %s |' % (d['config']['workload'].split(',')[0], d['value'], r['kernel_ms']))
PY
}
for V in $VALS; do
export $VAR=$V
echo "== $VAR=$V"
run --scene large/ajax_standin --width 1920 --height 1080 --steps 32 --warmup 2
run --scene glass --width 1920 --height 1080 --maxdepth 12 --steps 32 --warmup 2
run --scene veach --width 3840 --height 2160 --steps 8 --warmup 1
run --scene many_spheres --width 1024 --height 768 --steps 64 --warmup 2
done 2>&1 | tee $O/sweep_$VAR.md
