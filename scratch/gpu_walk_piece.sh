#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_walk.py -m gpu -q -x 2>&1 | tail -2
for S in 1 0; do for P in 0 1024 4096 16384; do
export TINSEL_HIP_WALK_LIST_STEP=$S TINSEL_HIP_WALK_PIECE=$P
timeout 600 python bench.py --scene large/ajax_standin --width 1920 --height 1080 --steps 32 --warmup 2 --no-cpu-baseline --no-pmc --no-second-config --no-api --no-fast 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('| step=$S piece=$P | %.1f | k_walk %.2f |' % (d['value'], d['roofline']['kernel_ms']['k_walk']))
PY
done; done
