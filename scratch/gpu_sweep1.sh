#!/bin/bash
# usage: gpu_sweep1.sh VAR "v1 v2 ..." -- bench.py args...   one bench line per value of an environment variable
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
VAR=$1; VALS=$2; shift 3
for V in $VALS; do
export $VAR=$V
timeout 600 python bench.py "$@" --no-cpu-baseline --no-pmc --no-second-config --no-api --no-fast 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('| $VAR=$V | %s | %.1f | %s |' % (d['config']['workload'].split(',')[0], d['value'], d['roofline']['kernel_ms']))
PY
done
