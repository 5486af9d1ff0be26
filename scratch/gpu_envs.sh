#!/bin/bash
# usage: gpu_envs.sh "ENV1=a ENV2=b" "ENV1=c" ... -- bench args     one table row per environment set ("-" = default); reads bench_detail.json
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
SETS=()
while [ "$1" != "--" ]; do SETS+=("$1"); shift; done; shift
for S in "${SETS[@]}"; do
( [ "$S" != "-" ] && export $S
timeout 600 python bench.py "$@" --no-cpu-baseline --no-pmc --no-second-config --no-more-configs --no-api --no-fast --no-ubench 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('bench_detail.json'))
print('| $S | %s | %.1f | %s |' % (d['config']['workload'].split(',')[0], d['value'], d['roofline']['kernel_ms']))
PY
)
done
