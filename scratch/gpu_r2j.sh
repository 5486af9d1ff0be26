#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2j; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -x -q -s 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -30 ) > $O/pytest.log; grep -E "L2|paths|passed|failed|Error|error" $O/pytest.log | head -30
python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline > $O/b.json 2> $O/b.err; python -c "
import json; d=json.load(open('$O/b.json')); print('cornell', d['value'], 'fast', d['fast']); print('ajax', d['configs'][1]['value'], d['configs'][1].get('fast'))"
