#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2i; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > $O/pytest.log; tail -3 $O/pytest.log
for d in 0 1 2 4 8; do
TINSEL_HIP_LOOKAHEAD_DEPTH=$d python bench.py --steps 20 --warmup 5 --no-pmc --no-second-config --no-cpu-baseline > $O/b.json 2> $O/b.err; python -c "
import json; d=json.load(open('$O/b.json')); print('depth $d 1024:', d['value'], 'api_1pass', d['api_1pass_msamples_s'], 'plain', d['api_1pass_plain_msamples_s'], 'pcie16', d['pcie_inclusive_msamples_s'])"
done
for d in 0 4 16 32; do
TINSEL_HIP_LOOKAHEAD_DEPTH=$d python bench.py --steps 16 --warmup 5 --width 256 --height 256 --no-pmc --no-second-config --no-cpu-baseline > $O/b256.json 2> $O/b256.err; python -c "
import json; d=json.load(open('$O/b256.json')); print('depth $d 256:', d['value'], 'api_1pass', d['api_1pass_msamples_s'], 'plain', d['api_1pass_plain_msamples_s'], 'pcie16', d['pcie_inclusive_msamples_s'])"
done
