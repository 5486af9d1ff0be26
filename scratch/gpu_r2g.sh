#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2g; mkdir -p $O
export TMPDIR=/tmp
( time python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | tail -3
python -c "
import json; d=json.load(open('$O/bench_default.json'))
print('headline', d['value'], d['ms_per_step'], d['timed_blocks'], d['roofline']['bound'], d['roofline']['frac'], d['roofline']['traffic'], d['api_1pass_msamples_s'], d['pcie_inclusive_msamples_s'])
for c in d.get('configs', []): print(c.get('value'), c.get('unavailable'), (c.get('roofline') or {}).get('kernel'), (c.get('roofline') or {}).get('frac'), (c.get('roofline') or {}).get('frac_hbm_algorithmic'), (c.get('roofline') or {}).get('frac_hbm_counter'))
"
tail -5 $O/bench_default.err
( timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 ) > $O/pytest.log; tail -3 $O/pytest.log
