#!/bin/bash
# call z4: after the tolerance arm got its light loops back (TN_LIGHT_RECS off there): its tests and the driver's bench command once more for the line's fast_* fields
O=gpurun_out/r6z4; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fast.py tests/test_gpu_arith.py tests/test_gpu_parity.py -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
( time python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real; cp bench_detail.json $O/bench_detail.json
python scratch/roofline_table.py $O/bench_detail.json > $O/roofline_inputs.md
python - <<'PY'
import json
d=json.load(open('gpurun_out/r6z4/bench_default.json'))
print('headline', d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['fast_over_exact'], len(json.dumps(d)))
for c in d.get('configs', []): print(c['workload'][:60], c.get('value'), c.get('kernel'), c.get('frac'), c.get('job_counter_over_compulsory'), c.get('fast_over_exact'))
PY
