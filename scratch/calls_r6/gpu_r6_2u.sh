#!/bin/bash
# call 2u: the committed tree once more on another box: smoke(), the GPU suite, the driver's bench command (how long, how large a line)
O=gpurun_out/r6_2u; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
( time python bench.py > $O/bench_noflags.json 2> $O/bench_noflags.err ) 2>&1 | grep real
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6_2u/bench_noflags.json').read().strip().splitlines()[-1])
print('no flags:', d['value'], d['steps'], d['warmup'], d['ms_per_step'], len(json.dumps(d)), [c['value'] for c in d['configs']])
PY
