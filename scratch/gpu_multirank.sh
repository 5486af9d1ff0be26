#!/bin/bash
export TINSEL_BENCH_BACKEND=gloo TINSEL_BENCH_ONE_DEVICE=1
for n in 2 4; do
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2951$n bench.py --gpus $n --steps 16 --warmup 2 > /tmp/mr.json 2> /tmp/mr.err
echo "rc=$?"; grep -i "validation\|error\|Traceback" /tmp/mr.err | head -5
python - <<PY
import json
try:
    d=json.loads(open('/tmp/mr.json').read().strip().splitlines()[-1])
    print('n_gpus', d['n_gpus'], 'value', round(d['value'],1), 'ms_per_step', round(d['ms_per_step'],3), d['config']['parallelism'], 'rays/sample', round(d['config']['rays_per_sample'],3), 'cpu', d['cpu_baseline'])
except Exception as e:
    print('no json', e); print(open('/tmp/mr.err').read()[-1500:])
PY
done
