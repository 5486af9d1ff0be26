#!/bin/bash
# bench.py's N-rank path end to end on ONE device (gloo stand-in for RCCL; validation, not a measurement): N = 2, 4, 8
cd $GRAFT_REPO_ROOT
export TINSEL_BENCH_BACKEND=gloo TINSEL_BENCH_ONE_DEVICE=1 HSA_ENABLE_IPC_MODE_LEGACY=0
for n in 2 4 8; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29600+n)) bench.py --gpus $n --steps 8 --warmup 2 > /tmp/mr_$n.json 2> /tmp/mr_$n.err
  echo "N=$n rc=$?"; grep -h "validation" /tmp/mr_$n.err | tail -1; python -c "
import json
for l in open('/tmp/mr_$n.json'):
    if l.startswith('{'):
        d=json.loads(l); print('  n_gpus', d['n_gpus'], 'value %.1f' % d['value'], 'ms_per_step %.3f' % d['ms_per_step'], d['config']['parallelism'], d['roofline']['kernel_ms'])"
done
