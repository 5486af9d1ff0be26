#!/bin/bash
mkdir -p gpurun_out/lbvh
timeout 600 python -m pytest tests/test_gpu_lbvh.py -m gpu -x -q -s 2>&1 | grep -i "LBVH\|passed\|failed\|hits"
for b in reference lbvh; do
  timeout 600 python bench.py --scene large/ajax_standin --width 1920 --height 1080 --steps 16 --warmup 2 --bvh $b --no-cpu-baseline 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
  python - <<PY
import json
d=json.load(open('/tmp/b.json'))
r=d['roofline']
print('ajax bvh=%-10s Msamples/s %7.1f Mrays/s %8.1f I %.1f T %.2f P %.2f build_ms %s' % ('$b', d['value'], d['mrays_per_s'], r['I'], r['T'], r['P'], d['config']['mesh_bvh_build_ms']), r['kernel_ms'])
PY
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/lbvh -o lbvh -- python $GRAFT_REPO_ROOT/scratch/lbvh_build.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; ls gpurun_out/lbvh | head
