#!/bin/bash
# tolerance arm of the split-pipeline configs for several builds: gpu_fast_ab.sh lib_a lib_b ...
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for L in "$@"; do export TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/$L; echo "== $L"
for sc in "large/ajax_standin --width 1920 --height 1080 --steps 32 --warmup 2" "glass --width 1920 --height 1080 --maxdepth 12 --steps 32 --warmup 2" "many_spheres --width 1024 --height 768 --steps 64 --warmup 2"; do
timeout 600 python bench.py --scene $sc --arith fast --no-cpu-baseline --no-pmc --no-second-config --no-api --no-fast 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python -c "
import json
d=json.load(open('/tmp/b.json')); print('| %s | fast %.1f | %s |' % (d['config']['workload'].split(',')[0], d['value'], d['roofline']['kernel_ms']))"
done; done
