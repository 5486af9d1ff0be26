#!/bin/bash
# round 5, call t: the shadow rays of a hit two at a time through the flat scan (k_bounce<..., PAIR>, trace_shadow2) in a scratch build:
# parity, then scenes with several light samples per hit with the switch off / on
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5t; mkdir -p $O
V=TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/libtinsel_hip_pair.so
( time env $V timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_scenes.py tests/test_gpu_probe.py -m gpu -q -x 2>&1 | tail -12 ) > $O/pytest_pair.log 2>&1; grep -a "passed\|failed\|rror" $O/pytest_pair.log | tail -8
{
echo "| environment | workload | Msamples/s | kernel ms |"; echo "|---|---|---|---|"
for W in "--scene veach --width 3840 --height 2160 --steps 20 --warmup 5" "--scene features --width 1920 --height 1080 --maxdepth 6 --steps 16 --warmup 2" \
         "--scene veach --width 1024 --height 1024 --steps 20 --warmup 5" "--scene large/env_loft --width 1920 --height 1080 --steps 20 --warmup 5"; do
  bash scratch/gpu_envs.sh "$V TINSEL_HIP_PAIR_SHADOW=0" "$V TINSEL_HIP_PAIR_SHADOW=1" "$V TINSEL_HIP_PAIR_SHADOW=0" "$V TINSEL_HIP_PAIR_SHADOW=1" -- $W
done
} > $O/ab_pair_shadow.md 2>&1; sed "s|TINSEL_HIP_LIB=[^ ]*libtinsel_hip_pair.so ||" $O/ab_pair_shadow.md
