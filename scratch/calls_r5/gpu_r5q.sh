#!/bin/bash
# round 5, call q: k_accumulate_piped (staging of pass s + 1 by six waves while four gather pass s; few tiles / a shard's long pass loop) in a
# scratch build with a switch: parity, then one shard of 8 and small one-shard frames with the switch off / on
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5q; mkdir -p $O
V=TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/libtinsel_hip_pipe.so
( time env $V timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_group.py tests/test_gpu_multirank.py tests/test_gpu_reference_scenes.py -m gpu -q -x 2>&1 | tail -12 ) > $O/pytest_pipe.log 2>&1; grep -a "passed\|failed\|rror" $O/pytest_pipe.log | tail -8
for P in 0 1; do
{
echo "TINSEL_HIP_ACC_PIPE=$P"; echo
echo "| workload | numbering | 20 steps | paths/s vs one shard | kernel ms (20 steps) |"; echo "|---|---|---|---|---|"
env $V TINSEL_HIP_ACC_PIPE=$P timeout 300 python scratch/shard_emul.py cornell 1024 1024 8
env $V TINSEL_HIP_ACC_PIPE=$P timeout 300 python scratch/shard_emul.py veach 3840 2160 8
env $V TINSEL_HIP_ACC_PIPE=$P timeout 300 python scratch/shard_emul.py large/ajax_standin 1920 1080 8 4
echo
} 2>&1 | grep -v amdgpu.ids
done > $O/shard_pipe.md; cat $O/shard_pipe.md | cut -c1-300
{
echo "| environment | workload | Msamples/s | kernel ms |"; echo "|---|---|---|---|"
for W in "--scene cornell --width 256 --height 256 --steps 16 --warmup 4" "--scene cornell --width 512 --height 512 --steps 20 --warmup 5" "--scene gloss --width 512 --height 512 --steps 64 --warmup 5"; do
  bash scratch/gpu_envs.sh "$V TINSEL_HIP_ACC_PIPE=0" "$V TINSEL_HIP_ACC_PIPE=1" "$V TINSEL_HIP_ACC_PIPE=0" "$V TINSEL_HIP_ACC_PIPE=1" -- $W
done
} > $O/ab_pipe_small.md 2>&1; sed "s|$GRAFT_REPO_ROOT/scratch/ab/||" $O/ab_pipe_small.md
