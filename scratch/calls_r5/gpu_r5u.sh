#!/bin/bash
# round 5, call u: k_accumulate_tiled's LDS sized by its window (14-16 KB a workgroup instead of 26.7: six or seven workgroups per CU instead of
# five), and the same forced to 64 VGPRs for eight -- scratch builds against the in-tree library
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5u; mkdir -p $O
A=TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/libtinsel_hip_acclds.so
B=TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/libtinsel_hip_acclds8.so
( time env $A timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "accumulate or filter or shard or golden" 2>&1 | tail -12 ) > $O/pytest_acclds.log 2>&1; grep -a "passed\|failed\|rror" $O/pytest_acclds.log | tail -8
{
echo "| environment | workload | Msamples/s | kernel ms |"; echo "|---|---|---|---|"
for W in "--scene cornell --width 1024 --height 1024 --steps 20 --warmup 5" "--scene veach --width 3840 --height 2160 --steps 20 --warmup 5" \
         "--scene gloss --width 1024 --height 1024 --steps 64 --warmup 8" "--scene cornell --width 512 --height 512 --steps 20 --warmup 5"; do
  bash scratch/gpu_envs.sh "-" "$A" "$B" "-" "$A" "$B" -- $W
done
} > $O/ab_acc_lds.md 2>&1; sed "s|TINSEL_HIP_LIB=[^ ]*libtinsel_hip_||" $O/ab_acc_lds.md
