#!/bin/bash
# round 5, call j: k_shade / k_generate run in two modes from process to process on one box (glass k_shade 8.4 or 9.7 ms: calls b, e, i).
# Is it where the path state's arrays land?  -DTN_ALLOC_SKEW: the k-th array of a batch starts k x 4352 B (skew) / k x 69 888 B (skew2) into its
# allocation.  Six processes each.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5j; mkdir -p $O
S=TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/libtinsel_hip_skew.so
T=TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/libtinsel_hip_skew2.so
{
echo "| library | workload | Msamples/s | kernel ms |"; echo "|---|---|---|---|"
for W in "--scene glass --width 1920 --height 1080 --maxdepth 12 --steps 20 --warmup 5" "--scene large/ajax_standin --width 1920 --height 1080 --maxdepth 4 --steps 20 --warmup 5"; do
  bash scratch/gpu_envs.sh "-" "$S" "$T" "-" "$S" "$T" "-" "$S" "$T" "-" "$S" "$T" "-" "$S" "$T" "-" "$S" "$T" -- $W
done
} > $O/ab_alloc_skew.md 2>&1; cat $O/ab_alloc_skew.md
