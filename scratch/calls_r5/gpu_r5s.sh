#!/bin/bash
# round 5, call s: what the shadow traces / the closest-hit traces of k_bounce spend their cycles on (trace_flat's section timers)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5s; mkdir -p $O
for S in veach cornell features; do
  for L in trnee trclo; do
  echo "== $S, $L"; TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/libtinsel_hip_$L.so timeout 200 python scratch/trace_sections.py $S 2>&1 | grep -v amdgpu.ids
  done
done > $O/trace_sections.txt; cat $O/trace_sections.txt
