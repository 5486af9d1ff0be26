#!/bin/bash
# round 5, call r: where k_bounce's cycles go at four waves per SIMD (library built with -DTN_PROFILE_SECTIONS: per-wave clock deltas per section)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5r; mkdir -p $O
for S in cornell veach gloss; do
  echo "== $S"; TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/libtinsel_hip_sections.so timeout 200 python scratch/prof_sections.py $S 2>&1 | grep -v amdgpu.ids
done > $O/bounce_sections.txt; cat $O/bounce_sections.txt
