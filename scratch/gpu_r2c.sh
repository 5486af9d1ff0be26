#!/bin/bash
cd $GRAFT_REPO_ROOT
export TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/libtinsel_hip_walkprof.so
python scratch/walk_prof.py large/ajax_standin 1920 1080 4 32 2>&1 | grep -v amdgpu.ids
TINSEL_HIP_WALK_LEAFMIN=8 python scratch/walk_prof.py large/ajax_standin 1920 1080 4 32 2>&1 | grep -v amdgpu.ids
TINSEL_HIP_WALK_REFILL=32 python scratch/walk_prof.py large/ajax_standin 1920 1080 4 32 2>&1 | grep -v amdgpu.ids
python scratch/walk_prof.py ajax_standin_96 1920 1080 4 32 2>&1 | grep -v amdgpu.ids
python scratch/walk_prof.py glass 1920 1080 12 32 2>&1 | grep -v amdgpu.ids
