#!/bin/bash
# builds scratch/libtinsel_hip_walkprof.so: the library with k_walk's section timers compiled in (-DTN_WALK_PROF), for walk_prof.py
set -e
cd "$(dirname "$0")/.."
cp tinsel_amd/libtinsel_hip.so /tmp/libtinsel_hip_keep.so
python -m tinsel_amd.build --force -DTN_WALK_PROF > /dev/null
cp tinsel_amd/libtinsel_hip.so scratch/libtinsel_hip_walkprof.so
cp /tmp/libtinsel_hip_keep.so tinsel_amd/libtinsel_hip.so
touch tinsel_amd/libtinsel_hip.so
echo built scratch/libtinsel_hip_walkprof.so
