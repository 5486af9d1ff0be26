#!/bin/bash
# builds scratch/libtinsel_hip_walkprof.so: the library with k_walk's section timers compiled in (-DTN_WALK_PROF), for walk_prof.py
# (through build_variant.sh: the in-tree library and its build record are left alone)
set -e
cd "$(dirname "$0")/.."
bash scratch/build_variant.sh walkprof -DTN_WALK_PROF > /dev/null
mv scratch/ab/libtinsel_hip_walkprof.so scratch/libtinsel_hip_walkprof.so
echo built scratch/libtinsel_hip_walkprof.so
