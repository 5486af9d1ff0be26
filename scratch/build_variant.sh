#!/bin/bash
# build_variant.sh NAME [extra hipcc flags...]: a second build of the library as scratch/ab/libtinsel_hip_NAME.so for an A/B
# within one gpurun call (TINSEL_HIP_LIB).  Flags are appended to BOTH translation units' (parity and tolerance arm).
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
mkdir -p scratch/ab /tmp/tn_variant_$NAME
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Wall -Wno-unused-function -Wno-unused-variable"
hipcc $COMMON -ffp-contract=off -fno-fast-math "$@" -c tinsel_amd/csrc/tinsel_hip.hip -o /tmp/tn_variant_$NAME/a.o &
hipcc $COMMON -DTN_FAST=1 -ffp-contract=fast -fno-hip-fp32-correctly-rounded-divide-sqrt -freciprocal-math -fgpu-flush-denormals-to-zero "$@" -c tinsel_amd/csrc/tinsel_fast.hip -o /tmp/tn_variant_$NAME/b.o &
wait
hipcc --offload-arch=gfx950 -fPIC -shared -o scratch/ab/libtinsel_hip_$NAME.so /tmp/tn_variant_$NAME/a.o /tmp/tn_variant_$NAME/b.o
echo built scratch/ab/libtinsel_hip_$NAME.so
