#!/bin/bash
# call d: k_walk_rays with the walked primitives' records staged in LDS (tn_walk.h kWalkPrimWords) against the tree before it; the whole GPU suite
mkdir -p gpurun_out/r6d
python -m pytest tests -m gpu -q > gpurun_out/r6d/pytest_gpu.log 2>&1; tail -6 gpurun_out/r6d/pytest_gpu.log
python scratch/ab_rates.py --lib primlds=scratch/ab/libtinsel_hip_primlds.so --lib base=scratch/ab/libtinsel_hip_base.so --repeat 3 glass table transmission > gpurun_out/r6d/ab_walk_prim_lds.md 2> gpurun_out/r6d/ab.err
cat gpurun_out/r6d/ab_walk_prim_lds.md
