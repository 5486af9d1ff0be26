#!/bin/bash
# call p: glass / table with EVERY mesh in HBM and walked (small_mesh_bytes 0, walk_min_tris 1), so that the paired pipeline's k_step runs its lean variant
O=gpurun_out/r6p; mkdir -p $O
timeout 900 python scratch/ab_rates.py --repeat 2 --lib 'split=args:--pipeline split' --lib 'paired=args:--pipeline paired' --lib 'paired_lean=tuning:{"walk_min_tris":1,"small_mesh_bytes":0};args:--pipeline paired' --lib 'split_lean=tuning:{"walk_min_tris":1,"small_mesh_bytes":0};args:--pipeline split' glass table > $O/ab_paired_lean.md 2> $O/ab.err; cat $O/ab_paired_lean.md
