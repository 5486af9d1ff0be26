#!/bin/bash
# call f: k_walk's grid in resident sets (tinsel_hip_tuning::walk_grid_mult): finished workgroups replaced by fresh ones instead of one set of static ranges
O=gpurun_out/r6f; mkdir -p $O
python -m pytest tests/test_gpu_switches.py -q -k "grid_mult or defaults or set_tuning" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
python scratch/ab_rates.py --repeat 2 --lib 'g1=tuning:{}' --lib 'g2=tuning:{"walk_grid_mult":2}' --lib 'g3=tuning:{"walk_grid_mult":3}' --lib 'g4=tuning:{"walk_grid_mult":4}' --lib 'g8=tuning:{"walk_grid_mult":8}' cfg3 aphrodite glass table > $O/ab_walk_grid.md 2> $O/ab.err; cat $O/ab_walk_grid.md
