#!/bin/bash
# call q: camera paths as 32-B records without an origin in the paired pipeline
O=gpurun_out/r6q; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_paired.py -q -x > $O/pytest.log 2>&1; tail -2 $O/pytest.log
timeout 600 python scratch/ab_rates.py --repeat 3 cfg3 aphrodite transmission > $O/ab_compact_fresh.md 2> $O/ab.err; cat $O/ab_compact_fresh.md
