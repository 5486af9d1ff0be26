#!/bin/bash
# call o: after the pending records' allocation was fixed (call n wrote past it with K > 1): the paired tests, then the scenes that failed there
O=gpurun_out/r6o; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_paired.py tests/test_gpu_switches.py -q -x > $O/pytest.log 2>&1; tail -2 $O/pytest.log
timeout 600 python scratch/ab_rates.py --repeat 1 --lib 'split=args:--pipeline split' --lib 'paired=args:--pipeline paired' table meshlight motionblur veach4k > $O/ab_paired_k.md 2> $O/ab.err; cat $O/ab_paired_k.md
