#!/bin/bash
# call 2j: the flat scan's loop software-pipelined (the next primitive's box and record requested before this one is tested)
O=gpurun_out/r6_2j; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_paired.py tests/test_gpu_walk.py -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 2400 python scratch/ab_rates.py --repeat 3 --lib one=scratch/ab/libtinsel_hip_nopf.so --lib pipe=tinsel_amd/libtinsel_hip.so \
    cornell veach4k cfg1 glass cfg3 > $O/ab_pipe.md 2> $O/ab.err; cat $O/ab_pipe.md
