#!/bin/bash
# call 2g: the whole GPU suite on the tree with the scan's fetch order (scanMask, record with box, quad offsets in the record) and the lean kernels' quad level
O=gpurun_out/r6_2g; mkdir -p $O
python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
python scratch/ab_rates.py --repeat 3 --lib base=scratch/ab/libtinsel_hip_base.so --lib now=tinsel_amd/libtinsel_hip.so cornell veach4k glass cfg3 aphrodite many_spheres motionblur table transmission meshlight > $O/ab_all.md 2> $O/ab.err; cat $O/ab_all.md
