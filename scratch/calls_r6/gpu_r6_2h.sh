#!/bin/bash
# call 2h: k_bounce's section shares before / after the scan's new fetch order (cornell, veach); parity of the tree (rcp by value)
O=gpurun_out/r6_2h; mkdir -p $O
for s in cornell veach; do for b in sections_base sections; do echo "== $s, $b"; TINSEL_HIP_LIB=scratch/ab/libtinsel_hip_$b.so python scratch/prof_sections.py $s 2>&1 | grep -v amdgpu.ids; done; done > $O/bounce_sections.txt; cat $O/bounce_sections.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_walk.py tests/test_gpu_paired.py -x -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
