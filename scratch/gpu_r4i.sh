#!/bin/bash
# round 4 (i): the reference's other shipped scenes through the C-ABI
mkdir -p gpurun_out/r4i
timeout 600 python -m pytest tests/test_gpu_reference_scenes.py -q -s -m gpu 2>&1 | tail -40 > gpurun_out/r4i/pytest_refscenes.log
tail -15 gpurun_out/r4i/pytest_refscenes.log
