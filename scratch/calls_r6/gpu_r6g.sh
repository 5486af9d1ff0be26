#!/bin/bash
# call g: the PAIRED pipeline (tn_paired.h): parity first
O=gpurun_out/r6g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_paired.py -q -x > $O/pytest_paired.log 2>&1; tail -30 $O/pytest_paired.log
