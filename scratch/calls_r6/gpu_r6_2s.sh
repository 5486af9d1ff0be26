#!/bin/bash
# call 2s: flat-scan scenes of 33 ... 64 primitives (the scan mask's upper half) against the live reference
O=gpurun_out/r6_2s; mkdir -p $O
timeout 900 python -m pytest tests/test_fuzz.py -x -q -k large_flat_scan > $O/pytest.log 2>&1; tail -8 $O/pytest.log
