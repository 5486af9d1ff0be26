#!/bin/bash
# round 3, call Q: full suite; PLOC build time after the one-word read-back; radius sweep is compile-time (skipped)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3q; mkdir -p $OUT
( time timeout 1500 python -m pytest tests -m gpu -q -s --maxfail=10 2>&1 | grep -E "passed|failed|built on the device|tris\) built|Error" ) 2>&1 | tee $OUT/pytest_gpu.log
