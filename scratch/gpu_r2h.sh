#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2h; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > $O/pytest.log; tail -12 $O/pytest.log
