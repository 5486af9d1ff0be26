#!/bin/bash
cd $GRAFT_REPO_ROOT/scratch/ubench
for n in 131072 262144 524288 1048576 4194304; do timeout 120 ./gather_bench $n 6 8; done
timeout 120 ./gather_bench 524288 6 1 | head -1
timeout 120 ./gather_bench 65536 6 1 | head -1
timeout 120 ./gather_bench 16384 6 1 | head -1
