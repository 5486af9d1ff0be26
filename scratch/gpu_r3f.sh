#!/bin/bash
# round 3, call F: do two half-batches on two streams overlap usefully (the tails of one under the other)?
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/r3f; mkdir -p $OUT
( python scratch/two_streams.py large/ajax_standin 1920 1080 32
  python scratch/two_streams.py large/ajax_standin 1920 1080 64
  python scratch/two_streams.py glass 1920 1080 32
  python scratch/two_streams.py many_spheres 1024 768 64
  python scratch/two_streams.py cornell 1024 1024 20
  python scratch/two_streams.py cornell 1024 1024 64
  python scratch/two_streams.py veach 3840 2160 8 ) 2>&1 | grep -v "^$" | tee $OUT/two_streams.txt
