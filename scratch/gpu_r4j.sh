#!/bin/bash
# round 4 (j): two half-batches on two streams, re-measured on the final tree
mkdir -p gpurun_out/r4j
{
for args in "cornell 1024 1024 20" "cornell 1024 1024 20" "cornell 1024 1024 64" "cornell 256 256 16" "cornell 256 256 16" "veach 3840 2160 8" "veach 3840 2160 20" "glass 1920 1080 20"; do
  timeout 120 python scratch/two_streams.py $args 2>&1 | grep passes
done
} > gpurun_out/r4j/two_streams.txt
cat gpurun_out/r4j/two_streams.txt
