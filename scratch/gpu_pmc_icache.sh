#!/bin/bash
mkdir -p gpurun_out/pmc_ic; export TMPDIR=/tmp; cd /tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_ic
rocprofv3 -L 2>/dev/null | grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQC_INST[A-Z_]*\|SQ_INST_CYCLES[A-Z_]*\|SQ_WAIT_INST[A-Z_]*" | sort -u | head -30 > $O/names.txt
cat $O/names.txt
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  tag=$(echo $set | md5sum | cut -c1-6)
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $O -o ic_$tag --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 16 --warmup 0 --no-cpu-baseline > /dev/null 2> $O/err_$tag.txt
  tail -2 $O/err_$tag.txt
done
ls $O
