#!/bin/bash
# call 2b: what bounds k_step on glass.tin (104 ps per path-step against 60 on the 524k-triangle config): its counters under the paired pipeline
O=gpurun_out/r6_2b; mkdir -p $O
PMC_EXTRA="--pipeline paired" bash scratch/gpu_pmc_kernels.sh $GRAFT_REPO_ROOT/$O paired "glass 1920 1080 12 20" "large/transmission 1920 1080 16 20"
( cd /tmp; export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/stats -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-fast --no-api --no-ubench --no-second-config --no-more-configs --scene glass --width 1920 --height 1080 --maxdepth 12 --pipeline paired > /dev/null 2>&1 )
DB=$(find $O/stats -name "*.db" | head -1); [ -n "$DB" ] && python scratch/rocprof_summary.py $DB > $O/kernel_stats_glass_paired.md; rm -rf $O/stats; head -20 $O/kernel_stats_glass_paired.md
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fast --no-api --no-ubench --no-second-config --no-more-configs --scene glass --width 1920 --height 1080 --maxdepth 12 --pipeline paired > $O/bench_glass_paired.json 2> $O/bench_glass_paired.err; cp bench_detail.json $O/bench_detail_glass_paired.json
python scratch/roofline_table.py $O/bench_detail_glass_paired.json > $O/roofline_glass_paired.md; tail -12 $O/roofline_glass_paired.md
