#!/bin/bash
# call e: evidence on the tree after the file split -- the GPU suite, the driver's bench command plain and under rocprofv3 --stats, per-kernel
# counters of every configuration, k_walk's section profile, the animation's per-frame cost, the spread over fresh processes, k_walk's two
# thresholds on the irregular tree
O=gpurun_out/r6e; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; cp bench_detail.json $O/bench_detail.json
tail -c 600 $O/bench_default.json
( cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/stats -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-pmc > $GRAFT_REPO_ROOT/$O/bench_under_stats.json 2> /dev/null )
DB=$(find $O/stats -name "*.db" | head -1); [ -n "$DB" ] && python scratch/rocprof_summary.py $DB > $O/kernel_stats.md; rm -rf $O/stats
head -12 $O/kernel_stats.md
bash scratch/gpu_pmc_kernels.sh $GRAFT_REPO_ROOT/$O z "cornell 1024 1024 4 20" "large/ajax_standin 1920 1080 4 20" "large/ajax_aphrodite 1920 1080 4 20" "glass 1920 1080 12 20" "veach 3840 2160 4 20" > /dev/null 2>&1
bash scratch/build_walkprof.sh > /dev/null 2>&1
for w in "large/ajax_standin 1920 1080 4 20" "large/ajax_aphrodite 1920 1080 4 20" "glass 1920 1080 12 20"; do TINSEL_HIP_LIB=scratch/libtinsel_hip_walkprof.so python scratch/walk_prof.py $w; done > $O/walk_profile.txt 2>&1
cat $O/walk_profile.txt
python scratch/anim_cost.py > $O/anim_cost.txt 2>&1; cat $O/anim_cost.txt
python scratch/ab_rates.py --repeat 5 glass cfg3 aphrodite > $O/spread.md 2> /dev/null; cat $O/spread.md
python scratch/ab_rates.py --repeat 1 --lib 'r24l8=tuning:{}' --lib 'r16l8=tuning:{"walk_refill_min":16}' --lib 'r32l8=tuning:{"walk_refill_min":32}' --lib 'r24l4=tuning:{"walk_leaf_min":4}' --lib 'r24l16=tuning:{"walk_leaf_min":16}' --lib 'r40l12=tuning:{"walk_refill_min":40,"walk_leaf_min":12}' --lib 'stack6=tuning:{"walk_lds_stack":6}' --lib 'stack12=tuning:{"walk_lds_stack":12}' aphrodite cfg3 > $O/ab_walk_thresholds.md 2> /dev/null; cat $O/ab_walk_thresholds.md
