#!/bin/bash
# call z6: the closing evidence run on the final tree (after the two parity fixes of the hunt at scale) -- the GPU suite, the driver's bench command (plain; under rocprofv3 --kernel-trace --stats),
# per-kernel counters of every configuration, k_walk's section profile, the animation's per-frame cost, the spread over fresh processes
O=gpurun_out/r6z6; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
( time python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real; cp bench_detail.json $O/bench_detail.json
python scratch/roofline_table.py $O/bench_detail.json > $O/roofline_inputs.md
( cd /tmp; export TMPDIR=/tmp; timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/stats -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-fast --no-api --no-ubench > $GRAFT_REPO_ROOT/$O/bench_under_stats.json 2> /dev/null )
DB=$(find $O/stats -name "*.db" | head -1); [ -n "$DB" ] && python scratch/rocprof_summary.py $DB > $O/kernel_stats.md; rm -rf $O/stats
head -14 $O/kernel_stats.md
bash scratch/gpu_pmc_kernels.sh $GRAFT_REPO_ROOT/$O z "cornell 1024 1024 4 20" "large/ajax_standin 1920 1080 4 20" "large/ajax_aphrodite 1920 1080 4 20" "glass 1920 1080 12 20" "veach 3840 2160 4 20" > /dev/null 2>&1
bash scratch/build_walkprof.sh > /dev/null 2>&1
for w in "large/ajax_standin 1920 1080 4 20" "large/ajax_aphrodite 1920 1080 4 20" "glass 1920 1080 12 20"; do TINSEL_HIP_LIB=scratch/libtinsel_hip_walkprof.so python scratch/walk_prof.py $w; done 2>&1 | grep -v amdgpu.ids > $O/walk_profile.txt
python scratch/anim_cost.py 2>&1 | grep -v amdgpu.ids > $O/anim_cost.txt; cat $O/anim_cost.txt
python scratch/ab_rates.py --repeat 5 glass cfg3 aphrodite > $O/spread.md 2> /dev/null; cat $O/spread.md
python scratch/ab_rates.py --repeat 1 cornell veach4k cfg1 many_spheres motionblur table transmission meshlight > $O/other_rates.md 2> /dev/null; cat $O/other_rates.md
python - <<'PY'
import json
d=json.load(open('gpurun_out/r6z6/bench_default.json'))
print('headline', d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], len(json.dumps(d)))
for c in d.get('configs', []): print(c['workload'][:60], c.get('value'), c.get('kernel'), c.get('frac'), c.get('job_counter_over_compulsory'))
PY
