#!/bin/bash
# round 4 evidence run on the tree as committed: GPU suite, the driver's bench line (live PMC + calibration, per-kernel table),
# rocprofv3 --kernel-trace --stats of the same command, per-kernel SQ tables of the BASELINE configs, one bench line per config in both
# arithmetic arms, k_walk's section profile, bench.py without flags, the group mode on one device
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4final; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -aE "passed|failed|per-pixel L2|rebuilt on the device" ) > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
( time timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O -o stats -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-fast --no-api --no-ubench > $O/bench_under_stats.json 2> $O/stats.err
cd $GRAFT_REPO_ROOT
python scratch/rocprof_summary.py $(ls $O/*stats*.db 2>/dev/null | head -1) > $O/kernel_stats.md 2>&1
bash scratch/gpu_pmc_kernels.sh $O configs "cornell 1024 1024 4 20" "large/ajax_standin 1920 1080 4 20" "glass 1920 1080 12 20" "veach 3840 2160 4 20" "many_spheres 1024 768 4 64" > /dev/null
run() { timeout 600 python bench.py "$@" --no-cpu-baseline --no-pmc --no-second-config --no-ubench --no-api 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
r=d['roofline']; f=d.get('fast') or {}
print('| %s | %.1f | %.1f | %.2f | %s | %s | %s |' % (d['config']['workload'].split(',')[0], d['value'], d['mrays_per_s'], d['config']['rays_per_sample'], r['kernel_ms'], ('%.1f' % f['msamples_s']) if f.get('msamples_s') else '-', ('%.2e' % f['l2_vs_exact_at_spp'][0]) if f.get('l2_vs_exact_at_spp') else '-'))
PY
}
( echo "| config | Msamples/s (exact) | Mrays/s | rays/sample | kernel ms of one timed block | Msamples/s (fast arm) | fast-vs-exact L2 @ 256 spp |"; echo "|---|---|---|---|---|---|---|"
run --scene cornell --width 256 --height 256 --steps 16 --warmup 4
run --scene cornell --steps 20 --warmup 5
run --scene cornell --steps 64 --warmup 8
run --scene large/ajax_standin --width 1920 --height 1080 --steps 20 --warmup 2
run --scene large/ajax_standin --width 1920 --height 1080 --steps 32 --warmup 2
run --scene glass --width 1920 --height 1080 --maxdepth 12 --steps 20 --warmup 2
run --scene veach --width 3840 --height 2160 --steps 8 --warmup 1
run --scene large/env_loft --width 1024 --height 512 --steps 64 --warmup 2
run --scene many_spheres --width 1024 --height 768 --steps 64 --warmup 2
run --scene features --width 1920 --height 1080 --maxdepth 6 --steps 16 --warmup 1
run --scene motionblur --width 1920 --height 1080 --steps 16 --warmup 1
run --scene gloss --steps 64 --warmup 8 ) > $O/configs.md 2>&1
cat $O/configs.md
TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/libtinsel_hip_walkprof.so python scratch/walk_prof.py large/ajax_standin 1920 1080 4 20 2>&1 | grep -v amdgpu.ids > $O/walk_profile.txt
TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/libtinsel_hip_walkprof.so python scratch/walk_prof.py ajax_standin_96 1920 1080 4 20 2>&1 | grep -v amdgpu.ids >> $O/walk_profile.txt
cat $O/walk_profile.txt
( time timeout 900 python bench.py > $O/bench_noflags.json 2> $O/bench_noflags.err ) 2>&1 | grep real
python bench.py --group --gpus 2 --steps 20 --warmup 2 > $O/bench_group2_one_device.json 2> $O/bench_group.err; cat $O/bench_group2_one_device.json
TINSEL_BENCH_BACKEND=gloo TINSEL_BENCH_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 8 --warmup 1 --no-cpu-baseline > $O/bench_2ranks_one_device.json 2> $O/bench_2ranks.err; python -c "
import json; d=json.load(open('$O/bench_2ranks_one_device.json')); print('2 ranks on one device (validation):', d['n_gpus'], d['value'], d.get('ranks'))"
find $O -name "*.csv" -size +4M -delete; find $O -name "*.db" -size +20M -delete
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4final/bench_default.json'))
print('headline', d['value'], d['roofline']['kernel'], d['roofline']['frac_model'], d['roofline']['frac'], 'fast/exact', d.get('fast_over_exact'))
for c in d.get('configs', []):
    r=c.get('roofline') or {}
    print(c['config']['workload'][:40], c.get('value'), r.get('kernel'), r.get('frac_model'), r.get('frac'), r.get('frac_of_stream_copy'), 'job', r.get('job_counter_over_compulsory'), c.get('unavailable'))
print('api', d.get('pcie_inclusive_msamples_s'), d.get('api_1pass_plain_msamples_s'), d.get('api_1pass_msamples_s'), d.get('api_1pass_pinned_output_msamples_s'))
n=json.load(open('gpurun_out/r4final/bench_noflags.json'))
print('noflags', n['value'], n['steps'], [(c['config']['workload'][:20], c.get('value')) for c in n.get('configs', [])])
PY
