#!/bin/bash
# round 5 evidence run on the tree as committed: GPU suite, the driver's bench line (small contract line + bench_detail.json: live PMC +
# calibration, per-kernel table), rocprofv3 --kernel-trace --stats of the same command, per-kernel SQ tables of the BASELINE configs, one
# bench line per config in both arithmetic arms, k_walk's section profile, bench.py without flags, the N-rank legs and the group mode on one device
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5final; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -m gpu -q -s 2>&1 | grep -aE "passed|failed|per-pixel L2|rebuilt on the device" ) > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
( time timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real
cp bench_detail.json $O/bench_detail_default.json; wc -c $O/bench_default.json
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O -o stats -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-fast --no-api --no-ubench > $O/bench_under_stats.json 2> $O/stats.err
cd $GRAFT_REPO_ROOT
python scratch/rocprof_summary.py $(ls $O/*stats*.db 2>/dev/null | head -1) > $O/kernel_stats.md 2>&1
bash scratch/gpu_pmc_kernels.sh $O configs "cornell 1024 1024 4 20" "large/ajax_standin 1920 1080 4 20" "glass 1920 1080 12 20" "veach 3840 2160 4 20" "many_spheres 1024 768 4 64" > /dev/null
run() { timeout 400 python bench.py "$@" --no-cpu-baseline --no-pmc --no-second-config --no-more-configs --no-ubench --no-api 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('bench_detail.json'))
r=d['roofline']; f=d.get('fast') or {}
print('| %s | %.1f | %.1f | %.2f | %s | %s | %s |' % (d['config']['workload'].split(',')[0], d['value'], d['mrays_per_s'], d['config']['rays_per_sample'], r['kernel_ms'], ('%.1f' % f['msamples_s']) if f.get('msamples_s') else '-', ('%.2e @ %d' % tuple(f['l2_vs_exact_at_spp'])) if f.get('l2_vs_exact_at_spp') else '-'))
PY
}
( echo "| config | Msamples/s (exact) | Mrays/s | rays/sample | kernel ms of the first timed block | Msamples/s (fast arm) | fast-vs-exact L2 @ spp |"; echo "|---|---|---|---|---|---|---|"
run --scene cornell --width 256 --height 256 --steps 16 --warmup 4
run --scene cornell --steps 20 --warmup 5
run --scene cornell --steps 64 --warmup 8
run --scene large/ajax_standin --width 1920 --height 1080 --steps 20 --warmup 2
run --scene glass --width 1920 --height 1080 --maxdepth 12 --steps 20 --warmup 2
run --scene veach --width 3840 --height 2160 --steps 20 --warmup 2
run --scene large/env_loft --width 1024 --height 512 --steps 64 --warmup 2
run --scene many_spheres --width 1024 --height 768 --steps 64 --warmup 2
run --scene features --width 1920 --height 1080 --maxdepth 6 --steps 16 --warmup 1
run --scene motionblur --width 1920 --height 1080 --steps 16 --warmup 1
run --scene large/table --width 1920 --height 1080 --steps 16 --warmup 1
run --scene large/transmission --width 1920 --height 1080 --steps 16 --warmup 1
run --scene gloss --steps 64 --warmup 8 ) > $O/configs.md 2>&1
cat $O/configs.md
TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/libtinsel_hip_walkprof.so python scratch/walk_prof.py large/ajax_standin 1920 1080 4 20 2>&1 | grep -v amdgpu.ids > $O/walk_profile.txt
TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/libtinsel_hip_walkprof.so python scratch/walk_prof.py glass 1920 1080 12 20 2>&1 | grep -v amdgpu.ids >> $O/walk_profile.txt
cat $O/walk_profile.txt
( time timeout 600 python bench.py > $O/bench_noflags.json 2> $O/bench_noflags.err ) 2>&1 | grep real
python bench.py --group --gpus 2 --steps 20 --warmup 2 > $O/bench_group2_one_device.json 2> $O/bench_group.err; cat $O/bench_group2_one_device.json
TINSEL_BENCH_BACKEND=gloo TINSEL_BENCH_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 8 --warmup 1 > $O/bench_2ranks_one_device.json 2> $O/bench_2ranks.err; wc -c $O/bench_2ranks_one_device.json; cat $O/bench_2ranks_one_device.json | cut -c1-1500
find $O -name "*.csv" -size +4M -delete; find $O -name "*.db" -size +20M -delete
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5final/bench_default.json'))
print('headline', d['value'], d['roofline'], 'fast/exact', d.get('fast_over_exact'), d.get('fast_l2_at_spp'), d.get('api_msamples_s'))
print('cpu', d['cpu_baseline'])
for c in d.get('configs', []): print(c)
n=json.load(open('gpurun_out/r5final/bench_noflags.json'))
print('noflags', n['value'], n['steps'], [(c['workload'][:20], c.get('value')) for c in n.get('configs', [])])
PY
