#!/bin/bash
# round 2 evidence run: tests, default bench (live PMC), rocprofv3 --stats of the same command, per-kernel PMC tables,
# one bench line per BASELINE config in both arithmetic arms, micro-benchmarks, k_walk section profile
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2ev; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed" ) > $O/pytest_gpu.log; cat $O/pytest_gpu.log
( time timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O -o stats -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-fast --no-api > $O/bench_under_stats.json 2> $O/stats.err
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
timeout 300 rocprofv3 --kernel-trace --pmc $SQ -d $O -o pmc_cornell --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --inner-pmc --scene cornell --width 1024 --height 1024 --maxdepth 4 --steps 20 > /dev/null 2> $O/pmc_cornell.err
timeout 300 rocprofv3 --kernel-trace --pmc $SQ -d $O -o pmc_ajax --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --inner-pmc --scene large/ajax_standin --width 1920 --height 1080 --maxdepth 4 --steps 20 > /dev/null 2> $O/pmc_ajax.err
cd $GRAFT_REPO_ROOT
python scratch/rocprof_summary.py $(ls $O/*stats*.db 2>/dev/null | head -1) > $O/kernel_stats.md 2>&1
python scratch/pmc_table.py $O/pmc_cornell_counter_collection.csv $O/pmc_cornell_kernel_trace.csv > $O/pmc_cornell.md 2>&1
python scratch/pmc_table.py $O/pmc_ajax_counter_collection.csv $O/pmc_ajax_kernel_trace.csv > $O/pmc_ajax.md 2>&1
run() { timeout 600 python bench.py "$@" --no-cpu-baseline --no-pmc --no-second-config 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
r=d['roofline']; f=d.get('fast') or {}
print('| %s | %.1f | %.1f | %.2f | %s | %s | %s |' % (d['config']['workload'].split(',')[0], d['value'], d['mrays_per_s'], d['config']['rays_per_sample'], r['kernel_ms'], ('%.1f' % f['msamples_s']) if f.get('msamples_s') else '-', ('%.2e' % f['l2_vs_exact_at_spp'][0]) if f.get('l2_vs_exact_at_spp') else '-'))
PY
}
( echo "| config | Msamples/s (exact) | Mrays/s | rays/sample | kernel ms of one timed block | Msamples/s (fast arm) | fast-vs-exact L2 @ 256 spp |"; echo "|---|---|---|---|---|---|---|"
run --scene cornell --width 256 --height 256 --steps 16 --warmup 2
run --scene cornell --steps 64 --warmup 8
run --scene large/ajax_standin --width 1920 --height 1080 --steps 32 --warmup 2
run --scene glass --width 1920 --height 1080 --maxdepth 12 --steps 32 --warmup 2
run --scene veach --width 3840 --height 2160 --steps 8 --warmup 1
run --scene large/env_loft --width 1024 --height 512 --steps 64 --warmup 2
run --scene many_spheres --width 1024 --height 768 --steps 64 --warmup 2 ) > $O/configs.md 2>&1
cat $O/configs.md
( cd scratch/ubench; ./gather_bench 524288 6 1; ./gather_bench 524288 6 8 | head -1; ./gather_bench 4194304 6 1 | head -1; ./valu_bench ) > $O/ubench.txt 2>&1
TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/libtinsel_hip_walkprof.so python scratch/walk_prof.py large/ajax_standin 1920 1080 4 32 2>&1 | grep -v amdgpu.ids > $O/walk_profile.txt
find $O -name "*.csv" -size +4M -delete; find $O -name "*.db" -size +20M -delete
ls $O
