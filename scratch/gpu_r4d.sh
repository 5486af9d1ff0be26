#!/bin/bash
# round 4, call D: flat-scan box tests with hardware min/max in all-finite waves (A/B against -DTN_FLAT_MINMAX=0), parity, the N-rank
# launch paths of bench.py, and the default bench line with the per-kernel table
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4d; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_fuzz.py tests/test_gpu_swalk.py -x -q 2>&1 | tail -15 ) > $O/pytest_parity.log 2>&1; tail -5 $O/pytest_parity.log
( time timeout 900 python -m pytest tests/test_gpu_multirank.py -x -q 2>&1 | tail -30 ) > $O/pytest_multirank.log 2>&1; tail -8 $O/pytest_multirank.log
run() { timeout 120 python bench.py "$@" --no-cpu-baseline --no-pmc --no-second-config --no-api --no-fast --no-ubench 2>/tmp/err.txt > /tmp/b.json || tail -3 /tmp/err.txt
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print('| $TAG | %s | %.1f | %s |' % (d['config']['workload'].split(',')[0], d['value'], d['roofline']['kernel_ms']), flush=True)
PY
}
ab() { local S="$1"; shift; ( [ "$S" != "-" ] && export $S; TAG="$S" run "$@" ); }
NOMM="TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/libtinsel_hip_nominmax.so"
( echo "| environment | config | Msamples/s | kernel ms of one timed block |"; echo "|---|---|---|---|"
for S in "$NOMM" "-" "$NOMM" "-"; do ab "$S" --scene cornell --steps 20 --warmup 5; done
for S in "$NOMM" "-" "$NOMM" "-"; do ab "$S" --scene veach --width 3840 --height 2160 --steps 8 --warmup 1; done
for S in "$NOMM" "-"; do ab "$S" --scene glass --width 1920 --height 1080 --maxdepth 12 --steps 20 --warmup 2; done
for S in "$NOMM" "-"; do ab "$S" --scene features --width 1920 --height 1080 --maxdepth 6 --steps 16 --warmup 1; done
for S in "$NOMM" "-"; do ab "$S" --scene cornell --width 256 --height 256 --steps 16 --warmup 4; done
) 2>&1 | sed "s#$GRAFT_REPO_ROOT/##" | tee $O/ab_flat_minmax.md
( time timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4d/bench_default.json'))
print('headline', d['value'], d['roofline']['kernel'], d['roofline']['frac_model'], d['roofline']['frac'], 'fast/exact', d.get('fast_over_exact'), 'job ratio', d['roofline'].get('job_counter_over_compulsory'))
for c in d.get('configs', []):
    r=c.get('roofline') or {}
    print(c['config']['workload'][:44], c.get('value'), r.get('kernel'), r.get('frac_model'), r.get('frac'), 'job', r.get('job_counter_over_compulsory'), c.get('unavailable'))
    for k in (r.get('kernels') or [])[:6]:
        print('   ', k['kernel'], k['ms'], k['frac_model'], k['frac'], 'GB/s', k['counter_GBs'], 'copy', k['frac_of_stream_copy'], 'lanes', k['valu_lanes_active'], 'waves', k['waves_per_simd'], 'valu', k['valu_frac_of_issue_peak'])
PY
