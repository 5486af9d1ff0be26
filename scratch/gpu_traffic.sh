#!/bin/bash
# HBM traffic per kernel (FETCH_SIZE and WRITE_SIZE in separate passes, the guide's gfx950 x2 on FETCH_SIZE) of one scene for
# one or more builds of the library: gpu_traffic.sh "<scene> <W> <H> <depth> <passes>" lib_a [lib_b ...]
cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/traffic; mkdir -p $O
set -- $1 "${@:2}"
SC=$1; W=$2; H=$3; D=$4; P=$5; shift 5
for L in "$@"; do
export TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/$L
T=$(basename $L .so)_$(basename $SC)
for C in FETCH_SIZE WRITE_SIZE; do
timeout 300 rocprofv3 --kernel-trace --pmc $C -d $O -o ${T}_$C --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --inner-pmc --scene $SC --width $W --height $H --maxdepth $D --steps $P > /dev/null 2> $O/${T}_$C.err
done
python - <<PY
import csv, collections
b=collections.defaultdict(lambda: collections.defaultdict(float)); t=collections.defaultdict(float); n=collections.defaultdict(int)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for r in csv.DictReader(open("$O/${T}_%s_counter_collection.csv" % c)):
        k=r['Kernel_Name'].split('(')[0].replace('void ','').replace('tn::','')
        b[k][c]+=float(r['Counter_Value'])
for r in csv.DictReader(open("$O/${T}_FETCH_SIZE_kernel_trace.csv")):
    k=r['Kernel_Name'].split('(')[0].replace('void ','').replace('tn::','')
    t[k]+=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e9; n[k]+=1
print("== $L  $SC ${W}x$H maxDepth $D, $P passes: HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB")
tot=0
for k in sorted(b, key=lambda k:-t[k]):
    if not k.startswith('k_'): continue
    by=(2*b[k]['FETCH_SIZE']+b[k]['WRITE_SIZE'])*1024; tot+=by
    print("%-30s %4d launches %8.2f ms  %8.2f GB  %6.0f GB/s" % (k, n[k], t[k]*1e3, by/1e9, by/t[k]/1e9 if t[k] else 0))
print("total %.2f GB" % (tot/1e9))
PY
done
find $O -name "*.csv" -delete
