#!/bin/bash
cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2m; mkdir -p $O
for sc in "glass 1920 1080 12 32" "large/ajax_standin 1920 1080 4 32"; do set -- $sc
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $O -o pmc_$(basename $1) --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --inner-pmc --scene $1 --width $2 --height $3 --maxdepth $4 --steps $5 > /dev/null 2> $O/pmc_$(basename $1).err
python - <<PY
import csv, collections
c=collections.defaultdict(lambda: collections.defaultdict(float)); t=collections.defaultdict(float)
for r in csv.DictReader(open("$O/pmc_$(basename $1)_counter_collection.csv")):
    n=r['Kernel_Name'].split('(')[0].replace('void ','').replace('tn::','')
    c[n][r['Counter_Name']]+=float(r['Counter_Value'])
for r in csv.DictReader(open("$O/pmc_$(basename $1)_kernel_trace.csv")):
    n=r['Kernel_Name'].split('(')[0].replace('void ','').replace('tn::','')
    t[n]+=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e9
print("== $1")
for k,v in sorted(c.items(), key=lambda kv:-t[kv[0]]):
    if not k.startswith('k_'): continue
    iv=v['SQ_INSTS_VALU']
    print("%-28s %7.2f ms  lanes %.0f%%  wait %.0f%%  VALU issue %.2f of peak  waves/SIMD %.1f" % (k, t[k]*1e3, 100*v['SQ_THREAD_CYCLES_VALU']/(64*iv) if iv else 0, 100*v['SQ_WAIT_ANY']/v['SQ_WAVE_CYCLES'], iv/t[k]/1.2288e12, 4*v['SQ_WAVE_CYCLES']/(v['GRBM_GUI_ACTIVE']/8*1024)))
PY
done
find $O -name "*.csv" -size +4M -delete
