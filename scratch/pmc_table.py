#!/usr/bin/env python3
"""Per-kernel table from one rocprofv3 --kernel-trace --pmc run (counter_collection.csv + kernel_trace.csv):
time, VALU lanes active, wave cycles waiting, VALU issue rate vs the 2-cycle peak, resident waves per SIMD."""
import csv, collections, sys
cc, kt = sys.argv[1], sys.argv[2]
c = collections.defaultdict(lambda: collections.defaultdict(float)); t = collections.defaultdict(float); n = collections.defaultdict(int)
key = lambda s: s.split('(')[0].replace('void ', '').replace('tn::', '')
for r in csv.DictReader(open(cc)):
    c[key(r['Kernel_Name'])][r['Counter_Name']] += float(r['Counter_Value'])
for r in csv.DictReader(open(kt)):
    t[key(r['Kernel_Name'])] += (int(r['End_Timestamp']) - int(r['Start_Timestamp']))/1e9; n[key(r['Kernel_Name'])] += 1
print("| kernel | launches | total ms | VALU lanes active | wave cycles waiting | VALU wave-inst/s (of 1.2288e12) | waves/SIMD | SQ_INSTS_VALU |")
print("|---|---|---|---|---|---|---|---|")
for k, v in sorted(c.items(), key=lambda kv: -t[kv[0]]):
    if not k.startswith('k_') or t[k] <= 0: continue
    iv = v.get('SQ_INSTS_VALU', 0.0)
    print("| %s | %d | %.2f | %.0f %% | %.0f %% | %.3e (%.2f) | %.1f | %.3e |" % (k, n[k], t[k]*1e3, 100*v.get('SQ_THREAD_CYCLES_VALU', 0)/(64*iv) if iv else 0,
          100*v.get('SQ_WAIT_ANY', 0)/max(1.0, v.get('SQ_WAVE_CYCLES', 1)), iv/t[k], iv/t[k]/1.2288e12, 4*v.get('SQ_WAVE_CYCLES', 0)/max(1.0, v.get('GRBM_GUI_ACTIVE', 8)/8*1024), iv))
