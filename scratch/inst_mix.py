#!/usr/bin/env python3
"""Per-kernel dynamic instruction mix from two rocprofv3 --pmc passes (counter_collection.csv each): the share of the VALU wave-instructions
that are fp32 add / mul / fma / transcendental, fp64 (add / mul / fma / transcendental), integer, conversions; SALU, LDS, VMEM beside them."""
import csv, collections, sys
c = collections.defaultdict(lambda: collections.defaultdict(float))
key = lambda s: s.split('(')[0].replace('void ', '').replace('tn::', '')
seen_valu = collections.defaultdict(int)
for path in sys.argv[1:]:
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(path)):
        per[key(r['Kernel_Name'])][r['Counter_Name']] += float(r['Counter_Value'])
    for k, v in per.items():
        for name, val in v.items():
            if name == 'SQ_INSTS_VALU':
                seen_valu[k] += 1
                c[k][name] += val
            else:
                c[k][name] = val
print("| kernel | VALU wave-insts | fp32 add | fp32 mul | fp32 fma | fp32 trans | fp64 add+mul+fma | fp64 trans | int32 | int64 | cvt | other VALU | SALU / VALU | LDS / VALU | VMEM / VALU |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
for k, v in sorted(c.items(), key=lambda kv: -kv[1].get('SQ_INSTS_VALU', 0)):
    if not k.startswith('k_'): continue
    iv = v['SQ_INSTS_VALU']/max(1, seen_valu[k])
    if iv <= 0: continue
    g = lambda n: v.get(n, 0.0)
    f64 = g('SQ_INSTS_VALU_ADD_F64') + g('SQ_INSTS_VALU_MUL_F64') + g('SQ_INSTS_VALU_FMA_F64')
    known = g('SQ_INSTS_VALU_ADD_F32') + g('SQ_INSTS_VALU_MUL_F32') + g('SQ_INSTS_VALU_FMA_F32') + g('SQ_INSTS_VALU_TRANS_F32') + f64 + g('SQ_INSTS_VALU_TRANS_F64') + g('SQ_INSTS_VALU_INT32') + g('SQ_INSTS_VALU_INT64') + g('SQ_INSTS_VALU_CVT')
    p = lambda x: "%.1f %%" % (100.0*x/iv)
    print("| %s | %.3e | %s | %s | %s | %s | %s | %s | %s | %s | %s | %s | %.2f | %.3f | %.3f |" % (k, iv, p(g('SQ_INSTS_VALU_ADD_F32')), p(g('SQ_INSTS_VALU_MUL_F32')), p(g('SQ_INSTS_VALU_FMA_F32')),
          p(g('SQ_INSTS_VALU_TRANS_F32')), p(f64), p(g('SQ_INSTS_VALU_TRANS_F64')), p(g('SQ_INSTS_VALU_INT32')), p(g('SQ_INSTS_VALU_INT64')), p(g('SQ_INSTS_VALU_CVT')), p(iv - known),
          g('SQ_INSTS_SALU')/iv, g('SQ_INSTS_LDS')/iv, g('SQ_INSTS_VMEM')/iv))
