#!/usr/bin/env python3
"""Raw per-kernel sums of whatever counters one rocprofv3 --pmc run collected (counter_collection.csv), one row per counter, per wave cycle where that helps."""
import csv, collections, sys
c = collections.defaultdict(lambda: collections.defaultdict(float))
key = lambda s: s.split('(')[0].replace('void ', '').replace('tn::', '')
for r in csv.DictReader(open(sys.argv[1])):
    c[key(r['Kernel_Name'])][r['Counter_Name']] += float(r['Counter_Value'])
want = sys.argv[2] if len(sys.argv) > 2 else 'k_bounce'
for k, v in c.items():
    if not k.startswith(want): continue
    print("### " + k)
    for name, x in sorted(v.items()):
        print("%-28s %.4e" % (name, x))
