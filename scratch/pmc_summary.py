#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel: sum of each counter over dispatches."""
import csv, glob, sys, collections, re
files = sys.argv[1:]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for f in files:
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = re.sub(r'\(.*', '', row['Kernel_Name']).replace('void ', '').replace('tn::', '')
            agg[k][row['Counter_Name']] += float(row['Counter_Value'])
            disp[k].add((f, row['Dispatch_Id']))
for k in sorted(agg):
    print("## %s  (dispatches %d)" % (k, len(disp[k])//max(1,len(files))))
    for c in sorted(agg[k]):
        print("   %-28s %18.0f" % (c, agg[k][c]))
