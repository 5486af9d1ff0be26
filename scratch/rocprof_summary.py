#!/usr/bin/env python3
"""Dump the per-kernel statistics of a rocprofv3 --kernel-trace --stats run (rocpd sqlite) as markdown."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
print("| kernel | calls | total_us | avg_us | % |")
print("|---|---|---|---|---|")
for n, c, t, a, p in rows:
    n = n.split('(')[0].replace('void ', '')
    print("| %s | %d | %.1f | %.1f | %.2f |" % (n, c, t, a, p))
