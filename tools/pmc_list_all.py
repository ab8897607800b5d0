#!/usr/bin/env python3
"""Every kernel of a PMC pass pair with its average traffic per dispatch (the listing that found the 1-NN kernel's
scratch writes):  python tools/pmc_list_all.py FETCH_DIR WRITE_DIR [KERNEL_STATS_CSV]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def table(directory, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") == counter:
                    a = acc[row.get("Kernel_Name", "")]
                    a[0] += 1
                    a[1] += float(row["Counter_Value"])
    return acc


fe, wr = table(sys.argv[1], "FETCH_SIZE"), table(sys.argv[2], "WRITE_SIZE")
stats = {}
if len(sys.argv) > 3:
    for row in csv.DictReader(open(sys.argv[3])):
        stats[row["Name"]] = float(row["AverageNs"])
rows = []
for name, (n, kb) in fe.items():
    wn, wkb = wr.get(name, [0, 0.0])
    rd, wt = 2.0 * kb * 1024 / max(n, 1), wkb * 1024 / max(wn, 1)
    rows.append((n * (rd + wt), name, n, rd, wt, stats.get(name)))
rows.sort(reverse=True)
print("| kernel | dispatches | read MB / dispatch | written MB / dispatch | us (kernel trace) | GB/s |")
print("|---|---|---|---|---|---|")
for _, name, n, rd, wt, ns in rows[:45]:
    short = name.replace("(anonymous namespace)::", "").split("(")[0][:60]
    print("| %s | %d | %.2f | %.2f | %s | %s |" % (short, n, rd / 1e6, wt / 1e6, "%.1f" % (ns / 1e3) if ns else "", "%.0f" % ((rd + wt) / ns) if ns else ""))
