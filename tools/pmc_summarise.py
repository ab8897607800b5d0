#!/usr/bin/env python3
"""Average a PMC counter per kernel from rocprofv3 --pmc CSV output.
   python tools/pmc_summarise.py DIR COUNTER [KERNEL_SUBSTRING]   -> prints JSON {kernel: {calls, avg}}"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

d, counter = sys.argv[1], sys.argv[2]
want = sys.argv[3] if len(sys.argv) > 3 else ""
acc = defaultdict(lambda: [0, 0.0])
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if row.get("Counter_Name") != counter:
                continue
            name = row.get("Kernel_Name", "")
            if want and want not in name:
                continue
            a = acc[name[:80]]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
print(json.dumps({k: {"calls": v[0], "avg": v[1] / max(v[0], 1)} for k, v in acc.items()}, indent=1))
