#!/usr/bin/env python3
"""Idle time of the device inside a profiled run: gaps between consecutive dispatches of a rocprofv3 kernel trace
(`--kernel-trace --output-format csv`), summed by the kernel that FOLLOWS the gap.  python tools/trace_gaps.py TRACE.csv"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    # pfl::k_one<K, Pack<...>> / pfl::k_two<...>: the functor's name (x2: one launch for the two meshes of a pair)
    m = re.search(r"k_(one|two)<(?:\(anonymous namespace\)::|pfl::)?(\w+(?:<\w+>)?)", name)
    if m:
        return m.group(2) + (" x2" if m.group(1) == "two" else "")
    m = re.search(r"(k_[a-z0-9_]+(<[^>]*>)?)", name)
    if m:
        return m.group(1)
    m = re.search(r"(__amd_rocclr_\w+|radix_sort\w*|merge_sort\w*|\w*scan\w*|\w+)", name.replace("void ", ""))
    return m.group(1)[:40] if m else name[:40]

rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
if len(sys.argv) > 2:  # keep the last FRACTION of the dispatches (steady state: skip start-up and warm-up)
    rows = rows[int(len(rows) * (1.0 - float(sys.argv[2]))):]
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
gaps = defaultdict(lambda: [0, 0])
prev_end, prev_name = rows[0][1], rows[0][2]
for s, e, name in rows[1:]:
    g = s - prev_end
    if g > 0:
        key = (short(prev_name), short(name))
        gaps[key][0] += 1
        gaps[key][1] += g
    prev_end, prev_name = max(prev_end, e), name
print("span %.1f ms, busy %.1f ms (%.0f %%), %d dispatches" % (span / 1e6, busy / 1e6, 100.0 * busy / span, len(rows)))
for key, (cnt, tot) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
    print("%9.3f ms in %5d gaps (avg %7.1f us)  after %-42s before %s" % (tot / 1e6, cnt, tot / cnt / 1e3, key[0], key[1]))
