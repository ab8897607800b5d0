#!/usr/bin/env python3
"""Timeline of the LAST bench step in a rocprofv3 kernel trace (`--kernel-trace --output-format csv`): every dispatch
with its start offset, duration and the idle gap before it; runs of the same kernel are folded.
python tools/trace_timeline.py TRACE.csv [first-kernel-of-a-step substring, default k_bbox]"""
import csv
import re
import sys


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
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
rows.sort()
marker = sys.argv[2] if len(sys.argv) > 2 else "k_bbox"
starts = [i for i, r in enumerate(rows) if marker in r[2] and (i == 0 or marker not in rows[i - 1][2])]
# a step assembles two meshes: in one chain of shared launches (the marker once per step, "x2"), or one after / beside the
# other (twice)
per_step = 1 if starts and rows[starts[-1]][2].endswith("x2") else 2
begin = starts[-per_step] if len(starts) >= per_step else 0
while begin > 0 and rows[begin][0] - rows[begin - 1][1] < 20000 and "fill_words" in rows[begin - 1][2]:
    begin -= 1
while begin > 0 and rows[begin][0] - rows[begin - 1][1] < 100000 and "copyBuffer" in rows[begin - 1][2]:
    begin -= 1
step = rows[begin:]
t0 = step[0][0]
print("last step: %d dispatches, span %.2f ms, busy %.2f ms" % (len(step), (step[-1][1] - t0) / 1e6, sum(e - s for s, e, _ in step) / 1e6))
i = 0
prev_end = t0
while i < len(step):
    j = i
    gap_in, busy = 0, 0
    while j < len(step) and step[j][2] == step[i][2] and (j == i or step[j][0] - step[j - 1][1] < 15000):
        if j > i:
            gap_in += max(0, step[j][0] - step[j - 1][1])
        busy += step[j][1] - step[j][0]
        j += 1
    gap = step[i][0] - prev_end
    print("%9.3f ms  gap %7.1f us  %4d x %-44s busy %8.1f us  (inner gaps %6.1f us)" % ((step[i][0] - t0) / 1e6, gap / 1e3, j - i, step[i][2], busy / 1e3, gap_in / 1e3))
    prev_end = max(e for _, e, _ in step[i:j])
    i = j
