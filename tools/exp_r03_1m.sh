#!/bin/bash
# kernel trace of the 1M / k=10 step (C5 on one GPU): where do assembly and eigsort go superlinear?
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r03_1m
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 700 rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 $root/tools/sweep.py 1000000 --cpu-max 0 > $out/sweep.txt 2>&1
tr=$(ls $out/trace/*/*kernel_trace.csv | tail -1)
python3 $root/tools/trace_timeline.py $tr > $out/timeline.txt
python3 - <<PY
import re
rows=[l for l in open("$out/timeline.txt")]
# stop at the first resident filter launch; then print eigsort/knn part
out=[]
for l in rows:
    out.append(l)
open("$out/timeline_head.txt","w").writelines(out[:140])
import collections
agg=collections.Counter(); cnt=collections.Counter()
for l in rows[1:]:
    m=re.match(r"\s*([\d.]+) ms\s+gap\s+([-\d.]+) us\s+(\d+) x (\S+)\s+busy\s+([\d.]+) us",l)
    if m:
        agg[m.group(4)]+=float(m.group(5)); cnt[m.group(4)]+=int(m.group(3))
with open("$out/per_kernel.txt","w") as f:
    for k,v in agg.most_common(45): f.write("%10.1f us %5d x %s\n"%(v,cnt[k],k))
PY
rm -rf $out/trace
