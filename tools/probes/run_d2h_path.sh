#!/bin/bash
# bash tools/probes/run_d2h_path.sh   (on the GPU box; binary built beforehand with hipcc, see d2h_path.hip)
set -e
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/d2h_path
rm -rf $out && mkdir -p $out
cd /tmp && export TMPDIR=/tmp
$root/tools/probes/d2h_path > $out/plain.txt 2>&1
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out/t -- $root/tools/probes/d2h_path > $out/under_rocprof.txt 2>&1
python3 - $(ls $out/t/*/*kernel_trace.csv | tail -1) $(ls $out/t/*/*memory_copy_trace.csv | tail -1) > $out/events.txt <<'PY'
import csv, sys
ev = []
for r in csv.DictReader(open(sys.argv[1])):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"]
    if "k_chase" in n or "copyBuffer" in n:
        ev.append((s, e, "kernel q%s %s" % (r["Queue_Id"], n[:40])))
for r in csv.DictReader(open(sys.argv[2])):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e - s > 50000:
        ev.append((s, e, "memcpy " + r["Direction"]))
ev.sort()
t0 = ev[0][0]
for s, e, n in ev:
    print(f"{(s - t0) / 1e3:12.1f} us {(e - s) / 1e3:9.1f} us  {n}")
PY
rm -rf $out/t
