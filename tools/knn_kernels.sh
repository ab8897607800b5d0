#!/bin/bash
# per-kernel time of the KNN stage of the 250k pair (rocprofv3 kernel stats of tools/profile_eigsort.py): run on the GPU box
set -e
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $root/gpurun_out/trace_knn
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/trace_knn -- python3 $root/tools/profile_eigsort.py > $root/gpurun_out/knn_prof_stdout.txt 2>&1
cd $root
s=$(ls gpurun_out/trace_knn/*/*kernel_stats.csv | tail -1)
python3 - "$s" <<'PY'
import csv, sys
for r in csv.reader(open(sys.argv[1])):
    if any(k in r[0] for k in ("knn", "gather_rows", "cell", "coords_from", "extent", "make_grid")):
        print("%-70s calls %5s  avg %10.1f us" % (r[0][:70], r[1], float(r[3]) / 1e3))
PY
rm -rf gpurun_out/trace_knn
