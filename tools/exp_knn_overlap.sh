#!/bin/bash
# How long the main 1-NN kernel runs with the eigenvector downloads beside it (PF_DOWNLOAD_DEFER=1, the default) and
# without (=0): kernel trace of 6 steps each, the rows of k_knn_coop / copyBuffer kept.   bash tools/exp_knn_overlap.sh
set -e
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/knn_overlap
rm -rf $out && mkdir -p $out
cd /tmp && export TMPDIR=/tmp
args="--steps 6 --warmup 2 --no-extras --no-cpu-baseline"
for defer in 1 0; do
    PF_DOWNLOAD_DEFER=$defer rocprofv3 --kernel-trace --output-format csv -d $out/t$defer -- python3 $root/bench.py $args > $out/bench_$defer.json 2> $out/err_$defer.txt
    f=$(ls $out/t$defer/*/*kernel_trace.csv | tail -1)
    python3 - "$f" > $out/rows_$defer.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
t0 = min(int(r["Start_Timestamp"]) for r in rows)
for r in rows:
    n = r["Kernel_Name"]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if "k_knn_coop" in n or ("copyBuffer" in n and e - s > 100000):
        print(f"{(s - t0) / 1e3:12.1f} us  {(e - s) / 1e3:9.1f} us  q{r['Queue_Id']}  {n[:60]}")
PY
    rm -rf $out/t$defer
done
for i in 1 2 3; do
    for defer in 1 0; do
        PF_DOWNLOAD_DEFER=$defer PF_BENCH_DETAIL=1 python3 $root/bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $out/plain_${defer}_$i.json 2>> $out/err_plain.txt
    done
done
