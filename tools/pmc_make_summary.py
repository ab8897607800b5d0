#!/usr/bin/env python3
"""profiles/r02_pmc_summary.json from rocprofv3 output directories (run on the GPU box, see profiles/README.md):

    python tools/pmc_make_summary.py --fetch DIR --write DIR [--fetch-stream DIR --write-stream DIR] \\
        --stats KERNEL_STATS_CSV --bench BENCH_JSON_OF_THE_PROFILED_RUN --out profiles/r02_pmc_summary.json

FETCH_SIZE / WRITE_SIZE come from SEPARATE --pmc passes (MI355X_MICROARCH.md: they do not fit one pass) and are in KB;
on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes, so read bytes = 2 x FETCH_SIZE (same guide, HBM section;
calibrated there for wide streaming loads - the 8-byte agent-scope hand-off accesses of the resident kernel are outside
that calibration, which the summary says)."""
import argparse
import csv
import glob
import json
import os
from collections import defaultdict


def counter_avg(directory, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                a = acc[row.get("Kernel_Name", "")]
                a[0] += 1
                a[1] += float(row["Counter_Value"])
    return {k: (v[0], v[1] / max(v[0], 1)) for k, v in acc.items()}


def pick(table, needle):
    hits = [(k, v) for k, v in table.items() if needle in k]
    if not hits:
        return None, (0, 0.0)
    return max(hits, key=lambda kv: kv[1][0])


ap = argparse.ArgumentParser()
ap.add_argument("--fetch", required=True)
ap.add_argument("--write", required=True)
ap.add_argument("--fetch-stream")
ap.add_argument("--write-stream")
ap.add_argument("--stats")
ap.add_argument("--bench")
ap.add_argument("--out", required=True)
args = ap.parse_args()

stats = {}
if args.stats and os.path.exists(args.stats):
    with open(args.stats) as fh:
        for row in csv.DictReader(fh):
            stats[row["Name"]] = dict(calls=int(row["Calls"]), avg_ns=float(row["AverageNs"]), pct=float(row["Percentage"]))
bench = {}
if args.bench and os.path.exists(args.bench):
    with open(args.bench) as fh:
        for line in fh:
            if line.startswith("{"):
                bench = json.loads(line)
kernels = {}
for label, needle, fdir, wdir in (("k_cheb_resident<2, 1, 8, true>", "k_cheb_resident<2, 1, 8, true>", args.fetch, args.write),
                                  ("k_sell_op2<true>", "k_sell_op2<true>", args.fetch_stream, args.write_stream)):
    if not fdir or not wdir:
        continue
    fname, (fcalls, fkb) = pick(counter_avg(fdir, "FETCH_SIZE"), needle)
    wname, (wcalls, wkb) = pick(counter_avg(wdir, "WRITE_SIZE"), needle)
    if fname is None or wname is None:
        continue
    entry = dict(dispatches_counted_fetch=fcalls, dispatches_counted_write=wcalls, FETCH_SIZE_avg_KB=fkb, WRITE_SIZE_avg_KB=wkb,
                 hbm_read_bytes_per_launch=2.0 * fkb * 1024.0, hbm_write_bytes_per_launch=wkb * 1024.0,
                 hbm_bytes_per_launch=2.0 * fkb * 1024.0 + wkb * 1024.0,
                 correction="read bytes = 2 x FETCH_SIZE x 1024 (gfx950: 128-B requests tallied at 64 B), WRITE_SIZE exact "
                            "(MI355X_MICROARCH.md, HBM)")
    st = next((v for k, v in stats.items() if needle in k), None)
    if st:
        entry.update(kernel_trace_calls=st["calls"], kernel_trace_avg_ns=st["avg_ns"], kernel_trace_pct_of_device_time=st["pct"])
    if label.startswith("k_cheb_resident") and bench.get("roofline", {}).get("steps_per_launch"):
        spl = bench["roofline"]["steps_per_launch"]
        entry.update(steps_per_launch_avg=spl, hbm_bytes_per_step=entry["hbm_bytes_per_launch"] / spl,
                     hbm_bytes_per_launch_fixed=0.0, lds_bytes_per_launch=bench["roofline"].get("lds_bytes_per_launch"),
                     note="per step of the pair; includes the once-per-launch load of the operators into registers spread over "
                          "the launch's steps; hand-off accesses are 8-byte agent-scope loads / stores (outside the guide's "
                          "calibration of FETCH_SIZE, which is for wide streaming reads)")
    kernels[label] = entry
# the other kernels of the step, from the same two passes: average traffic per dispatch, for comparison with their
# algorithmic bytes
others = {}
if args.fetch and args.write:
    ftab, wtab = counter_avg(args.fetch, "FETCH_SIZE"), counter_avg(args.write, "WRITE_SIZE")
    for needle in ("k_orth_dots", "k_orth_project", "k_knn_coop", "k_count_edges", "k_scatter_edges", "k_vec_apply", "k_combine",
                   "k_fill_sell_entries"):
        fname, (fcalls, fkb) = pick(ftab, needle)
        wname, (wcalls, wkb) = pick(wtab, needle)
        if fname is None or wname is None:
            continue
        e = dict(dispatches_counted_fetch=fcalls, dispatches_counted_write=wcalls, hbm_read_bytes_per_launch=2.0 * fkb * 1024.0,
                 hbm_write_bytes_per_launch=wkb * 1024.0)
        st = next((v for k, v in stats.items() if needle in k), None)
        if st:
            e.update(kernel_trace_calls=st["calls"], kernel_trace_avg_ns=st["avg_ns"],
                     GBps_of_counter_traffic=(e["hbm_read_bytes_per_launch"] + e["hbm_write_bytes_per_launch"]) / st["avg_ns"])
        others[needle] = e
# per stage of a step: counter traffic of every kernel that belongs to the stage's source files, summed over the run and
# divided by the steps the profiled command ran (timed + warm-up) - what bench.py's `roofline_assembly` / `roofline_knn`
# entries quote as `traffic`
import re

repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernels_of(*files):
    names = set()
    for f in files:
        with open(os.path.join(repo, "pyfocusr_amd", "csrc", f)) as fh:
            text = fh.read()
            names.update(re.findall(r"__global__[^;{]*?void\s+(\w+)\s*\(", text))
            names.update(re.findall(r"struct\s+(\w+)\s*\{\s*static constexpr int BOUNDS", text))  # functors of pf_launch.h
    return names


def kernel_of(kname):
    """The kernel's (or, for pfl::k_one / k_two launches, the functor's) plain name, and how many meshes the launch serves."""
    k = kname.replace("(anonymous namespace)::", "")
    m = re.search(r"k_(one|two)<(?:pfl::)?(\w+)", k)
    if m:
        return m.group(2), (2 if m.group(1) == "two" else 1)
    m = re.search(r"(\w+)(<[^(]*>)?\(", k)
    return (m.group(1) if m else None), 1


stage_files = dict(assembly=("pf_assemble.hip", "pf_reorder.hip", "pf_scan.hip", "pf_launch.h"), knn=("pf_knn.hip", "pf_knn_tree.hip"))
stages = {}
if args.fetch and args.write:
    ftab, wtab = counter_avg(args.fetch, "FETCH_SIZE"), counter_avg(args.write, "WRITE_SIZE")
    # steps the PROFILED command ran (timed + warm-up + any extra untimed step): every step assembles two meshes
    meshes = sum(calls * kernel_of(kname)[1] for kname, (calls, kb) in ftab.items() if "k_count_edges" in kname)
    n_steps = max(meshes // 2, 1) if meshes else max(int(bench.get("steps", 2)) + int(bench.get("warmup", 1)), 1)
    for stage, files in stage_files.items():
        names = kernels_of(*files)
        rd = wr = 0.0
        n_disp = 0
        for kname, (calls, kb) in ftab.items():
            if kernel_of(kname)[0] in names:
                rd += 2.0 * kb * 1024.0 * calls
                n_disp += calls
        for kname, (calls, kb) in wtab.items():
            if kernel_of(kname)[0] in names:
                wr += kb * 1024.0 * calls
        stages[stage] = dict(hbm_read_bytes_per_step=rd / n_steps, hbm_write_bytes_per_step=wr / n_steps,
                             hbm_bytes_per_step=(rd + wr) / n_steps, dispatches_per_step=n_disp / n_steps, steps_counted=n_steps,
                             kernels_from=list(files))
out = dict(source="rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python3 bench.py --steps 2 --warmup 1 "
                  "--no-extras --no-cpu-baseline (250k-vertex pair, k=5); streaming kernel: the same with PF_PERSIST=0",
           kernels=kernels, other_kernels_average_per_dispatch=others, stages=stages)
with open(args.out, "w") as fh:
    json.dump(out, fh, indent=1)
print(json.dumps(out, indent=1))
