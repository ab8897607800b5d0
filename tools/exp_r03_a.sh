#!/bin/bash
# round-3 experiment A: what would conflict-free LDS placement / free hand-offs buy the resident kernel (timing only)
set -e
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r03_a
mkdir -p $out
cd $root
V=$root/pyfocusr_amd/csrc/variants
for name in base fake nopoll fakenopoll; do
  if [ $name = base ]; then unset PYFOCUSR_HIP_LIB; else export PYFOCUSR_HIP_LIB=$V/libpyfocusr_hip_$name.so; fi
  echo "== $name" >> $out/cheb.txt
  timeout -k 10 200 python3 tools/bench_cheb.py 250000 --modes 1 >> $out/cheb.txt 2>&1
done
unset PYFOCUSR_HIP_LIB
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --output-format csv -d $out/pmc_lds -- python3 $root/tools/bench_cheb.py 250000 --modes 1 --reps 5 > $out/pmc_lds.txt 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("$out/pmc_lds/*/*counter_collection.csv")[0]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$out/pmc_lds_summary.txt","w") as o:
    for k,d in acc.items():
        if "resident" in k:
            for c,v in d.items(): o.write("%s %s n=%d avg=%.1f\n"%(k,c,len(v),sum(v)/len(v)))
PY
rm -rf $out/pmc_lds
