#!/bin/bash
# LDS bank conflicts of the resident kernel (250k pair), by counters: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
set -e
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/n_lds
rm -rf $out && mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $out/pmc -- python3 $root/bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline > $out/bench.json 2> $out/rocprof.err
cd $root
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/n_lds/pmc/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'][:60]
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'SQ_LDS_IDX_ACTIVE': cnt[k] += 1
rows = sorted(acc.items(), key=lambda kv: -kv[1].get('SQ_LDS_IDX_ACTIVE', 0))[:12]
with open('gpurun_out/n_lds/summary.md', 'w') as o:
    o.write('| kernel | dispatches | SQ_LDS_IDX_ACTIVE | SQ_LDS_BANK_CONFLICT | conflict share | GRBM_GUI_ACTIVE |\n|---|---|---|---|---|---|\n')
    for k, v in rows:
        a, c = v.get('SQ_LDS_IDX_ACTIVE', 0), v.get('SQ_LDS_BANK_CONFLICT', 0)
        o.write('| %s | %d | %.3g | %.3g | %.2f | %.3g |\n' % (k, cnt[k], a, c, c / a if a else 0, v.get('GRBM_GUI_ACTIVE', 0)))
print(open('gpurun_out/n_lds/summary.md').read())
PY
rm -rf $out/pmc
