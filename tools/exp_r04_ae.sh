#!/bin/bash
set -e
for v in 1 0; do
echo "## PF_DOWNLOAD_DEFER=$v"
PF_DOWNLOAD_DEFER=$v python - <<'PY' 2>/dev/null
import sys, os
sys.path.insert(0, os.getcwd())
import bench
from pyfocusr_amd import _hip
ctx = _hip.default_context()
r = bench.c5_1m_k10(ctx, reps=3)
print(r['ms'], r['breakdown_ms'], r['knn_index_mismatches'])
PY
done
