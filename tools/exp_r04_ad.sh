#!/bin/bash
set -e
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu -k "knn" > gpurun_out/ad_tests.log 2>&1 || { tail -30 gpurun_out/ad_tests.log; exit 1; }
tail -2 gpurun_out/ad_tests.log
python bench.py --steps 3 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); v=d['c5_1m_k10']; print('c5', v['ms'], v['breakdown_ms'], v.get('knn_index_mismatches'))"
