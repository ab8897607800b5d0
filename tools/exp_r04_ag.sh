#!/bin/bash
set -e
for h in 70 58 50 42 34 26; do
echo "## 1M pair k=10 PF_PERSIST_HOLD=$h"
PF_PERSIST_HOLD=$h SWEEP_CUTS=8 SWEEP_STRENGTHS=1.8 python tools/sweep_filter.py 1000000 10 2>&1 | tail -1
done
for h in 74 66 58 50; do
echo "## 400k pair k=5 PF_PERSIST_HOLD=$h"
PF_PERSIST_HOLD=$h SWEEP_CUTS=8 SWEEP_STRENGTHS=1.8 python tools/sweep_filter.py 400000 5 2>&1 | tail -1
done
