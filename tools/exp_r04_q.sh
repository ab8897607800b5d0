#!/bin/bash
# overflow entries read in pairs (RX_OV_PAIRS) against one by one: 1M pair (k = 10), 250k pair, 15k sweep point
set -e
mkdir -p gpurun_out
tools/build_variant.sh ov0 -DRX_OV_PAIRS=0 > /dev/null
V=pyfocusr_amd/csrc/variants/libpyfocusr_hip_ov0.so
{
for rep in 1 2; do
echo "## 1M pair k=10: pairs"; SWEEP_CUTS=8 SWEEP_STRENGTHS=1.8 python tools/sweep_filter.py 1000000 10 2>&1 | tail -1
echo "## 1M pair k=10: one by one"; PYFOCUSR_HIP_LIB=$V SWEEP_CUTS=8 SWEEP_STRENGTHS=1.8 python tools/sweep_filter.py 1000000 10 2>&1 | tail -1
echo "## 250k pair: pairs"; SWEEP_CUTS=8 SWEEP_STRENGTHS=1.8 python tools/sweep_filter.py 250000 5 2>&1 | tail -1
echo "## 250k pair: one by one"; PYFOCUSR_HIP_LIB=$V SWEEP_CUTS=8 SWEEP_STRENGTHS=1.8 python tools/sweep_filter.py 250000 5 2>&1 | tail -1
done
echo "## 15k pair, new placement rule"; python tools/sweep_15k.py | tail -1
} 2>&1 | tee gpurun_out/q_ov.log
