#!/bin/bash
# the randomised sweeps with seeds other than the suite's (a longer run by hand): every line must end "0 failures"
mkdir -p gpurun_out
{
for seed in ${FUZZ_SEEDS:-211 212 213 214}; do
  python tools/fuzz_solvers.py $seed 24 2>&1 | tail -2
  python tools/fuzz_spectrum.py $seed 40 2>&1 | tail -1
  python tools/fuzz_pipeline.py $seed 12 2>&1 | tail -1
  python tests/fuzz_assembly_icp.py $seed 12 2>&1 | tail -1
  python tests/fuzz_knn.py $seed 30 2>&1 | tail -1
done
} | grep -v amdgpu.ids | tee gpurun_out/fuzz_long.log
