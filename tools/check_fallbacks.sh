#!/bin/bash
# the solver-related part of the GPU suite with each fast path switched off in turn (the paths a device shared with another
# tenant, or an untrusted resident kernel, falls back to): every run must pass
mkdir -p gpurun_out
sel="eigs or spectr or pair or orth or end_to_end or focusr or resident or timeout or partial or messy or open_mesh"
for v in "PF_PERSIST=0" "PF_ORTH_LOCAL=0" "PF_EIGS_PRO=0" "PF_PAIR_DRIVER=python" "PF_PAIR_BUILD_STREAMS=1"; do
  echo "## $v"
  env $v python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$sel" 2>&1 | tail -4
done | tee gpurun_out/fallbacks.log
