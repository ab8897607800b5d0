#!/bin/bash
# Build variants of the library with different operator-kernel knobs (in-tree, next to the default .so) and
# print the commands that time them on the GPU box:  bash tools/tune_spmv.sh
set -e
cd "$(dirname "$0")/../pyfocusr_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wno-unused-value -Wno-unused-result"
for v in "b128:-DPF_OP_BLOCK=128" "b512:-DPF_OP_BLOCK=512" "nt:-DPF_OP_NT=1" "b512nt:-DPF_OP_BLOCK=512 -DPF_OP_NT=1"; do
  name=${v%%:*}; defs=${v#*:}
  /opt/rocm/bin/hipcc $FLAGS $defs -o libpyfocusr_hip_$name.so *.hip
  echo "PYFOCUSR_HIP_LIB=pyfocusr_amd/csrc/libpyfocusr_hip_$name.so python tools/bench_spmv.py 250000 1000000"
done
