#!/bin/bash
# Tuning builds of the library with extra -D switches: tools/build_variant.sh NAME -DFOO [-DBAR ...]
# -> pyfocusr_amd/csrc/variants/libpyfocusr_hip_NAME.so (git-ignored; use with PYFOCUSR_HIP_LIB=...)
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
src=$root/pyfocusr_amd/csrc
out=$src/variants
mkdir -p $out/obj_$name
flags="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value -Wno-unused-result -Wno-unused-function"
pids=()
for f in $src/*.hip; do
  b=$(basename $f .hip)
  # only pf_persist.hip and friends see the switches; everything else is reused from the main build when present
  if grep -q "RX2_\|RX_EXP_\|PF_EXP_" $f; then
    /opt/rocm/bin/hipcc $flags "$@" -c $f -o $out/obj_$name/$b.o &
    pids+=($!)
  else
    cp $src/build/$b.o $out/obj_$name/$b.o
  fi
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o $out/libpyfocusr_hip_$name.so $out/obj_$name/*.o
rm -rf $out/obj_$name
echo built $out/libpyfocusr_hip_$name.so
