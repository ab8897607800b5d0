#!/bin/bash
# bench_cheb over every tuning build in pyfocusr_amd/csrc/variants: tools/exp_variants.sh OUTDIR [bench_cheb args]
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/$1; shift
mkdir -p $out
cd $root
: > $out/cheb.txt
for lib in "" $root/pyfocusr_amd/csrc/variants/*.so; do
  if [ -z "$lib" ]; then unset PYFOCUSR_HIP_LIB; echo "== base" >> $out/cheb.txt; else export PYFOCUSR_HIP_LIB=$lib; echo "== $(basename $lib .so | sed s/libpyfocusr_hip_//)" >> $out/cheb.txt; fi
  timeout -k 10 120 python3 tools/bench_cheb.py "$@" >> $out/cheb.txt 2>&1 || echo "FAILED rc=$?" >> $out/cheb.txt
done
