#!/bin/bash
# usage: gpu_fp_variants.sh <clock-variant names...> -- <variant names...>   (libraries under percepnet_amd/lib/variants/)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT=$R/gpurun_out/fp_variants.log; : > $OUT
PERCEPNET_FE=mono timeout 300 python tools/fp_variants.py 2>&1 | grep -v amdgpu.ids >> $OUT
timeout 300 python tools/fp_variants.py 2>&1 | grep -v amdgpu.ids >> $OUT
clk=1
for v in "$@"; do
  if [ "$v" == "--" ]; then clk=0; continue; fi
  L=$R/percepnet_amd/lib/variants/$v/libpercepnet_hip.so
  if [ $clk == 1 ]; then PERCEPNET_LIB=$L timeout 300 python tools/fp_clocks.py 2>&1 | grep -v amdgpu.ids >> $OUT
  else PERCEPNET_LIB=$L timeout 300 python tools/fp_variants.py 2>&1 | grep -v amdgpu.ids >> $OUT; fi
done
cat $OUT
