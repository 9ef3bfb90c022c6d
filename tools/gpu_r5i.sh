#!/bin/bash
# round-5 call i: real-time robustness sweep + pitch-kernel experiment variants (steady-state times and output hash)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out; mkdir -p $O
for v in default x0 x_pre x_rdpk x_both; do
  if [ $v != default ]; then export PERCEPNET_LIB=$R/percepnet_amd/lib/variants/$v/libpercepnet_hip.so; else unset PERCEPNET_LIB; fi
  echo "== $v" >> $O/r05i_fe_ab.log
  timeout 300 python tools/fe_ab.py 2>&1 | grep -v amdgpu.ids >> $O/r05i_fe_ab.log
done
unset PERCEPNET_LIB
cat $O/r05i_fe_ab.log
bash tools/gpu_soak.sh > $O/r05i_soak.log 2>&1; cat $O/r05i_soak.log | cut -c1-600
