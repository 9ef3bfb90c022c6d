#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out; mkdir -p $O
bash tools/gpu_fp_variants.sh -- be_pipe s_w4 > $O/r5f_variants.log 2>&1
cp $O/fp_variants.log $O/r5f_fp_variants.log
LOG=$O/r5f_fe_ab.log; : > $LOG
for v in default be_pipe s_w4; do
  if [ $v != default ]; then export PERCEPNET_LIB=$R/percepnet_amd/lib/variants/$v/libpercepnet_hip.so; else unset PERCEPNET_LIB; fi
  timeout 300 python tools/fe_ab.py 2>&1 | grep -v amdgpu.ids >> $LOG
done
unset PERCEPNET_LIB
cat $O/r5f_fp_variants.log $LOG
