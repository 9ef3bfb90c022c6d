#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out; LOG=$O/r5h_frame_time.log; : > $LOG
for B in 1024 65536; do
  for v in default r4 default r4; do
    if [ $v != default ]; then export PERCEPNET_LIB=$R/percepnet_amd/lib/variants/$v/libpercepnet_hip.so; else unset PERCEPNET_LIB; fi
    K=200; [ $B = 65536 ] && K=20
    timeout 300 python tools/frame_time.py $B $K 5 2>&1 | grep -v amdgpu.ids >> $LOG
  done
done
for m in f16 x3; do for v in default r4; do
  if [ $v != default ]; then export PERCEPNET_LIB=$R/percepnet_amd/lib/variants/$v/libpercepnet_hip.so; else unset PERCEPNET_LIB; fi
  PN_MODE=$m timeout 300 python tools/frame_time.py 65536 20 5 2>&1 | grep -v amdgpu.ids >> $LOG
done; done
unset PERCEPNET_LIB
cat $LOG
