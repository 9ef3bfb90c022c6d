#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out; LOG=$O/r5l_ws.log; : > $LOG
for v in ws_abl1 ws_w2; do
  PERCEPNET_SELFTEST=0 PERCEPNET_LIB=$R/percepnet_amd/lib/variants/$v/libpercepnet_hip.so timeout 300 python tools/frame_time.py 65536 20 3 2>&1 | grep -v amdgpu.ids >> $LOG
done
cut -c1-330 $LOG
