#!/bin/bash
# round 6: timing ablations of the direct-operand GRU kernel (lib/variants/d_abl*: -DPN_D_ABL=<mask>, results wrong, self-test off)
tag=${1:-r06d}; shift
mkdir -p gpurun_out
log=gpurun_out/${tag}_direct_abl.log
: > $log
for v in "" $@; do
  lib=""; [ -n "$v" ] && lib=$PWD/percepnet_amd/lib/variants/$v/libpercepnet_hip.so
  echo "== variant ${v:-production}" | tee -a $log
  PERCEPNET_LIB=$lib PERCEPNET_SELFTEST=0 PERCEPNET_NN_DIRECT=1 PN_NN_CHAINS=1 timeout 300 python tools/frame_time.py 65536 20 2 2>&1 | tail -1 | tee -a $log
done
