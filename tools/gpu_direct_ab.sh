#!/bin/bash
# round 6: direct-operand fp32 family (pn_nn_d.hip) against the batch family — parity tests, then ms per frame (tools/frame_time.py)
# usage (on the GPU box, through gpurun): bash tools/gpu_direct_ab.sh <tag> [sizes...]
tag=${1:-r06d}; shift
sizes=${@:-65536}
mkdir -p gpurun_out
log=gpurun_out/${tag}_direct_ab.log
: > $log
timeout 900 python -m pytest tests/test_gpu_direct.py -x -q -m gpu 2>&1 | tail -15 | tee -a $log
for B in $sizes; do
  for cfg in "0 1" "1 1" "0 2" "1 2"; do
    set -- $cfg
    echo "== direct=$1 chains=$2 B=$B" | tee -a $log
    PERCEPNET_NN_DIRECT=$1 PN_NN_CHAINS=$2 timeout 300 python tools/frame_time.py $B 30 3 2>&1 | tail -2 | tee -a $log
  done
done
