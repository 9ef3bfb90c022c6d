#!/bin/bash
# round 6: direct-operand GRU kernels against the batch family over batch sizes (default chain rule), tools/frame_time.py
tag=${1:-r06d}; shift
mkdir -p gpurun_out
log=gpurun_out/${tag}_direct_sweep.log
: > $log
for B in $@; do
  for d in 0 1; do
    echo "== direct=$d B=$B" | tee -a $log
    PERCEPNET_NN_DIRECT=$d timeout 300 python tools/frame_time.py $B 20 2 2>&1 | tail -1 | tee -a $log
  done
done
