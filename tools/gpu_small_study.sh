#!/bin/bash
# Crossover between the small-batch and the batch-GEMM network kernel families: per-kernel HIP-event times at several batch
# sizes with each family forced (PERCEPNET_SMALL_ROWS).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
for B in 256 1024 2048 4096 8192 16384; do
  for fam in 1000000 0; do
    PERCEPNET_SMALL_ROWS=$fam python tools/kernel_times.py $B 30 2>/dev/null | sed "s/^/small_rows=$fam /"
  done
done | tee $O/small_study.txt
