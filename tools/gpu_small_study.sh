#!/bin/bash
# Crossover between the small-batch and the batch-GEMM network kernel families (and the 2-streams-per-wave front end):
# per-kernel HIP-event times at several batch sizes with each family forced (PERCEPNET_SMALL_ROWS / _GRU_ROWS / _FE_G2).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
for B in 256 1024 2048 4096 8192; do
  for fam in 1000000 0; do
    PERCEPNET_SMALL_ROWS=$fam PERCEPNET_SMALL_GRU_ROWS=$fam PERCEPNET_FE_G2=0 python tools/kernel_times.py $B 30 2>/dev/null | sed "s/^/small_rows=$fam fe_g2=0 /"
  done
  PERCEPNET_SMALL_ROWS=1000000 PERCEPNET_SMALL_GRU_ROWS=1000000 PERCEPNET_FE_G2=1 python tools/kernel_times.py $B 30 2>/dev/null | sed "s/^/small_rows=1000000 fe_g2=1 /"
done | tee $O/small_study.txt
