#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out; mkdir -p $O
bash tools/gpu_fp_variants.sh -- s_sets7 s_sets5 s_quarter > $O/r5c_variants.log 2>&1
cp $O/fp_variants.log $O/r5c_fp_variants.log
for v in s_sets7; do
  export PERCEPNET_LIB=$R/percepnet_amd/lib/variants/$v/libpercepnet_hip.so
  timeout 600 bash tools/gpu_pmc_any.sh pn_fe_spec_out 196608 "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum" > $O/r5c_comb_$v.log 2>&1
  tail -2 $O/r5c_comb_$v.log
done
cat $O/r5c_fp_variants.log
