#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out; mkdir -p $O
bash tools/gpu_fp_variants.sh -- p_dppasm1 p_dppasm2 s_pipe s_nopf > $O/r5e_variants.log 2>&1
cp $O/fp_variants.log $O/r5e_fp_variants.log
bash tools/gpu_fe_ab.sh r5e default p_dppasm1 p_dppasm2 s_pipe s_nopf > /dev/null 2>&1
cat $O/r5e_fp_variants.log; grep -v "^   " $O/r5e_fe_ab.log; grep "spec_out" $O/r5e_fe_ab.log
