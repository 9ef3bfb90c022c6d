#!/bin/bash
# round 5, call 1: front-end variants (hash vs the single-launch kernel + kernel times), phase clocks, PMC of the fe kernels
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out; mkdir -p $O
bash tools/gpu_fp_variants.sh p_clk -- p_old p_dsdpp p_perm p_both s_sets2 s_sets4 > $O/r5a_variants.log 2>&1
cp $O/fp_variants.log $O/r5a_fp_variants.log
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/r5a_parity.log 2>&1; echo "rc=$?" >> $O/r5a_parity.log; tail -3 $O/r5a_parity.log
timeout 1200 bash tools/gpu_pmc_any.sh pn_fe_ 0 "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" > $O/r5a_pmc_fe.log 2>&1
cat $O/r5a_fp_variants.log; tail -8 $O/r5a_pmc_fe.log
