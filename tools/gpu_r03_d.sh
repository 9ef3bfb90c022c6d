#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python tools/fe_split_check.py > $O/fe_split_check_d.log 2>&1; echo "rc=$?" >> $O/fe_split_check_d.log; tail -12 $O/fe_split_check_d.log | cut -c1-1500
PERCEPNET_LIB=$R/percepnet_amd/lib/variants/fs_out4/libpercepnet_hip.so timeout 600 python tools/fe_split_check.py --streams 65536 --frames 3 > $O/fe_split_check_d_out4.log 2>&1; tail -3 $O/fe_split_check_d_out4.log | cut -c1-1500
bash tools/gpu_pmc_fe.sh > $O/pmc_fe_r03d.txt 2>&1; grep "pn_fe_spec" $O/pmc_fe_r03d.txt | cut -c1-900
