#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
V=$R/percepnet_amd/lib/variants
timeout 600 python tools/fe_split_check.py --streams 8195,65536 > $O/fe_split_check_g.log 2>&1; echo "rc=$?" >> $O/fe_split_check_g.log; tail -4 $O/fe_split_check_g.log | cut -c1-1200
for v in fp_rows; do PERCEPNET_LIB=$V/$v/libpercepnet_hip.so timeout 600 python tools/fe_split_check.py --streams 65536 --frames 3 > $O/fe_split_check_g_$v.log 2>&1; echo "== $v"; tail -2 $O/fe_split_check_g_$v.log | cut -c600-1200; done
bash tools/gpu_pmc_fe.sh > $O/pmc_fe_r03g.txt 2>&1; grep "pn_fe_pitch" $O/pmc_fe_r03g.txt | cut -c1-900
