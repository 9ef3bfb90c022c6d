#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python tools/fe_split_check.py > $O/fe_split_check.log 2>&1; echo "rc=$?" >> $O/fe_split_check.log; tail -30 $O/fe_split_check.log | cut -c1-1500
timeout 300 python tools/nan_hunt.py > $O/nan_hunt.log 2>&1; tail -25 $O/nan_hunt.log | cut -c1-400
