#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
bash tools/gpu_pmc_fe.sh > $O/pmc_fe_r03c.txt 2>&1; cat $O/pmc_fe_r03c.txt | cut -c1-900
PERCEPNET_SELFTEST=2 timeout 1200 python -m pytest tests/test_gpu_stress_weights.py -q -s > $O/pytest_stress_r03c.log 2>&1; echo "rc=$?" >> $O/pytest_stress_r03c.log
grep -E "self-test|passed|failed|rc=|Error|assert" $O/pytest_stress_r03c.log | cut -c1-300 | head -40
