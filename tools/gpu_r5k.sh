#!/bin/bash
# round-5 call k: priorities of the two copy streams of the host pipeline (h / n / l each), back-to-back rate
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out; mkdir -p $O; LOG=$O/r05k_pipe_prio.log; : > $LOG
export PN_PROBE_QUICK=1
for p in hl nn; do PN_PIPE_PRIO=$p timeout 120 python tools/host_pipeline_probe.py 65536 150 2>&1 | grep -v amdgpu.ids >> $LOG; done
export PN_PROBE_BUSY=1
for p in hl nn ll hh nl hn; do PN_PIPE_PRIO=$p timeout 120 python tools/host_pipeline_probe.py 65536 150 2>&1 | grep -v amdgpu.ids >> $LOG; done
cat $LOG
