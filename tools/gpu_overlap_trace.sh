#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
V=$R/percepnet_amd/lib/variants
rm -rf /tmp/ovt; PERCEPNET_LIB=$V/${OVL:-oneblk}/libpercepnet_hip.so PN_SKEW_CYCLES=9000000 rocprofv3 --kernel-trace --output-format csv -d /tmp/ovt -o t -- python $R/tools/two_ctx.py 32768 2 6 > $O/overlap_trace_run.log 2>&1
F=$(find /tmp/ovt -name "*kernel_trace.csv" | head -1)
python $R/tools/overlap_trace.py $F > $O/overlap_trace_${OVL:-oneblk}.txt 2>&1; head -70 $O/overlap_trace_${OVL:-oneblk}.txt
