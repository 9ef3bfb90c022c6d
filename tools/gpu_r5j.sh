#!/bin/bash
# round-5 call j: the host pipeline back to back (probe + env variants + a kernel/copy trace of mode B)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out; mkdir -p $O; LOG=$O/r05j_host_pipeline.log; : > $LOG
timeout 200 python tools/host_pipeline_probe.py 65536 200 2>&1 | grep -v amdgpu.ids >> $LOG
HSA_ENABLE_SDMA=0 timeout 200 python tools/host_pipeline_probe.py 65536 200 2>&1 | grep -v amdgpu.ids >> $LOG
GPU_MAX_HW_QUEUES=8 timeout 200 python tools/host_pipeline_probe.py 65536 200 2>&1 | grep -v amdgpu.ids >> $LOG
(cd /tmp && export TMPDIR=/tmp && PN_PROBE_ONLY=B timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/r05j_trace -o t -- python $R/tools/host_pipeline_probe.py 65536 60 > /dev/null 2>&1)
python - <<PY >> $LOG 2>&1
import csv, glob, collections
for f in glob.glob("$O/r05j_trace/**/t_memory_copy_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    by = collections.defaultdict(list)
    for r in rows:
        by[r.get("Direction") or r.get("Kind")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    for k, v in by.items():
        big = [x for x in v if x > 0.3]
        print("copies", k, "n", len(v), "large n", len(big), "avg ms of large", sum(big) / max(len(big), 1), "max", max(v))
for f in glob.glob("$O/r05j_trace/**/t_kernel_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    by = collections.defaultdict(list)
    for r in rows:
        by[(r["Kernel_Name"].split("(")[0][:40], r["Grid_Size"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    tot = 0
    for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        last = v[len(v) // 2:]
        print("kernel", k, "n", len(v), "avg ms (2nd half)", round(sum(last) / len(last), 4))
    # frame period from the backend kernel's start times
    st = sorted(int(r["Start_Timestamp"]) for r in rows if r["Kernel_Name"].startswith("pn_backend"))
    d = [(b - a) / 1e6 for a, b in zip(st, st[1:])]
    print("backend start-to-start ms, last 40:", [round(x, 3) for x in d[-40:]])
PY
rm -rf $O/r05j_trace
cat $LOG | cut -c1-1500
