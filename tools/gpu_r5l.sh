#!/bin/bash
# round-5 call l: gates before the final evidence cycle — output hash of the pitch kernel with lane-parallel products, the probed copy
# streams (fresh and busy process), the paced capacity search inside bench.py
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out; mkdir -p $O; LOG=$O/r05l_gates.log; : > $LOG
PERCEPNET_FE=mono timeout 300 python tools/fp_variants.py 2>&1 | grep -v amdgpu.ids >> $LOG
timeout 300 python tools/fp_variants.py 2>&1 | grep -v amdgpu.ids >> $LOG
export PN_PROBE_QUICK=1
timeout 120 python tools/host_pipeline_probe.py 65536 150 2>&1 | grep -v amdgpu.ids >> $LOG
PN_PROBE_BUSY=1 timeout 120 python tools/host_pipeline_probe.py 65536 150 2>&1 | grep -v amdgpu.ids >> $LOG
timeout 600 python bench.py --steps 5 --warmup 3 --no-distinct --no-other-configs --no-cpu-baseline --no-parity --sustained-seconds 2 --realtime-seconds 4 --realtime-runs 1 --realtime-soak-seconds 10 > $O/r05l_bench.json 2>> $LOG
python - <<PY >> $LOG
import json
d = json.load(open("$O/r05l_bench.json"))
rt = d["realtime"]
print("value", d["value"], "ms_per_step", d["ms_per_step"], "realtime_streams_p99", d["realtime_streams_p99"], "next", rt["next_size"])
for b, v in rt["sizes"].items(): print(b, json.dumps(v))
for r in rt["confirmation_runs"]: print("confirm", {k: r.get(k) for k in ("streams", "seconds", "deadline_misses", "delivery_latency_ms", "met_contract", "copy_streams")})
print("copy streams", {r.get("copy_streams") for r in rt["paced_runs"]})
PY
cat $LOG | cut -c1-900
