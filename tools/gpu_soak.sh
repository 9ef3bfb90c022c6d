#!/bin/bash
# paced real-time robustness: back-to-back rate of the host pipeline, an undisturbed 20 s run and a run with an injected 50 ms
# host stall per size; then (optional "$@") front-end variants
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python - <<PY 2>&1 | grep -v amdgpu.ids
import sys, json
sys.path.insert(0, "$R")
import bench
from percepnet_amd import api, synth, weights
model = api.Model(weights.default_blob(1234))
keys = ("streams","seconds","frames","deadline_misses","delivery_latency_ms","frames_delivered_late","finished_behind_schedule_ms","host_pipeline_back_to_back_ms","recovery","met_contract","sclk_mhz_min","sclk_mhz_max")
for b in (65536, 66560, 67072, 67584):
    ctx = api.Context(model, b, device=0, nn_mode=api.NN_MFMA)
    for tag, kw in (("plain", {}), ("stall50", {"stall": (300, 50.0)}), ("stall200", {"stall": (300, 200.0)})):
        r = bench.paced_realtime(api, synth, model, 0, b, api.NN_MFMA, seconds=15.0, ctx=ctx, **kw)
        print("soak", tag, json.dumps({k: r.get(k) for k in keys}), flush=True)
    ctx.close()
PY
