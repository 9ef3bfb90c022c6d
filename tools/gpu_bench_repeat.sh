#!/bin/bash
# three back-to-back default bench runs on one box (run-to-run spread; the paced capacity search each time) + one 60 s paced soak at the
# proven batch size
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out
for i in 1 2 3; do timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-distinct > $O/bench_rep$i.json 2> $O/bench_rep$i.err; done
python - <<PY
import json
for i in (1,2,3):
    d=json.loads(open("$O/bench_rep%d.json"%i).read().strip().splitlines()[-1])
    s=d["sustained"]; r=d["realtime"]
    print("run",i,"value",d["value"],"ms",d["ms_per_step"],"sustained",s["ms_per_step"],s["value"],"p50/p99/max",s["frame_ms_p50"],s["frame_ms_p99"],s["frame_ms_max"],"sclk",s.get("sclk_mhz_min"),s.get("sclk_mhz_max"),
          "realtime_streams_p99",r["realtime_streams_p99"],"next",r["next_size"],"sizes",{k:(v["passed"],v["runs"],v["delivery_latency_ms_p99"]) for k,v in r["sizes"].items()})
PY
python - <<PY
import sys, json
sys.path.insert(0, "$R")
import bench
from percepnet_amd import api, synth, weights
model = api.Model(weights.default_blob(1234))
for b in (67584,):
    r = bench.paced_realtime(api, synth, model, 0, b, api.NN_MFMA, seconds=60.0)
    print("soak", json.dumps({k: r[k] for k in ("streams","seconds","frames","deadline_misses","delivery_latency_ms","frames_delivered_late","finished_behind_schedule_ms","met_contract","sclk_mhz_min","sclk_mhz_max")}))
PY
