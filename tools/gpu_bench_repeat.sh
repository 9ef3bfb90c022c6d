#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out
for i in 1 2 3; do timeout 600 python bench.py --no-cpu-baseline --no-other-configs > $O/bench_rep$i.json 2> $O/bench_rep$i.err; done
python - <<PY
import json
for i in (1,2,3):
    d=json.loads(open("$O/bench_rep%d.json"%i).read().strip().splitlines()[-1])
    s=d["sustained"]; r=d["realtime"]
    print("run",i,"value",d["value"],"ms",d["ms_per_step"],"sustained",s["ms_per_step"],s["value"],"p50/p99/max",s["frame_ms_p50"],s["frame_ms_p99"],s["frame_ms_max"],"sclk",s.get("sclk_mhz_min"),s.get("sclk_mhz_max"),
          "paced",[(x["streams"],x["deadline_misses"],x["submit_call_ms"]["p99"],x["finished_behind_schedule_ms"]) for x in r["paced_runs"]],"rt",r["realtime_streams_p99"])
PY
