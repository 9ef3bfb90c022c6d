#!/bin/bash
# the four bench lines (headline, configs[1], fp16, split precision) with the committed profiles in place: `traffic` resolves
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r04s}; O=$R/gpurun_out; mkdir -p $O; cd $R
T0=$(date +%s); timeout 900 python bench.py > $O/bench_$TAG.json 2> $O/bench_$TAG.err; echo "default bench.py wall seconds: $(( $(date +%s) - T0 ))" | tee $O/bench_${TAG}_wall.txt
timeout 300 python bench.py --streams 1024 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_${TAG}_1024.json 2>> $O/bench_$TAG.err
timeout 300 python bench.py --fp16 --no-cpu-baseline > $O/bench_${TAG}_fp16.json 2>> $O/bench_$TAG.err
timeout 300 python bench.py --x3 --no-cpu-baseline > $O/bench_${TAG}_x3.json 2>> $O/bench_$TAG.err
python - <<PY
import json
for s in ("", "_1024", "_fp16", "_x3"):
    d = json.loads(open("$O/bench_$TAG" + s + ".json").read().strip().splitlines()[-1])
    print(s or "headline", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], {k: v.get("traffic_over_algorithmic") for k, v in d["dsp_roofline"].items() if isinstance(v, dict)})
PY
