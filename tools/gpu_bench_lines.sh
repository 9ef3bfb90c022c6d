#!/bin/bash
# the bench lines with the committed profiles in place (`traffic` resolves):  tools/gpu_bench_lines.sh <tag> [all|side]
#   side: configs[1], fp16, split precision only (the default line was already taken)
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r06}; WHAT=${2:-all}; O=$R/gpurun_out; mkdir -p $O; cd $R
if [ "$WHAT" = "all" ]; then
  T0=$(date +%s); timeout 900 python bench.py --detail-out $O/bench_${TAG}_detail.json > $O/bench_$TAG.json 2> $O/bench_$TAG.err
  echo "default bench.py wall seconds: $(( $(date +%s) - T0 )); line bytes: $(tail -1 $O/bench_$TAG.json | wc -c)" | tee $O/bench_${TAG}_wall.txt
fi
timeout 300 python bench.py --streams 1024 --steps 200 --warmup 20 --no-cpu-baseline --detail-out $O/bench_${TAG}_1024_detail.json > $O/bench_${TAG}_1024.json 2>> $O/bench_$TAG.err
timeout 300 python bench.py --fp16 --no-cpu-baseline --detail-out $O/bench_${TAG}_fp16_detail.json > $O/bench_${TAG}_fp16.json 2>> $O/bench_$TAG.err
timeout 300 python bench.py --x3 --no-cpu-baseline --detail-out $O/bench_${TAG}_x3_detail.json > $O/bench_${TAG}_x3.json 2>> $O/bench_$TAG.err
python - <<PY
import json
for s in ("", "_1024", "_fp16", "_x3"):
    try:
        d = json.loads(open("$O/bench_$TAG" + s + ".json").read().strip().splitlines()[-1])
        print(s or "headline", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d.get("dsp_roofline"))
    except Exception as e:
        print(s, "unreadable:", e)
PY
