#!/bin/bash
# One gpurun call = the evidence set of a snapshot:  tools/gpu_round6.sh <tag> [all|tests|bench|quick]
#   tests: `pytest -m gpu`;  bench: steady-state profile summaries per configuration (copied into profiles/ on the box so that the
#   bench lines that follow resolve `traffic` from the SAME kernels), the default bench line with its wall time, smoke(), the other
#   configurations' lines;  quick: default bench line + smoke only.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r06}; MODE=${2:-all}
O=$R/gpurun_out; mkdir -p $O
cd $R
if [ "$MODE" = "all" ] || [ "$MODE" = "tests" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q -s > $O/pytest_$TAG.log 2>&1; echo "pytest rc=$?" >> $O/pytest_$TAG.log
  tail -5 $O/pytest_$TAG.log
fi
if [ "$MODE" = "all" ] || [ "$MODE" = "bench" ]; then
  rm -rf $O/prof_summary
  timeout 1500 bash tools/gpu_profile.sh $TAG > $O/profile_$TAG.log 2>&1
  timeout 900 bash tools/gpu_profile.sh $TAG _1024 --streams 1024 >> $O/profile_$TAG.log 2>&1
  timeout 900 bash tools/gpu_profile.sh $TAG _fp16 --fp16 >> $O/profile_$TAG.log 2>&1
  timeout 900 bash tools/gpu_profile.sh $TAG _x3 --x3 >> $O/profile_$TAG.log 2>&1
  cp $O/prof_summary/${TAG}_* $R/profiles/ 2>/dev/null
fi
if [ "$MODE" != "tests" ]; then
  T0=$(date +%s); timeout 900 python bench.py --detail-out $O/bench_${TAG}_detail.json > $O/bench_$TAG.json 2> $O/bench_$TAG.err
  echo "default bench.py wall seconds: $(( $(date +%s) - T0 )); line bytes: $(tail -1 $O/bench_$TAG.json | wc -c)" | tee $O/bench_${TAG}_wall.txt; tail -c 600 $O/bench_$TAG.err
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke_$TAG.txt
  tail -1 $O/bench_$TAG.json
fi
if [ "$MODE" = "all" ] || [ "$MODE" = "bench" ]; then
  bash tools/gpu_bench_lines.sh $TAG side
  T0=$(date +%s); timeout 1200 python bench.py --full --detail-out $O/bench_${TAG}_full_detail.json > $O/bench_${TAG}_full.json 2>> $O/bench_$TAG.err
  echo "bench.py --full wall seconds: $(( $(date +%s) - T0 ))" | tee -a $O/bench_${TAG}_wall.txt
  timeout 900 python tools/tail_sweep.py 1 2 > $O/${TAG}_row_chains_sweep.log 2>&1; tail -4 $O/${TAG}_row_chains_sweep.log
  ls $O/prof_summary
fi
