#!/bin/bash
# round 5: one gpurun call = the evidence set of a snapshot: `pytest -m gpu`, the steady-state profile summaries of every configuration
# (copied into profiles/ on the box so that the bench lines that follow resolve `traffic` from the SAME kernels), the four bench lines, smoke()
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r05}; MODE=${2:-all}
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
  T0=$(date +%s); timeout 900 python bench.py > $O/bench_$TAG.json 2> $O/bench_$TAG.err; echo "default bench.py wall seconds: $(( $(date +%s) - T0 ))" | tee $O/bench_${TAG}_wall.txt; tail -c 400 $O/bench_$TAG.err
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke_$TAG.txt
  timeout 300 python bench.py --streams 1024 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_${TAG}_1024.json 2>> $O/bench_$TAG.err
  timeout 300 python bench.py --fp16 --no-cpu-baseline > $O/bench_${TAG}_fp16.json 2>> $O/bench_$TAG.err
  timeout 300 python bench.py --x3 --no-cpu-baseline > $O/bench_${TAG}_x3.json 2>> $O/bench_$TAG.err
  head -c 600 $O/bench_$TAG.json; echo
  ls $O/prof_summary
fi
