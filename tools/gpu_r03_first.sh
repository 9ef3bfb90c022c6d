#!/bin/bash
# round 3, first GPU call: issue probe, full GPU suite (incl. the new RCCL / stress / CLI tests), bench line.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 120 tools/probes/valu_issue_probe > $O/valu_issue_probe.log 2>&1; tail -60 $O/valu_issue_probe.log
timeout 2400 python -m pytest tests -m gpu -x -q -s > $O/pytest_r03a.log 2>&1; echo "pytest rc=$?" >> $O/pytest_r03a.log
tail -8 $O/pytest_r03a.log
timeout 900 python bench.py > $O/bench_r03a.json 2> $O/bench_r03a.err; tail -c 400 $O/bench_r03a.err; head -c 3000 $O/bench_r03a.json
