#!/bin/bash
# round 6: paced real-time runs at ONE size, many undisturbed runs, under different settings (chains, hardware queues):
#   tools/gpu_paced_ab.sh <tag> <streams> <runs> "<ENV=.. ENV=..>" "<ENV=..>" ...
tag=$1; B=$2; R=$3; shift 3
mkdir -p gpurun_out; log=gpurun_out/${tag}_paced_ab.log; : > $log
for cfg in "$@"; do
  echo "== $cfg  B=$B runs=$R" | tee -a $log
  env $cfg timeout 600 python tools/realtime_capacity.py --seconds 5 --runs $R --soak-seconds 0 --grid $B:$B:512 --out gpurun_out/${tag}_paced_tmp.json 2>&1 | grep -v amdgpu.ids | cut -c1-110,230-420 | tee -a $log
done
