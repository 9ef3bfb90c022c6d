#!/bin/bash
# usage: gpu_pmc_any.sh <kernel-prefix> <grid or 0> "<counter list>" ["<counter list 2>" ...]
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_any
rm -rf $OUT; mkdir -p $OUT
KP=$1; GRID=$2; shift 2
cd /tmp && export TMPDIR=/tmp
i=0
for set in "$@"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/s$i -o k -- python $R/tools/kernel_times.py 65536 3 > $OUT/s$i.log 2>&1
done
python - <<PY
import csv, collections, glob
d = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/s*/k_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ","")
        if k.startswith("$KP") and ($GRID == 0 or int(r["Grid_Size"]) == $GRID):
            d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            d[k]["dur_us"].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k, v in d.items():
    print(k, {c: "%.4g" % (sum(x)/len(x)) for c, x in sorted(v.items())})
PY
