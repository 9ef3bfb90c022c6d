#!/bin/bash
# usage: gpu_fe_ab.sh <out tag> <variant names ... | default>: steady-state front-end times, then PMC passes restricted to the last 8
# dispatches of pn_fe_spec_out_kernel / pn_fe_pitch_kernel at the full grid
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out; mkdir -p $O; TAG=$1; shift
LOG=$O/${TAG}_fe_ab.log; : > $LOG
for v in "$@"; do
  if [ $v != default ]; then export PERCEPNET_LIB=$R/percepnet_amd/lib/variants/$v/libpercepnet_hip.so; else unset PERCEPNET_LIB; fi
  timeout 300 python tools/fe_ab.py 2>&1 | grep -v amdgpu.ids >> $LOG
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    n=$(echo $set | tr ' ' '_')
    (cd /tmp && TMPDIR=/tmp timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/${TAG}_pmc_${v}_$n -o k -- python $R/tools/fe_ab.py > /dev/null 2>&1)
  done
  python - <<PY >> $LOG
import csv, glob, collections
d = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/${TAG}_pmc_${v}_*/k_counter_collection.csv"):
    rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith(("pn_fe_spec_out", "pn_fe_pitch", "pn_fe_spec_in", "void pn_fe_spec_in", "pn_backend")) and int(r["Grid_Size"]) >= 131072]
    by = collections.defaultdict(list)
    for r in rows: by[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for (k, c), v in by.items():
        v.sort(); last = [x for _, x in v[-8:]]
        d[k][c] = sum(last) / len(last)
for k, v in sorted(d.items()):
    print("   $v", k, {c: "%.5g" % x for c, x in sorted(v.items())}, "  (FETCH_SIZE*2+WRITE_SIZE) per stream: %.0f B" % ((v.get("FETCH_SIZE", 0) * 2 + v.get("WRITE_SIZE", 0)) * 1024 / 65536))
PY
done
unset PERCEPNET_LIB
cat $LOG
