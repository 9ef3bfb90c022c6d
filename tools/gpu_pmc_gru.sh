#!/bin/bash
# PMC counters for the NN kernels (default lib and optional variant)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_gru
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in default $@; do
  if [ "$v" != "default" ]; then export PERCEPNET_LIB=$R/percepnet_amd/lib/variants/$v/libpercepnet_hip.so; else unset PERCEPNET_LIB; fi
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/$v -o k -- python $R/tools/kernel_times.py 65536 3 > $OUT/$v.log 2>&1
  python - <<PY
import csv, collections
d = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open("$OUT/$v/k_counter_collection.csv")):
    k = r["Kernel_Name"].split("(")[0].replace("void ","")
    if k.startswith("pn_gru") and int(r["Grid_Size"]) == 2097152:
        d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        d[k]["dur_us"].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k, v in d.items():
    print("$v", k, {c: round(sum(x)/len(x)) for c, x in v.items()})
PY
done
