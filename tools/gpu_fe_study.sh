#!/bin/bash
# Where the front end's LDS bank conflicts and cycles come from: PMC counters of the cumulative phase-ablation builds
# (-DPN_FE_ABL=1..4, then the full kernel) and the shader-clock phase profile (-DPN_FE_CLOCKS).
#   usage: gpu_fe_study.sh <variant-prefix>   (variants <prefix>abl1..4, <prefix>clk built by tools/build_fe_variants.py)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/fe_study; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
P=${1:-fe_}
for v in ${P}abl1 ${P}abl2 ${P}abl3 ${P}abl4 default; do
  if [ "$v" != "default" ]; then export PERCEPNET_LIB=$R/percepnet_amd/lib/variants/$v/libpercepnet_hip.so; else unset PERCEPNET_LIB; fi
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS \
     --output-format csv -d $O/$v -o k -- python $R/tools/kernel_times.py 65536 3 > $O/$v.log 2>&1
  python - <<PY
import csv, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open("$O/$v/k_counter_collection.csv")):
    if "pn_frontend" in r["Kernel_Name"]:
        d[r["Counter_Name"]].append(float(r["Counter_Value"])); d["dur_us"].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
print("$v", {c: "%.4g" % (sum(x)/len(x)) for c, x in sorted(d.items())})
PY
done | tee $O/summary.txt
unset PERCEPNET_LIB
if [ -f $R/percepnet_amd/lib/variants/${P}clk/libpercepnet_hip.so ]; then
  PERCEPNET_LIB=$R/percepnet_amd/lib/variants/${P}clk/libpercepnet_hip.so python $R/tools/fe_clocks.py 65536 | tee $O/phase_cycles.txt
fi
