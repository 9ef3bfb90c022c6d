#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_fe
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/s$i -o k -- python $R/tools/kernel_times.py 65536 3 > $OUT/s$i.log 2>&1
done
python - <<PY
import csv, collections, glob
d = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/s*/k_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ","")
        if k.startswith("pn_frontend") or k.startswith("pn_backend") or k.startswith("pn_fe_"):
            d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            d[k]["dur_us"].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k, v in d.items():
    print(k, {c: "%.4g" % (sum(x)/len(x)) for c, x in sorted(v.items())})
PY
