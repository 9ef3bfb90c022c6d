#!/bin/bash
# round 5, call 2: new tests (active set, refused launch, bench record), comb-filter order A/B with L2 counters, FETCH_SIZE calibration
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_lifecycle.py -x -q -m gpu > $O/r5b_lifecycle.log 2>&1; echo "rc=$?" >> $O/r5b_lifecycle.log; tail -4 $O/r5b_lifecycle.log
timeout 900 python -m pytest tests/test_gpu_bench.py -x -q -m gpu > $O/r5b_benchtests.log 2>&1; echo "rc=$?" >> $O/r5b_benchtests.log; tail -4 $O/r5b_benchtests.log
# comb order A/B: same inputs (kernel_times.py), FETCH_SIZE / WRITE_SIZE / L2 hit counters of pn_fe_spec_out_kernel
for v in default s_quarter; do
  if [ $v != default ]; then export PERCEPNET_LIB=$R/percepnet_amd/lib/variants/$v/libpercepnet_hip.so; else unset PERCEPNET_LIB; fi
  timeout 600 bash tools/gpu_pmc_any.sh pn_fe_spec_out 196608 "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum" > $O/r5b_comb_$v.log 2>&1
  tail -2 $O/r5b_comb_$v.log
done
unset PERCEPNET_LIB
# calibration probe
cd /tmp && export TMPDIR=/tmp
for set in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum"; do
  n=$(echo $set | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/r5b_probe_$n -o k -- $R/tools/probes/fetch_unaligned_probe 65536 418 > $O/r5b_probe_$n.log 2>&1
done
python - <<PY
import csv, glob, collections
d = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/r5b_probe_*/k_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(d.items()):
    print(k, {c: "%.5g" % (sum(x)/len(x)) for c, x in sorted(v.items())})
PY
