#!/bin/bash
# round 4: measurements of the paired-phase GRU kernel (pn_gru_x3p_kernel): kernel times with the one-tile-per-block
# kernel (PERCEPNET_X3_RG=2) and the paired one (=3) in both shadow-operand modes, phase clocks of the variants named on the
# command line (after the tag), then the correctness subset.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=${1:-r04a}; shift
{
for m in f16 x3; do for rg in 2 3; do
  PN_MODE=$m PERCEPNET_X3_RG=$rg timeout 300 python tools/kernel_times.py 65536 10 2>&1 | grep -v amdgpu.ids | sed "s/^/rg=$rg /"
done; done
for v in "$@"; do for m in f16 x3; do
  echo "== variant $v mode $m"
  PERCEPNET_X3_RG=3 PERCEPNET_SELFTEST=0 PN_MODE=$m PERCEPNET_LIB=$R/percepnet_amd/lib/variants/$v/libpercepnet_hip.so timeout 300 python tools/x3p_clocks.py 2>&1 | grep -v "amdgpu.ids\|^{"
done; done
} > $O/x3p_$TAG.log 2>&1
cat $O/x3p_$TAG.log
timeout 1200 python -m pytest tests/test_gpu_x3.py -m gpu -x -q -k "paired or row_group or ragged or never_read" > $O/pytest_x3p_$TAG.log 2>&1; echo "pytest rc=$?" >> $O/pytest_x3p_$TAG.log
tail -15 $O/pytest_x3p_$TAG.log
