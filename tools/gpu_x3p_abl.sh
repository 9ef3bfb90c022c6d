#!/bin/bash
# phase clocks of pn_gru_x3p_kernel and of its timing ablations (xpabl1: no gating arithmetic, xpabl2: no K-loop work)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=${1:-r04b}; shift
{
for v in "$@"; do for m in f16 x3; do
  echo "== variant $v mode $m"
  PERCEPNET_X3_RG=3 PERCEPNET_SELFTEST=0 PN_MODE=$m PERCEPNET_LIB=$R/percepnet_amd/lib/variants/$v/libpercepnet_hip.so timeout 300 python tools/x3p_clocks.py 2>&1 | grep -v "amdgpu.ids\|^{"
done; done
} > $O/x3p_abl_$TAG.log 2>&1
cat $O/x3p_abl_$TAG.log
