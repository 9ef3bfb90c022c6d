#!/bin/bash
# L1 (TCP) / texture-addresser counters of the GRU kernels in the fp16-operand mode: is operand delivery the limit?
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp
for rg in 2 3; do
  export PN_MODE=f16 PERCEPNET_X3_RG=$rg
  bash $R/tools/gpu_pmc_any.sh pn_gru_x3 0 \
    "TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" \
    "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE" \
    "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_READ_sum GRBM_GUI_ACTIVE" \
    "TA_TOTAL_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum GRBM_GUI_ACTIVE" > $O/pmc_l1_${TAG}_f16_rg$rg.txt 2>&1
  cat $O/pmc_l1_${TAG}_f16_rg$rg.txt | tr ',' '\n'
done
