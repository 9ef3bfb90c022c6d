#!/bin/bash
# PMC passes for the GRU kernels of a shadow-operand mode: PN_MODE=f16|x3, PERCEPNET_X3_RG=2|3
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/rocprof_counters_list.txt 2>&1
for m in f16 x3; do for rg in 2 3; do
  export PN_MODE=$m PERCEPNET_X3_RG=$rg
  bash $R/tools/gpu_pmc_any.sh pn_gru_x3 0 \
    "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
    "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
    "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA" \
    "FETCH_SIZE" "WRITE_SIZE" > $O/pmc_x3p_${TAG}_${m}_rg$rg.txt 2>&1
  cat $O/pmc_x3p_${TAG}_${m}_rg$rg.txt | tr ',' '\n' | head -80
done; done
