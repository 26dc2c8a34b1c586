#!/bin/bash
# PMC passes over one level of the depthwise microbench (GPU box): bash tools/dw_pmc.sh C H W [batch]
C=$1; H=$2; W=$3; B=${4:-128}
export PROF_SCRIPT=$(pwd)/tools/dw_microbench.py
for CNT in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TA_BUSY_avr TA_TA_BUSY_sum"; do
  bash tools/prof_pmc.sh "$CNT" --c $C --h $H --w $W --batch $B --iters 3 2>&1 | grep -E "kernel|dwconv" | cut -c1-260
done
