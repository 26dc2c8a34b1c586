#!/bin/bash
# PMC passes over the dwconv microbench (GPU box): bash tools/dw_pmc.sh [C H W B]
C=${1:-192}; H=${2:-256}; W=${3:-64}; B=${4:-128}
export PROF_SCRIPT=$(pwd)/tools/dw_microbench.py
for CNT in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU" \
           "SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_LDS_UNALIGNED_STALL" \
           "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
  bash tools/prof_pmc.sh "$CNT" --c $C --h $H --w $W --batch $B --iters 3 2>&1 | grep -E "kernel|dwconv" | cut -c1-240
done
