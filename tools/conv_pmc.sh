#!/bin/bash
# PMC passes over one layer of the conv microbench (GPU box): bash tools/conv_pmc.sh cin cout h w act res [batch]
# Counter sets in separate passes (8 SQ slots per pass): wave-cycle split, instruction counts, MFMA-busy + clock (GRBM_GUI_ACTIVE / 8 / duration).
cin=$1; cout=$2; h=$3; w=$4; act=$5; res=$6; B=${7:-128}
export PROF_SCRIPT=$(pwd)/tools/conv_microbench.py
for CNT in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES"; do
  bash tools/prof_pmc.sh "$CNT" --cin $cin --cout $cout --h $h --w $w --batch $B --tile 11 --act $act --res $res --iters 3 2>&1 | grep -E "kernel|conv3x3" | cut -c1-260
done
