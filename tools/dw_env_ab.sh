#!/bin/bash
# depthwise 7x7 per level at the headline U-Net batch: the MFMA kernel against the LDS-tile stencil kernel (DS_DW_NO_MFMA=1)
for L in "96 256 64" "192 256 64" "288 256 64" "192 128 32" "384 128 32" "576 128 32" "384 64 16" "768 64 16" "1152 64 16" "768 32 8" "1536 32 8"; do
  set -- $L
  for v in ${DW_AB:-0}; do
    if [ $v = 1 ]; then export DS_DW_NO_MFMA=1; else unset DS_DW_NO_MFMA; fi
    printf "no_mfma=%s " $v
    timeout -k 10 120 python tools/dw_microbench.py --c $1 --h $2 --w $3 --batch 128 --iters 20 2>&1 | tail -1
  done
done
unset DS_DW_NO_MFMA
