#!/bin/bash
# first- vs second-generation wide-tile depthwise kernel (DS_DW_V1=1 selects the old one), per level at the headline U-Net batch
for L in "96 256 64" "192 256 64" "288 256 64" "192 128 32" "384 128 32" "576 128 32"; do
  set -- $L
  for v in 1 0; do
    if [ $v = 1 ]; then export DS_DW_V1=1; else unset DS_DW_V1; fi
    printf "v1=%s " $v
    timeout -k 10 120 python tools/dw_microbench.py --c $1 --h $2 --w $3 --batch 128 --iters 20 2>&1 | tail -1
  done
done
unset DS_DW_V1
