#!/bin/bash
# wide-tile depthwise kernel generations per level at the headline U-Net batch: DS_DW_V1=1 the first one, DS_DW_W8=1 the persistent one
# with 8 waves of 4 channels, default the persistent one with 16 waves of 2 channels
for L in "96 256 64" "192 256 64" "288 256 64" "192 128 32" "384 128 32" "576 128 32"; do
  set -- $L
  for v in DS_DW_V1 DS_DW_W8 none; do
    unset DS_DW_V1 DS_DW_W8
    [ $v != none ] && export $v=1
    printf "%-9s " $v
    timeout -k 10 120 python tools/dw_microbench.py --c $1 --h $2 --w $3 --batch 128 --iters 20 2>&1 | tail -1
  done
done
unset DS_DW_V1 DS_DW_W8
