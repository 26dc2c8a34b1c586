#!/bin/bash
# conv3x3_halo2 (tile 10) vs conv3x3_halo3 (tile 11) on representative 3x3 layers at the headline U-Net batch (GPU box).
B=${1:-128}
LAYERS="96:192:256:64:1:0 192:96:256:64:0:1 192:192:256:64:1:0 384:192:128:32:0:1 576:384:128:32:1:0 768:768:64:16:1:0 768:384:32:8:0:1 384:768:32:8:1:0"
for L in $LAYERS; do
  IFS=: read cin cout h w act res <<< "$L"
  for tile in 10 11; do
    printf "tile %-3s " "$tile"
    timeout -k 10 120 python tools/conv_microbench.py --cin $cin --cout $cout --h $h --w $w --batch $B --tile $tile --act $act --res $res --iters 10 2>&1 | tail -1
  done
done
