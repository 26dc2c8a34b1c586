#!/bin/bash
# A/B of halo tile variants on representative 3x3 layers (GPU box): bash tools/tile_ab.sh "9 4 8" [batch] [lib]
TILES="$1"; B=${2:-128}; LIB=${3:-libdiffusynth_hip.so}
LAYERS="96:192:256:64:1:0 192:192:256:64:1:0 384:192:128:32:0:1 768:768:64:16:1:0 768:384:32:8:0:1"
for L in $LAYERS; do
  IFS=: read cin cout h w act res <<< "$L"
  for t in $TILES; do
    DS_LIB=$LIB timeout -k 10 120 python tools/conv_microbench.py --cin $cin --cout $cout --h $h --w $w --batch $B --tile $t --act $act --res $res --iters 10 2>&1 | tail -1
  done
done
