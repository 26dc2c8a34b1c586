#!/bin/bash
B=${1:-128}
LAYERS="96:192:256:64:1:0 192:96:256:64:0:1 384:192:128:32:0:1 768:768:64:16:1:0"
for L in $LAYERS; do
  IFS=: read cin cout h w act res <<< "$L"
  for sg in 0 6 10 14 20 30; do
    printf "stagger %4s us: " $sg
    timeout -k 10 120 python tools/conv_microbench.py --cin $cin --cout $cout --h $h --w $w --batch $B --tile 9 --act $act --res $res --iters 10 --stagger $sg 2>&1 | tail -1
  done
done
