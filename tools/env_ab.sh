#!/bin/bash
# Same-box A/B of one environment switch of the product library, per conv layer and on the headline bench:  bash tools/env_ab.sh DS_HALO3_NOLUT
SW=$1; B=128
LAYERS="96:192:256:64:1:0 192:192:256:64:1:0 288:192:256:64:1:0 192:384:128:32:1:0 384:768:64:16:1:0"
for L in $LAYERS; do
  IFS=: read cin cout h w act res <<< "$L"
  for v in 1 0; do
    printf "%s=%s " "$SW" "$v"
    if [ $v = 1 ]; then export $SW=1; else unset $SW; fi
    timeout -k 10 120 python tools/conv_microbench.py --cin $cin --cout $cout --h $h --w $w --batch $B --tile 11 --act $act --res $res --iters 10 2>&1 | tail -1
  done
done
unset $SW
bash tools/bench_ab.sh "$SW=1:libdiffusynth_hip.so libdiffusynth_hip.so" 2
