#!/bin/bash
# Same-box A/B of the multi-pass switch of conv3x3_halo3 (DS_HALO3_NPASS = 1 / 2 / 4), per layer and on the headline bench.
B=128
LAYERS="96:192:256:64:1:0 192:96:256:64:0:1 192:192:256:64:1:0 288:192:256:64:1:0 192:384:128:32:1:0 384:192:128:32:0:1"
for L in $LAYERS; do
  IFS=: read cin cout h w act res <<< "$L"
  for np in 1 2 4; do
    printf "npass<=%s " "$np"
    DS_HALO3_NPASS=$np timeout -k 10 120 python tools/conv_microbench.py --cin $cin --cout $cout --h $h --w $w --batch $B --tile 11 --act $act --res $res --iters 10 2>&1 | tail -1
  done
done
bash tools/bench_ab.sh "DS_HALO3_NPASS=1:libdiffusynth_hip.so DS_HALO3_NPASS=2:libdiffusynth_hip.so DS_HALO3_NPASS=4:libdiffusynth_hip.so" 2
