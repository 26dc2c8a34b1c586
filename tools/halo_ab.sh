#!/bin/bash
# A/B of library variants on representative 3x3 layers at the headline U-Net batch (GPU box).
#   bash tools/halo_ab.sh "<lib1> <lib2> ..." [batch] [extra microbench args]
# libs are file names under diffusynth_amd/ (libdiffusynth_hip.so = the product build)
LIBS="$1"; B=${2:-128}; shift; shift
LAYERS="96:192:256:64:1:0 192:96:256:64:0:1 384:192:128:32:0:1 768:768:64:16:1:0 768:384:32:8:0:1"
for L in $LAYERS; do
  IFS=: read cin cout h w act res <<< "$L"
  for lib in $LIBS; do
    printf "%-34s " "$lib"
    DS_LIB=$lib timeout -k 10 120 python tools/conv_microbench.py --cin $cin --cout $cout --h $h --w $w --batch $B --tile 9 --act $act --res $res --iters 10 "$@" 2>&1 | tail -1
  done
done
