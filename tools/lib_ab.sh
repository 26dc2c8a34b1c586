#!/bin/bash
# Same-box A/B of library variants, per layer (conv microbench) and on the headline bench.
#   bash tools/lib_ab.sh "libA.so libB.so" [bench rounds]
LIBS="$1"; R=${2:-1}; B=128
LAYERS="96:192:256:64:1:0 192:96:256:64:0:1 192:192:256:64:1:0 384:192:128:32:0:1 192:384:128:32:1:0 768:768:64:16:1:0 768:384:32:8:0:1"
for L in $LAYERS; do
  IFS=: read cin cout h w act res <<< "$L"
  for lib in $LIBS; do
    printf "%-34s " "$lib"
    DS_LIB=$lib timeout -k 10 120 python tools/conv_microbench.py --cin $cin --cout $cout --h $h --w $w --batch $B --tile 11 --act $act --res $res --iters 10 2>&1 | tail -1
  done
done
bash tools/bench_ab.sh "$LIBS" $R
