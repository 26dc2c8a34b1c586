#!/bin/bash
# same-box A/B of the line-sized (LDS-staged) epilogues against the register-only ones: bash tools/rows_ab.sh "libA.so libB.so"
LIBS="$1"
for L in 192:96:256:64:0:1 192:96:256:64:0:0 384:192:128:32:0:1 288:96:256:64:0:0; do
  IFS=: read cin cout h w act res <<< "$L"
  for lib in $LIBS $LIBS; do
    printf "%-30s " $lib
    DS_LIB=$lib timeout -k 10 120 python tools/conv_microbench.py --cin $cin --cout $cout --h $h --w $w --batch 128 --tile 11 --act $act --res $res --iters 10 2>&1 | tail -1
  done
done
