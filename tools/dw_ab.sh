#!/bin/bash
# same-box A/B of library variants on the depthwise microbench: bash tools/dw_ab.sh "libA.so libB.so"
LIBS="$1"
for L in "96 256 64" "192 256 64" "288 256 64" "192 128 32" "384 128 32" "384 64 16" "768 64 16"; do
  set -- $L
  for lib in $LIBS; do
    printf "%-30s " $lib
    DS_LIB=$lib timeout -k 10 120 python tools/dw_microbench.py --c $1 --h $2 --w $3 --batch 128 --iters 20 2>&1 | tail -1
  done
done
