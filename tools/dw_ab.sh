#!/bin/bash
LIBS=${1:-"libdiffusynth_hip.so"}; B=${2:-128}
for S in "96 256 64" "192 256 64" "288 256 64" "384 128 32" "768 64 16" "768 32 8"; do set -- $S
for lib in $LIBS; do printf "%-28s" $lib; DS_LIB=$lib python tools/dw_microbench.py --c $1 --h $2 --w $3 --batch $B --iters 10 2>&1 | tail -1; done; done
