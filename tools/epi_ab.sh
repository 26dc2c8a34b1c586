#!/bin/bash
for L in "96 192 256 64 1 0" "192 96 256 64 0 1" "768 768 64 16 1 0"; do set -- $L
for lib in libdiffusynth_hip.so libdiffusynth_hip_e1.so libdiffusynth_hip_e2.so libdiffusynth_hip_e3.so libdiffusynth_hip_e4.so libdiffusynth_hip_e7.so; do printf "%-28s" $lib; DS_LIB=$lib python tools/conv_microbench.py --cin $1 --cout $2 --h $3 --w $4 --batch 128 --tile 10 --act $5 --res $6 --iters 10 2>&1 | tail -1; done; done
