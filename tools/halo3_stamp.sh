#!/bin/bash
# per-wave prologue / K loop / epilogue wall times of conv3x3_halo3 (needs libdiffusynth_hip_stamp.so: tools/build_variants.py stamp=-DDS_STAMP=1)
B=${1:-128}
for L in 96:192:256:64:1:0 192:96:256:64:0:1 192:192:256:64:1:0 384:192:128:32:0:1 768:768:64:16:1:0 768:384:32:8:0:1; do
  IFS=: read cin cout h w act res <<< "$L"
  DS_LIB=libdiffusynth_hip_stamp.so timeout -k 10 120 python tools/conv_microbench.py --cin $cin --cout $cout --h $h --w $w --batch $B --tile 11 --act $act --res $res --iters 5 --stamp 1 2>&1 | tail -4
done
