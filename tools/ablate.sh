#!/bin/bash
# Diagnostic builds of the library with parts of the halo-conv K loop removed (results are wrong by design;
# only timings matter).  Usage: tools/ablate.sh 1 2 4 8 15 ; then DS_LIB=libdiffusynth_hip_abl<N>.so python tools/conv_microbench.py ...
cd "$(dirname "$0")/../diffusynth_amd"
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DDS_ABLATE=$n -o libdiffusynth_hip_abl$n.so csrc/*.hip &
done
wait
