#!/bin/bash
# diagnostic builds of the depthwise kernel (timings only): tools/ablate_dw.sh 1 2 4 7
cd "$(dirname "$0")/../diffusynth_amd"
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DDS_DW_ABL=$n -o libdiffusynth_hip_dw$n.so csrc/*.hip 2>/dev/null &
done
wait
