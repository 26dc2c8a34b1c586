#!/bin/bash
# diagnostic builds of the fused attention (timings only): tools/ablate_attn.sh 1 2 4 8 16
cd "$(dirname "$0")/../diffusynth_amd"
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DDS_ATTN_ABL=$n -o libdiffusynth_hip_attn$n.so csrc/*.hip 2>/dev/null &
done
wait
