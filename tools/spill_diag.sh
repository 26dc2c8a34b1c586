#!/bin/bash
# Round-4 diagnosis of the r03 "spilling build computes garbage" event (GPU box).  Variant libraries are built on the CPU box by
#   python tools/build_variants.py "b2=-DDS_BOUNDS=1,-DDS_MINBLK=2" "b2n=-DDS_BOUNDS=1,-DDS_MINBLK=2,-mllvm,-amdgpu-opt-vgpr-liverange=false" \
#          "spill3=-DDS_MINBLK=3" "spill3w0=-DDS_MINBLK=3,-DDS_LGKM0=1"
#   b2       the failing configuration: bounds checker at two blocks per CU (256 VGPRs, 270-290 B of scratch per lane)
#   b2n      the same without the VGPR live-range optimisation of divergent regions
#   spill3   the PRODUCT kernels compiled for a 168-register budget (520-600 B of scratch): counted lgkmcnt wait kept
#   spill3w0 the same with a full lgkmcnt(0) in front of every step barrier
OUT=gpurun_out/spill_diag; mkdir -p $OUT
for v in b2 b2n; do
  echo "== $v: fused res_conv cases under the checker" | tee -a $OUT/log.txt
  DS_LIB=libdiffusynth_hip_$v.so timeout -k 10 300 python tools/debug_bounds_kernel.py 2>&1 | tail -30 | tee -a $OUT/log.txt
done
for v in spill3 spill3w0; do
  echo "== $v: halo kernel tests" | tee -a $OUT/log.txt
  DS_LIB=libdiffusynth_hip_$v.so timeout -k 10 600 python -m pytest tests/test_hip_kernels.py -m gpu -q -k "halo or quad" 2>&1 | tail -15 | tee -a $OUT/log.txt
done
