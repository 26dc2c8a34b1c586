#!/bin/bash
# split-precision tier check (GPU box): the new kernel's tests, the tier's parity tests, then the tier's throughput with / without it
timeout -k 10 600 python -m pytest tests/test_hip_kernels.py tests/test_hip_unet.py -m gpu -x -q -k "x3 or split" 2>&1 | tail -8
for v in 1 0; do
  printf "DS_NO_X3=%s " $v
  DS_NO_X3=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --dtype bf16x3 --steps 5 --warmup 1 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%.1f steps/s  %.2f ms/step' % (d['value'], d['ms_per_step']))"
done
