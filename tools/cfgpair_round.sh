set -o pipefail
mkdir -p gpurun_out/cfgpair
timeout -k 10 600 python -m pytest tests/test_hip_unet.py tests/test_hip_fullsize.py -m gpu -x -q -k "paired or headline or trajector" 2>&1 | tail -5 | tee gpurun_out/cfgpair/tests.txt
for dt in bf16x3 bf16; do for v in 0 1; do
  printf "%s DS_NO_CFG_PAIR=%s " $dt $v
  DS_NO_CFG_PAIR=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --dtype $dt --steps 8 --warmup 2 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%.1f steps/s  %.2f ms/step' % (d['value'], d['ms_per_step']))"
done; done | tee gpurun_out/cfgpair/ab.txt
