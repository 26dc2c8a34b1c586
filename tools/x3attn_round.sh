set -o pipefail
mkdir -p gpurun_out/x3attn
timeout -k 10 600 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "attn_x3" 2>&1 | tail -5 | tee gpurun_out/x3attn/kern.txt
timeout -k 10 600 python -m pytest tests/test_hip_unet.py tests/test_hip_fullsize.py -m gpu -x -q -k "x3 or variants or bf16x3" 2>&1 | tail -5 | tee gpurun_out/x3attn/unet.txt
for v in 0 1; do
  printf "DS_X3_ATTN_APPLY_PASS=%s " $v
  DS_X3_ATTN_APPLY_PASS=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --steps 5 --warmup 1 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%.1f steps/s  %.2f ms/step' % (d['value'], d['ms_per_step']))"
done | tee gpurun_out/x3attn/ab.txt
bash tools/stats_any.sh x3attn_stats --steps 5 --warmup 1 2>&1 | grep -v "^at::\|pack_conv\|rocclr" | head -28
