for r in 1 2; do for sw in 1 0; do
  printf "DS_X3_ATTN_NO_PLANES=%s " $sw
  DS_X3_ATTN_NO_PLANES=$sw timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --steps 10 --warmup 2 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%.1f steps/s  %.2f ms/step' % (d['value'], d['ms_per_step']))"
done; done
