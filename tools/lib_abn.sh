#!/bin/bash
# same-box A/B of several libraries on the headline bench: bash tools/lib_abn.sh libA.so libB.so ... (two rounds, interleaved)
for r in 1 2; do for lib in "$@"; do
  printf "%-34s " $lib
  DS_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --steps 10 --warmup 2 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.1f steps/s  %.2f ms/step  dominant kernel %.1f us %.4f' % (d['value'], d['ms_per_step'], r['avg_launch_us'], r['frac']))"
done; done
