#!/bin/bash
# the small-batch secondary workloads (BASELINE configs[1]: batch 16, CFG 1; configs[0]'s shape: batch 1) with the product library
# usage: bash tools/small_batch_ab.sh [bench args, e.g. --dtype bf16]
for wl in config2 config1; do
  printf "%s: " $wl
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --workload $wl --steps 20 "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%.1f steps/s  %.3f ms/step' % (d['value'], d['ms_per_step']))"
done
