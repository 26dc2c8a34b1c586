#!/bin/bash
# Same-box A/B of library variants on the headline bench (devices of the pool differ by several %: never compare across gpurun calls).
#   bash tools/bench_ab.sh "<lib or ENV=..:lib> ..." [rounds]      e.g.  "libdiffusynth_hip.so libdiffusynth_hip_noxcd.so DS_NO_HALO3=1:libdiffusynth_hip.so"
VARS="$1"; R=${2:-2}
for r in $(seq 1 $R); do
  for v in $VARS; do
    lib=${v##*:}; envs=""
    [ "$v" != "$lib" ] && envs=${v%:*}
    printf "%-52s " "$v"
    env $envs DS_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%.1f steps/s  %.3f ms/step  3x3 %.1f TF (%.1f us avg)' % (d['value'], d['ms_per_step'], r['achieved'], r['avg_launch_us']))"
  done
done
