for r in 1 2; do for lib in libdiffusynth_hip_ng0.so libdiffusynth_hip.so; do
  printf "%-30s bf16: " $lib
  DS_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --steps 10 --warmup 2 --dtype bf16 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.1f steps/s  %.2f ms/step  dominant kernel %.1f us' % (d['value'], d['ms_per_step'], r['avg_launch_us']))"
done; done
