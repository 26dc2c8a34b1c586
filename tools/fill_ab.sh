for f in ${FILLS:-256 512 1024}; do
  for dt in bf16 bf16x3; do
    printf "fill=%s %s: " $f $dt
    DS_KSPLIT_FILL=$f timeout -k 10 200 python tools/latency_bench.py --batch 1 --height 256 --steps 10 --dtype $dt 2>/dev/null | grep "eager" | sed 's/.*= //'
    printf "   B16: "
    DS_KSPLIT_FILL=$f timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --workload config2 --steps 20 --dtype $dt 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%.1f steps/s  %.3f ms/step' % (d['value'], d['ms_per_step']))"
  done
done
