#!/bin/bash
# Runs ON THE GPU BOX: BASELINE configs[1] (batch 16, CFG 1) under rocprofv3 --kernel-trace --stats: kernels by time per step
DT=${1:-bf16x3}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/b16prof
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT" -o run --output-format csv -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-secondary --workload config2 --steps 20 --warmup 3 --dtype $DT > "$OUT/bench.json" 2> "$OUT/err.txt"
find "$OUT" -name "*kernel_trace.csv" -delete
python3 - "$OUT" <<'PY'
import csv, glob, sys, json
d = json.loads(open(sys.argv[1] + '/bench.json').read().strip().splitlines()[-1])
print(d['value'], 'steps/s', d['ms_per_step'], 'ms/step')
f = glob.glob(sys.argv[1] + '/**/run_kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
steps = 23.0
tot = sum(float(r['TotalDurationNs']) for r in rows) / 1e3 / steps
print(f"kernel time per step {tot:.0f} us")
for r in rows[:26]:
    n = int(r['Calls']); t = float(r['TotalDurationNs']) / 1e3
    print(f"{t/steps:8.1f} us/step  launches/step {n/steps:6.1f}  avg {float(r['AverageNs'])/1e3:7.1f}  {r['Name'][:100]}")
PY
