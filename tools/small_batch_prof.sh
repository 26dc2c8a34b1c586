#!/bin/bash
# Runs ON THE GPU BOX: the batch-1 sampling loop (tools/latency_bench.py, 256x64, eager plan + graph replay) under rocprofv3 --kernel-trace --stats;
# prints launches per step and the kernels by total time.   gpurun -- 'bash tools/small_batch_prof.sh [dtype] [tag]'
DT=${1:-bf16}
TAG=${2:-sbprof}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT" -o run --output-format csv -- python3 "$ROOT/tools/latency_bench.py" --batch 1 --height 256 --steps 10 --dtype $DT > "$OUT/bench.txt" 2>&1
find "$OUT" -name "*kernel_trace.csv" -delete
grep "ms/step" "$OUT/bench.txt"
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/run_kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
steps = 60.0            # 2 modes x 3 repetitions x 10 steps
tot = sum(float(r['TotalDurationNs']) for r in rows) / 1e3 / steps
calls = sum(int(r['Calls']) for r in rows) / steps
print(f"kernel time per step {tot:.1f} us in {calls:.1f} launches (averaged over {steps:.0f} steps incl. pack / warm-up launches)")
for r in rows[:40]:
    n = int(r['Calls']); t = float(r['TotalDurationNs']) / 1e3
    print(f"{t/steps:8.1f} us/step  launches/step {n/steps:6.1f}  avg {float(r['AverageNs'])/1e3:7.1f}  {r['Name'][:100]}")
PY
