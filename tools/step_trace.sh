#!/bin/bash
# Runs ON THE GPU BOX: ordered kernel sequence of the last sampling step at batch 1 (names + durations) -> gpurun_out/step_trace.txt
DT=${1:-bf16}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/steptrace
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d "$OUT" -o run --output-format csv -- python3 "$ROOT/tools/latency_bench.py" --batch 1 --height 256 --steps 4 --dtype $DT > "$OUT/bench.txt" 2>&1
python3 - "$OUT" > "$ROOT/gpurun_out/step_trace.txt" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/run_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
idx = [i for i, n in enumerate(names) if 'ddim_step' in n]
# eager-mode steps come first (3 repetitions x 4 steps): take the 8th step's span
a, b = idx[6] + 1, idx[7] + 1
t0 = int(rows[a]['Start_Timestamp'])
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  {r['Kernel_Name'][:100]}")
PY
find "$OUT" -name "*kernel_trace.csv" -delete
