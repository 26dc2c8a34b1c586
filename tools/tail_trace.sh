#!/bin/bash
# Runs ON THE GPU BOX: the ordered kernel sequence of the last tail iteration (names + durations) -> gpurun_out/tail_trace.txt
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/tailtrace
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d "$OUT" -o run --output-format csv -- python3 "$ROOT/tools/tail_bench.py" --iters 2 > "$OUT/bench.txt" 2>&1
python3 - "$OUT" > "$ROOT/gpurun_out/tail_trace.txt" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/run_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
# last iteration = from the last vq_pack_kernel on
last = max(i for i, n in enumerate(names) if 'vq_pack' in n)
t0 = int(rows[last]['Start_Timestamp'])
for r in rows[last:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  {r['Kernel_Name'][:110]}")
PY
find "$OUT" -name "*kernel_trace.csv" -delete
