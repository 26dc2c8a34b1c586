#!/bin/bash
# Runs ON THE GPU BOX: tools/tail_bench.py under rocprofv3 --kernel-trace --stats; prints the per-iteration kernel table.
#   gpurun -- 'bash tools/tail_prof.sh [tag]'   -> gpurun_out/<tag>/ (default tailprof)
TAG=${1:-tailprof}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT" -o run --output-format csv -- python3 "$ROOT/tools/tail_bench.py" > "$OUT/bench.txt" 2>&1
find "$OUT" -name "*kernel_trace.csv" -delete
tail -1 "$OUT/bench.txt"
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/run_kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = 0.0
for r in rows:
    tot += float(r['TotalDurationNs']) / 1e3 / 6
for r in rows[:26]:
    n = int(r['Calls']); t = float(r['TotalDurationNs']) / 1e3
    print(f"{t/6:8.1f} us/iter  calls {n:4d}  avg {float(r['AverageNs'])/1e3:8.1f}  {r['Name'][:96]}")
print(f"kernel time per iteration (6 iterations incl. the untimed first): {tot:.1f} us")
PY
