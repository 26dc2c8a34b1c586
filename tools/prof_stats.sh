#!/bin/bash
# kernel-time breakdown of the default bench (GPU box): bash tools/prof_stats.sh [bench args]
ROOT=$(pwd); OUT=$ROOT/gpurun_out/stats_tmp; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$OUT" -o run --output-format csv -- python3 "$ROOT/bench.py" --no-cpu-baseline "$@" > "$OUT/bench.json" 2> "$OUT/err.txt"
cd "$ROOT"; rm -f "$OUT"/*kernel_trace.csv "$OUT"/*/*kernel_trace.csv
python3 - <<'PY'
import csv, glob, sys
sys.argv=['x']
exec(open('tools/summarize_profiles.py').read().split("for n in (")[0])
f = glob.glob('gpurun_out/stats_tmp/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:26]:
    print(f"{short(r['Name'])[:58]:58s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:8.2f} ms {float(r['AverageNs'])/1e3:8.1f} us {float(r['TotalDurationNs'])/tot*100:5.1f}%")
print("total ms", tot/1e6)
PY
tail -1 gpurun_out/stats_tmp/bench.json | cut -c1-200
