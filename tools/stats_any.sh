#!/bin/bash
# rocprofv3 kernel stats of bench.py with arbitrary args (GPU box): bash tools/stats_any.sh <tag> [bench args...]
TAG=$1; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o run --output-format csv -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-secondary "$@" > "$OUT/bench.json" 2> "$OUT/err.txt"
find "$OUT" -name "*kernel_trace.csv" -delete
cd "$ROOT"
python3 - "$TAG" <<'PY'
import csv, glob, sys
mytag = sys.argv[1]
sys.argv = ['x']
exec(open('tools/summarize_profiles.py').read().split("for n in (")[0])
f = glob.glob(f'gpurun_out/{mytag}/stats/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:30]:
    print(f"{short(r['Name'])[:62]:62s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:8.2f} ms {float(r['AverageNs'])/1e3:8.1f} us {float(r['TotalDurationNs'])/tot*100:5.1f}%")
print("total ms", tot/1e6)
PY
tail -c 600 "$OUT/bench.json"
