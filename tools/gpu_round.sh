#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): the -m gpu suite, the default bench, and the bench under rocprofv3 --stats.
# usage: bash tools/gpu_round.sh <tag> [pytest args]
set -o pipefail
TAG=${1:-run}; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 900 python -m pytest tests -m gpu -x -q "$@" > "$OUT/pytest.txt" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/pytest.txt"
tail -5 "$OUT/pytest.txt"
grep -q "pytest rc=0" "$OUT/pytest.txt" || exit 1
timeout -k 10 400 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err" || { tail -20 "$OUT/bench_default.err"; exit 1; }
tail -1 "$OUT/bench_default.json" | cut -c1-1500
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o run --output-format csv -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-secondary --steps 10 > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err"
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
for r in rows[:24]:
    print(f"{short(r['Name'])[:58]:58s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:8.2f} ms {float(r['AverageNs'])/1e3:8.1f} us {float(r['TotalDurationNs'])/tot*100:5.1f}%")
print("total ms", tot/1e6)
PY
