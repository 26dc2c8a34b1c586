#!/bin/bash
# rocprofv3 kernel stats of the configs[4] tail (GPU box): bash tools/tail_prof.sh <tag> [dtype]
TAG=${1:-tail}; DT=${2:-bf16}      # (the tag names gpurun_out/<tag>_tail: it may equal the tag of tools/profile_round.sh)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${TAG}_tail; rm -rf "$OUT"; mkdir -p "$OUT"
python tools/tail_bench.py --batch 64 --dtype $DT | tee "$OUT/tail_bench.txt"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o run --output-format csv -- python3 "$ROOT/tools/tail_bench.py" --batch 64 --dtype $DT --iters 3 > "$OUT/under_rocprof.txt" 2> "$OUT/stats.err"
find "$OUT" -name "*kernel_trace.csv" -delete
cd "$ROOT"
python3 - "$TAG" <<'PY'
import csv, glob, sys
mytag = sys.argv[1]
sys.argv = ['x']
exec(open('tools/summarize_profiles.py').read().split("for n in (")[0])
f = glob.glob(f'gpurun_out/{mytag}_tail/stats/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:22]:
    print(f"{short(r['Name'])[:58]:58s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:8.2f} ms {float(r['AverageNs'])/1e3:8.1f} us {float(r['TotalDurationNs'])/tot*100:5.1f}%")
print("total ms", tot/1e6)
PY
