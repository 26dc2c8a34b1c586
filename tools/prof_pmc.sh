#!/bin/bash
# per-kernel PMC averages of the default bench (GPU box): bash tools/prof_pmc.sh "SQ_WAVE_CYCLES SQ_WAIT_ANY ..." [bench args]
# PROF_SCRIPT=/abs/path/to/tools/dw_microbench.py profiles another script instead (its args follow)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_tmp; rm -rf "$OUT"; mkdir -p "$OUT"
CNT="$1"; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --pmc $CNT -d "$OUT" -o run --output-format csv -- python3 ${PROF_SCRIPT:-"$ROOT/bench.py" --no-cpu-baseline --steps 3 --warmup 1} "$@" > "$OUT/bench.json" 2> "$OUT/err.txt"
cd "$ROOT"
python3 - <<'PY'
import csv, glob, sys, collections
sys.argv=['x']
exec(open('tools/summarize_profiles.py').read().split("for n in (")[0])
f = glob.glob('gpurun_out/pmc_tmp/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set); dur = collections.defaultdict(float)
for r in csv.DictReader(open(f)):
    k = short(r['Kernel_Name'])
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Dispatch_Id'] not in cnt[k]:
        cnt[k].add(r['Dispatch_Id']); dur[k] += float(r['End_Timestamp']) - float(r['Start_Timestamp'])
names = sorted({c for v in acc.values() for c in v})
print("kernel".ljust(46), "n".rjust(5), "us".rjust(8), *[n.replace("SQ_", "")[:14].rjust(15) for n in names])
for k in sorted(acc, key=lambda k: -dur[k])[:18]:
    n = len(cnt[k])
    print(k[:46].ljust(46), str(n).rjust(5), f"{dur[k]/n/1e3:8.1f}", *[f"{acc[k][c]/n:15.3g}" for c in names])
PY
rm -rf "$OUT"/*/ "$OUT"/*.csv
