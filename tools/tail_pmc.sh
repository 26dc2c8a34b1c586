#!/bin/bash
# Runs ON THE GPU BOX: issue / stall / wait split and matrix-pipe occupancy of the tail's kernels (one PMC pass of tools/tail_bench.py)
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/tailpmc
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE -d "$OUT" -o run --output-format csv -- python3 "$ROOT/tools/tail_bench.py" --iters 2 > "$OUT/bench.txt" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/run_counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'][:60]
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'GRBM_GUI_ACTIVE': cnt[k] += 1
rows = []
for k, c in acc.items():
    wc = c.get('SQ_WAVE_CYCLES', 0)
    if wc <= 0 or cnt[k] == 0: continue
    gui = c['GRBM_GUI_ACTIVE'] / 8
    rows.append((gui / cnt[k], k, c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * gui) if gui else 0, c['SQ_ACTIVE_INST_ANY'] / wc, c['SQ_WAIT_INST_ANY'] / wc, c['SQ_WAIT_ANY'] / wc, c['SQ_INSTS_VALU'] / cnt[k]))
for gui, k, mf, a, wi, w, nv in sorted(rows, reverse=True)[:14]:
    print(f"{gui/1e3:9.1f} kcyc  mfma_busy {mf:5.3f}  issuing {a:5.3f}  stalled {wi:5.3f}  waitcnt {w:5.3f}  valu/launch {nv/1e6:8.1f} M  {k}")
PY
find "$OUT" -name "*.csv" -delete
