#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): default bench, the same command under rocprofv3 --kernel-trace --stats, and two
# separate PMC passes (FETCH_SIZE, WRITE_SIZE) as MI355X_MICROARCH.md prescribes.  Output under gpurun_out/<tag>/;
# tools/summarize_profiles.py turns it into the committed profiles/<tag>_* files.
set -e
TAG=${1:-r04}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
rm -rf "$OUT" && mkdir -p "$OUT"
timeout -k 10 400 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o run --output-format csv -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-secondary --steps 10 > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o run --output-format csv -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-secondary --steps 4 --warmup 3 > "$OUT/pmc_fetch.json" 2> "$OUT/pmc_fetch.err"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmc_write" -o run --output-format csv -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-secondary --steps 4 --warmup 3 > "$OUT/pmc_write.json" 2> "$OUT/pmc_write.err"
# matrix-pipe occupancy and clock (its own pass: 8 SQ counters + GRBM): MFMA-busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8),
# clock = GRBM_GUI_ACTIVE / 8 / duration (MI355X_MICROARCH.md, DVFS give-back)
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE -d "$OUT/pmc_mfma" -o run --output-format csv -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-secondary --steps 4 --warmup 3 > "$OUT/pmc_mfma.json" 2> "$OUT/pmc_mfma.err"
# keep the merge-back small: only the stats and counter tables
find "$OUT" -name "*kernel_trace.csv" -delete
tail -1 "$OUT/bench_default.json"
