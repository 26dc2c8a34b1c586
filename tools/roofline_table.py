#!/usr/bin/env python3
"""Per-kernel table of one profile round: launches and time per U-Net step, PMC bytes per launch (FETCH_SIZE x 2 / WRITE_SIZE as summarize_profiles.py
corrected them), the HBM rate those bytes imply, MFMA-busy and clock — from profiles/<tag>_bench_kernel_stats.csv, <tag>_pmc_hbm_traffic.json and
<tag>_pmc_mfma_clock.json.      python tools/roofline_table.py r05 > profiles/r05_kernel_table.md"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = sys.argv[:2] if len(sys.argv) > 1 else [sys.argv[0], "r05"]
tag = sys.argv[1]
exec(open(os.path.join(ROOT, "tools", "summarize_profiles.py")).read().split("def last_json")[0].split("ROOT = ")[0] + "\n" +
     "\n".join(l for l in open(os.path.join(ROOT, "tools", "summarize_profiles.py")).read().split("def last_json")[0].splitlines()
               if not l.startswith(("tag =", "src, dst"))))
P = os.path.join(ROOT, "profiles")
rows = list(csv.DictReader(open(os.path.join(P, f"{tag}_bench_kernel_stats.csv"))))
hbm = {k.replace(" ", ""): v for k, v in json.load(open(os.path.join(P, f"{tag}_pmc_hbm_traffic.json")))["kernels"].items()}
mf = {k.replace(" ", ""): v for k, v in json.load(open(os.path.join(P, f"{tag}_pmc_mfma_clock.json")))["kernels"].items()}
bench = json.load(open(os.path.join(P, f"{tag}_bench_under_rocprof.json")))
forwards = bench["steps"] + bench["warmup"]            # one U-Net evaluation (batch 128) per step of the profiled run
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"| kernel ({tag}, bf16x3, U-Net batch 128) | launches / step | avg µs | ms / step | % | PMC read + write GB / launch | implied TB/s | MFMA-busy | GHz |")
print("|---|---|---|---|---|---|---|---|---|")
for r in rows:
    ms = float(r["TotalDurationNs"]) / 1e6 / forwards
    if ms < 0.05:
        continue
    n = short(r["Name"])[:72]
    key = n.replace(" ", "")
    h, m = hbm.get(key), mf.get(key)
    avg = float(r["AverageNs"]) / 1e3
    gb = f"{h['read_bytes'] / 1e9:.2f} + {h['write_bytes'] / 1e9:.2f}" if h else "—"
    tbs = f"{h['hbm_bytes'] / 1e12 / (avg * 1e-6):.2f}" if h else "—"
    print(f"| `{n}` | {int(r['Calls']) / forwards:.1f} | {avg:.1f} | {ms:.2f} | {float(r['TotalDurationNs']) / tot * 100:.1f} | {gb} | {tbs} | "
          f"{m['mfma_busy_frac']:.2f} | {m['clock_ghz']:.2f} |" if m else f"| `{n}` | {int(r['Calls']) / forwards:.1f} | {avg:.1f} | {ms:.2f} | {float(r['TotalDurationNs']) / tot * 100:.1f} | {gb} | {tbs} | — | — |")
print(f"\nsum of kernel time: {tot / 1e6 / forwards:.2f} ms per step over {forwards} profiled steps ({bench['ms_per_step']} ms per step wall under rocprofv3)")
