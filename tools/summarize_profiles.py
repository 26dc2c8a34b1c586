"""Turn gpurun_out/<tag>/ (written by tools/profile_round.sh on the GPU box) into the committed profiles/<tag>_* files."""
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")


_CXXFILT = "c++filt"
_memo = {}


def short(name):
    """`conv3x3_halo_kernel<256, 192, 4, 2>` from the demangled or mangled kernel name rocprofv3 reports."""
    if name not in _memo:
        n = name
        if n.startswith("_Z"):
            n = subprocess.run([_CXXFILT, n.replace("DF16b", "u6__bf16")], capture_output=True, text=True).stdout.strip() or n
            m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", n)           # binutils gives up on some plain names: take <len><name>
            if m:
                n = n[m.end():m.end() + int(m.group(1))]
        n = re.sub(r"^void ", "", n).replace("(anonymous namespace)::", "").replace("__bf16", "bf16")
        depth, cut = 0, len(n)
        for i, ch in enumerate(n):                       # strip the argument list, keep template arguments
            depth += ch == "<"
            depth -= ch == ">"
            if ch == "(" and depth == 0:
                cut = i
                break
        _memo[name] = n[:cut]
    return _memo[name]


def last_json(path):
    with open(path) as f:
        lines = [l for l in f if l.startswith("{")]
    return json.loads(lines[-1])


for n in ("bench_default", "bench_under_rocprof"):
    with open(os.path.join(dst, f"{tag}_{n}.json"), "w") as f:
        json.dump(last_json(os.path.join(src, n + ".json")), f, indent=1)
        f.write("\n")
stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)[0]
shutil.copy(stats, os.path.join(dst, f"{tag}_bench_kernel_stats.csv"))


def counters(sub, counter):
    path = glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True)[0]
    acc = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            d = acc.setdefault(short(row["Kernel_Name"]), {})      # dispatch -> summed over XCD rows
            d[row["Dispatch_Id"]] = d.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
    return {k: (len(v), sum(v.values()) / len(v)) for k, v in acc.items()}


fetch, write = counters("pmc_fetch", "FETCH_SIZE"), counters("pmc_write", "WRITE_SIZE")
kern = {}
for k in sorted(set(fetch) | set(write)):
    n, fkib = fetch.get(k, (0, 0.0))
    _, wkib = write.get(k, (0, 0.0))
    rb, wb = int(fkib * 1024 * 2), int(wkib * 1024)                # FETCH_SIZE doubled: gfx950 correction (MI355X_MICROARCH.md)
    kern[k] = {"launches": n, "read_bytes": rb, "write_bytes": wb, "hbm_bytes": rb + wb}
bd = last_json(os.path.join(src, "bench_default.json"))
wl = bd["config"]["workload"]
key = "config%d/%s/B%d/%s" % (int(wl.split("configs[")[1][0]) + 1, bd["dtype"], bd["config"]["global_batch"] // bd["n_gpus"],
                              wl.split("latent (4,")[1].split(")")[0].replace(",", "x"))
out = {"note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `python bench.py --no-cpu-baseline "
               "--no-secondary --steps 4 --warmup 3` (the default workload); per-launch averages in bytes. FETCH_SIZE is doubled (gfx950 "
               "reports half of a wide coalesced read, MI355X_MICROARCH.md §HBM); counter unit = KiB.",
       "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py --no-cpu-baseline --no-secondary --steps 4 --warmup 3 (tools/profile_round.sh)",
       "workload_key": key, "kernels": kern}
with open(os.path.join(dst, f"{tag}_pmc_hbm_traffic.json"), "w") as f:
    json.dump(out, f, indent=1)
    f.write("\n")

# ---- matrix-pipe occupancy and clock per kernel (third PMC pass)
mp = glob.glob(os.path.join(src, "pmc_mfma", "**", "*counter_collection.csv"), recursive=True)
if mp:
    acc, dur = {}, {}
    with open(mp[0]) as f:
        for row in csv.DictReader(f):
            k = short(row["Kernel_Name"])
            d = acc.setdefault(k, {}).setdefault(row["Dispatch_Id"], {})
            d[row["Counter_Name"]] = d.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
            dur.setdefault(k, {})[row["Dispatch_Id"]] = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
    rows = {}
    for k, disp in acc.items():
        n = len(disp)
        tot = {}
        for d in disp.values():
            for c, v in d.items():
                tot[c] = tot.get(c, 0.0) + v
        ns = sum(dur[k].values())
        gui = tot.get("GRBM_GUI_ACTIVE", 0.0) / 8.0                       # the counter is summed over the 8 XCDs
        rows[k] = {"launches": n, "avg_us": round(ns / n / 1e3, 1), "total_ms": round(ns / 1e6, 2),
                   "mfma_busy_frac": round(tot.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * gui), 4) if gui else None,
                   "clock_ghz": round(gui / ns, 3) if ns else None,
                   "mfma_insts_per_launch": int(tot.get("SQ_INSTS_MFMA", 0.0) / n), "valu_insts_per_launch": int(tot.get("SQ_INSTS_VALU", 0.0) / n),
                   "wave_cycles_split": {c: round(tot.get(c, 0.0) / tot["SQ_WAVE_CYCLES"], 3) for c in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY")}
                   if tot.get("SQ_WAVE_CYCLES") else None}
    top = dict(sorted(rows.items(), key=lambda kv: -kv[1]["total_ms"])[:24])
    with open(os.path.join(dst, f"{tag}_pmc_mfma_clock.json"), "w") as f:
        json.dump({"note": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY "
                           "SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE -- python3 bench.py --no-cpu-baseline --no-secondary --steps 4 --warmup 3. "
                           "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8); clock_ghz = GRBM_GUI_ACTIVE / 8 / duration "
                           "(reads high on dispatches shorter than ~0.3 ms, MI355X_MICROARCH.md); wave_cycles_split = share of SQ_WAVE_CYCLES spent issuing / "
                           "stalled at issue (pipe busy, dependencies) / parked on s_waitcnt or a barrier. Profiled runs clock lower than un-profiled ones.",
                   "kernels": top}, f, indent=1)
        f.write("\n")
print("wrote", sorted(os.listdir(dst)))
