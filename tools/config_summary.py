#!/usr/bin/env python3
"""One config's rocprofv3 passes (tools/profile_configs.sh) -> a JSON summary: per kernel of the config its launch count and average
duration (kernel_stats.csv) and the mean per launch of every counter collected; HBM bytes per launch = (2 FETCH_SIZE + WRITE_SIZE) KiB
(gfx950: FETCH_SIZE counts half of wide streaming reads, MI355X_MICROARCH.md).   tools/config_summary.py OUTDIR N"""
import csv
import glob
import hashlib
import json
import os
import sys
from collections import defaultdict

out, c = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for path in glob.glob(os.path.join(out, f"pmc{c}_*", "**", "*counter_collection.csv"), recursive=True):
    with open(path, newline="") as fh:
        for r in csv.DictReader(fh):
            k = r["Kernel_Name"].split("(")[0]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]] += 1
stats = {}
sp = os.path.join(out, f"config{c}_kernel_stats.csv")
if os.path.exists(sp):
    with open(sp, newline="") as fh:
        for r in csv.DictReader(fh):
            stats[r["Name"].split("(")[0]] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "min_us": float(r["MinNs"]) / 1e3,
                                              "max_us": float(r["MaxNs"]) / 1e3, "percent": float(r["Percentage"])}
with open(os.path.join(ROOT, "cyclistsocialforce_amd", "libcsf_hip.so"), "rb") as fh:
    bid = hashlib.sha256(fh.read()).hexdigest()[:16]
kernels = {}
for k in sorted(set(acc) | set(stats)):
    if not k.startswith("void csf::") and "csf::" not in k:
        continue
    m = {n: acc[k][n] / cnt[k][n] for n in sorted(acc[k])}
    rec = {"trace": stats.get(k), "launches_counted": max(cnt[k].values()) if cnt[k] else 0, "counters_per_launch": m}
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        rec["hbm_bytes_per_launch"] = (2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024
        if stats.get(k):
            rec["hbm_frac"] = rec["hbm_bytes_per_launch"] / (stats[k]["avg_us"] * 1e-6) / 8e12
    if "SQ_ACTIVE_INST_VALU" in m and "GRBM_GUI_ACTIVE" in m and m["GRBM_GUI_ACTIVE"] > 0:
        rec["valu_issue_occupancy"] = m["SQ_ACTIVE_INST_VALU"] * 4 / (m["GRBM_GUI_ACTIVE"] / 8 * 1024)
    kernels[k] = rec
print(json.dumps({"config": c, "build_id": bid, "note": "rocprofv3: --kernel-trace --stats and three --pmc passes, each its own run of "
                  f"tools/large_configs.py {c} --plain (the engine's own choice of kernels)", "kernels": kernels}, indent=1))
