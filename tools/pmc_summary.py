#!/usr/bin/env python3
"""Average the rocprofv3 --pmc counters of one kernel over its launches.

usage: pmc_summary.py DIR [kernel-substring]      (DIR as written by tools/pmc_passes.sh / pmc_ab.sh)
Prints one JSON object {counter: mean per launch}; the first launches (warm-up of bench.py) are included.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def summarise(root, needle="pair_cull_kernel"):
    acc, cnt = defaultdict(float), defaultdict(int)
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                if needle in row["Kernel_Name"]:
                    acc[row["Counter_Name"]] += float(row["Counter_Value"])
                    cnt[row["Counter_Name"]] += 1
    return {k: acc[k] / cnt[k] for k in sorted(acc)}, (max(cnt.values()) if cnt else 0)


if __name__ == "__main__":
    needle = sys.argv[2] if len(sys.argv) > 2 else "pair_cull_kernel"
    config = sys.argv[3] if len(sys.argv) > 3 else "N=16384 TwoDBicycle, bench.py --steps 20 --warmup 5"
    out, n = summarise(sys.argv[1], needle)
    names = set()
    for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as fh:
            names |= {row["Kernel_Name"] for row in csv.DictReader(fh) if needle in row["Kernel_Name"]}
    import hashlib

    lib = os.environ.get("CSF_LIB") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cyclistsocialforce_amd", "libcsf_hip.so")
    with open(lib, "rb") as fh:
        bid = hashlib.sha256(fh.read()).hexdigest()[:16]
    rec = {"kernel": sorted(names)[0] if names else needle, "config": config, "launches_averaged": n, "build_id": bid,
           "note": "rocprofv3 --pmc, one counter set per run (tools/pmc_passes.sh). FETCH_SIZE/WRITE_SIZE are reported in "
                   "KiB; on gfx950 FETCH_SIZE counts half of wide streaming reads (MI355X_MICROARCH.md), so HBM bytes "
                   "per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024."}
    if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
        rec["hbm_bytes_per_launch"] = (2 * out["FETCH_SIZE"] + out["WRITE_SIZE"]) * 1024
    if "SQ_ACTIVE_INST_VALU" in out and "GRBM_GUI_ACTIVE" in out:       # quad-cycles x 4 / (cycles per XCD x 1024 SIMDs)
        rec["valu_issue_occupancy"] = out["SQ_ACTIVE_INST_VALU"] * 4 / (out["GRBM_GUI_ACTIVE"] / 8 * 1024)
    rec["counters"] = out
    print(json.dumps(rec, indent=1))
