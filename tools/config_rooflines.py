#!/usr/bin/env python3
"""profiles/<TAG>_config{1..5}_{line.json,kernel_stats.csv,kernel_outliers.txt,pmc.json} -> profiles/<TAG>_config_rooflines.json:
one `roofline` object per BASELINE config for its dominant kernel, the way bench.py forms it - (100 op-equivalents x pairs evaluated
+ 25 x sources tested) per launch / the kernel's MEDIAN launch duration / 157.3 TFLOP/s - with the duration taken from the
rocprofv3 kernel trace of the engine's OWN choice of kernels (so that the one-launch tick and the one-wave kernel, which the engine's
HIP-event sampling cannot time without leaving them, are covered), the HIP-event median beside it where there is one, and the HBM
traffic per launch from the PMC passes.    tools/config_rooflines.py [TAG]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import HBM_PEAK_GBS, OPS_PER_PAIR, OPS_PER_TEST, VALU_PEAK_TFLOPS  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r6"
P = os.path.join(ROOT, "profiles")
DOMINANT = {"1": "small_tick_kernel", "2": "mid_tick_kernel", "3": "pair_cull_kernel", "4": "pair_cull_kernel", "5": "pair_cull_kernel"}
TICKS_PER_LAUNCH = {"1": 100}
out = {}
for c, needle in DOMINANT.items():
    line = json.load(open(os.path.join(P, f"{tag}_config{c}_line.json")))
    pmc = json.load(open(os.path.join(P, f"{tag}_config{c}_pmc.json")))
    # median / min / max of the dominant kernel from the trace (tools/kernel_outliers.py prints them under the kernel's name)
    txt = open(os.path.join(P, f"{tag}_config{c}_kernel_outliers.txt")).read()
    m = re.search(re.escape(needle) + r"[^\n]*\n\s+launches (\d+)\s+min ([\d.]+)\s+median ([\d.]+)\s+mean ([\d.]+)\s+max ([\d.]+)", txt)
    launches, tmin, tmed, tmean, tmax = int(m.group(1)), *(float(m.group(k)) for k in (2, 3, 4, 5))
    per = TICKS_PER_LAUNCH.get(c, 1)
    ev, te = line["roofline"]["pairs_evaluated"], line["roofline"]["sources_tested"]
    ops = None if ev is None else (OPS_PER_PAIR * ev + OPS_PER_TEST * te) * per
    kern = next((v for k, v in pmc["kernels"].items() if needle in k), {})
    traffic = kern.get("hbm_bytes_per_launch")
    roof = {"bound": "valu" if c in "345" else "latency (one launch per tick; the VALU fraction says how little of the device a tick of this size can use)",
            "kernel": next((k for k in pmc["kernels"] if needle in k), needle), "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
            "launch_us": tmed, "launch_us_is": "median over the launches of a rocprofv3 --kernel-trace run", "launch_us_min_med_max": [tmin, tmed, tmax],
            "launch_us_mean": tmean, "launches": launches, "ticks_per_launch": per,
            "launch_us_hip_events_median": line["roofline"].get("launch_us") if c in "345" else None,
            "pairs_evaluated": ev, "sources_tested": te, "counted": line["roofline"]["counted"],
            "achieved": None if ops is None else ops / (tmed * 1e-6) / 1e12, "traffic": traffic,
            "hbm_frac": None if traffic is None else traffic / (tmed * 1e-6) / 1e9 / HBM_PEAK_GBS,
            "valu_issue_occupancy": kern.get("valu_issue_occupancy"),
            "counters_per_launch": kern.get("counters_per_launch")}
    roof["frac"] = None if roof["achieved"] is None else roof["achieved"] / VALU_PEAK_TFLOPS
    out[c] = {"config": line["config"], "ms_per_tick": line["ms_per_tick"], "agent_steps_per_s": line.get("agent_steps_per_s"),
              "kernels_us_hip_events": {k: line.get(k) for k in ("pair_us", "road_us", "agent_us")}, "roofline": roof,
              "build_id": pmc["build_id"], "sources": [f"profiles/{tag}_config{c}_{s}" for s in ("line.json", "kernel_stats.csv", "kernel_outliers.txt", "pmc.json")]}
json.dump(out, open(os.path.join(P, f"{tag}_config_rooflines.json"), "w"), indent=1)
for c, v in out.items():
    r = v["roofline"]
    print(f"config {c}: {v['ms_per_tick'] * 1e3:9.1f} us/tick  {r['kernel'][:52]:52s} {r['launch_us']:9.1f} us  frac {r['frac'] if r['frac'] is None else round(r['frac'], 4)}  hbm {r['hbm_frac'] if r['hbm_frac'] is None else round(r['hbm_frac'], 4)}")
