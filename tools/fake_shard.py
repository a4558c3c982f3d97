#!/usr/bin/env python3
"""Per-rank compute of a w-way index shard, emulated on one GPU (CSF_FAKE_SHARD=r/w: only rank r's receiver block is
computed, no communicator): bench.py's tick and kernel MEDIANS (min / max beside them) for the unsharded population and ranks 0 of
2, 4 and 8, with the per-agent launch behind the pair launch (CSF_CHASE=0) and beside it (CSF_CHASE=2), each with optional
environment variants (A/B of grid shapes).   tools/fake_shard.py [VAR=val,VAR=val ...]  -> JSON lines"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
variants = sys.argv[1:] or [""]
for shard in ("unsharded", "0/2", "0/4", "0/8"):
    for chase in ("0", "2"):
        for var in variants:
            env = dict(os.environ, CSF_CHASE=chase)
            if shard != "unsharded":
                env["CSF_FAKE_SHARD"] = shard
            for kv in filter(None, var.split(",")):
                k, v = kv.split("=")
                env[k] = v
            runs = []
            for _ in range(3):      # (a tick of 35 us is at the mercy of the host: 35 ... 55 us from run to run on one box, either way - the median of three)
                out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "600", "--warmup", "30", "--cpu-ticks", "0",
                                      "--every-pair-steps", "0"], env=env, capture_output=True, text=True)
                try:
                    runs.append(json.loads(out.stdout.strip().splitlines()[-1]))
                except Exception:  # noqa: BLE001
                    pass
            if not runs:
                print(json.dumps({"CSF_FAKE_SHARD": shard, "CSF_CHASE": chase, "variant": var, "error": out.stderr[-300:]}))
                continue
            runs.sort(key=lambda r: r["kernels_us"]["tick"])
            b = runs[len(runs) // 2]
            k, st = b["kernels_us"], b["kernels_us_stats"]
            print(json.dumps({"CSF_FAKE_SHARD": shard, "per_agent_launch": "beside the pair launch" if chase == "2" else "behind the pair launch", "variant": var,
                              "tick_us": k["tick"], "tick_us_of_three_runs": [r["kernels_us"]["tick"] for r in runs], "pair_us_median": k["pair"], "pair_us_min_max": st["pair"] and [st["pair"]["min"], st["pair"]["max"]],
                              "agent_us_median": k["agent"], "agent_us_min_max": st["agent"] and [st["agent"]["min"], st["agent"]["max"]],
                              "ticks_side_by_side": b["per_agent_launch"]["timed_engine"]["ticks_side_by_side"], "healthy": b["healthy"]}), flush=True)
