#!/usr/bin/env python3
"""Per-rank compute of a w-way index shard, emulated on one GPU (CSF_FAKE_SHARD=r/w: only rank r's receiver block is
computed, no communicator): bench.py's kernel times for the unsharded population and ranks 0 of 2, 4 and 8, each with
optional environment variants (A/B of grid shapes).   tools/fake_shard.py [VAR=val,VAR=val ...]  -> JSON lines"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
variants = sys.argv[1:] or [""]
for shard in ("unsharded", "0/2", "0/4", "0/8"):
    for var in variants:
        env = dict(os.environ)
        if shard != "unsharded":
            env["CSF_FAKE_SHARD"] = shard
        for kv in filter(None, var.split(",")):
            k, v = kv.split("=")
            env[k] = v
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "600", "--warmup", "30", "--cpu-ticks", "0",
                              "--every-pair-steps", "0"], env=env, capture_output=True, text=True)
        try:
            b = json.loads(out.stdout.strip().splitlines()[-1])
        except Exception:  # noqa: BLE001
            print(json.dumps({"CSF_FAKE_SHARD": shard, "variant": var, "error": out.stderr[-300:]}))
            continue
        k = b["kernels_us"]
        print(json.dumps({"CSF_FAKE_SHARD": shard, "variant": var, "tick_us": k["tick"], "pair": k["pair"], "agent": k["agent"],
                          "healthy": b["healthy"]}), flush=True)
