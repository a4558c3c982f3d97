#!/usr/bin/env python3
"""Tick time by population size for the pair-kernel variants of one TwoDBicycle parameter set: the engine's default (the
culling kernel; binned from 1024 road users) against the plain all-pairs kernel (CSF_PAIR_VARIANT=1), density as in the
headline workload (0.41 road users per m^2).  One process per variant (the variant is read once)."""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

MODEL = os.environ.get("VARIANT_MODEL", "twod")
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from bench import synthetic_population
    from cyclistsocialforce_amd import parameters
    from cyclistsocialforce_amd.engine import Engine
    res = {}
    for n in (64, 256, 512, 1024, 2048, 4096, 8192):
        box = float(np.sqrt(n / 0.41))
        s0, off, dq = synthetic_population(n, box)
        e = Engine(parameters.default_pod(MODEL), n)
        e.add_agents(s0, 5.0)
        e.set_dest_queue(np.arange(n), off, dq, reset=True)
        e.step(40, sync=True)
        K = 400
        t0 = time.perf_counter()
        e.step(K, sync=True)
        res[n] = round((time.perf_counter() - t0) / K * 1e6, 2)
        e.close()
    print(json.dumps(res))
else:
    out = {}
    for label, env in (("default", {}), ("plain", {"CSF_PAIR_VARIANT": "1"})):
        r = subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, **env), capture_output=True, text=True)
        out[label] = json.loads(r.stdout.strip().split("\n")[-1]) if r.returncode == 0 else r.stderr[-400:]
    out["model"] = MODEL
    print(json.dumps(out))
