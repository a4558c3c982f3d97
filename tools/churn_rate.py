#!/usr/bin/env python3
"""Tick rate with road users arriving and leaving EVERY tick (the traffic of SUMO co-simulation, intersection.py:458-634)
against the static population: N = 16 384 TwoDBicycle, +-5 % of the population replaced per tick, through the array-level
API (csf_remove_agents + csf_add_agents + csf_set_dest_queue + csf_step(1) per tick).  Prints one JSON object."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synthetic_population  # noqa: E402
from cyclistsocialforce_amd import parameters  # noqa: E402
from cyclistsocialforce_amd.engine import Engine  # noqa: E402

n, box, ticks = 16384, 200.0, int(sys.argv[1]) if len(sys.argv) > 1 else 1000
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
s0, off, dq = synthetic_population(n, box)
pool, _, pdq = synthetic_population(8 * n, box, seed=1)
pdq = pdq.reshape(-1, 4, 3)
out = {"agents": n, "ticks": ticks, "churn_per_tick": frac}
for label, inc, churn in (("static", True, False), ("incremental", True, True), ("host_mirror", False, True)):
    t_run = ticks if label != "host_mirror" else max(20, ticks // 20)
    e = Engine(parameters.default_pod("twod"), n)
    e.set_incremental(inc)
    e.add_agents(s0, 5.0)
    e.set_dest_queue(np.arange(n), off, dq, reset=True)
    e.step(300, sync=True)
    rng = np.random.default_rng(0)
    k = int(frac * n)
    nxt = 0
    t0 = time.perf_counter()
    for t in range(t_run):
        if churn:
            kill = np.sort(rng.choice(n, k, replace=False))
            new = (np.arange(k) + nxt) % (8 * n)
            nxt += k
            e.remove_agents(kill)
            e.add_agents(pool[new], 5.0)
            e.set_dest_queue(np.arange(n - k, n), np.arange(k + 1) * 4, pdq[new].reshape(-1, 3), reset=True)
        e.step(1)
    e.sync()
    dt = time.perf_counter() - t0
    healthy = bool(np.isfinite(e.state()).all() and (e.status() == 0).all())
    out[label] = {"us_per_tick": dt / t_run * 1e6, "ticks": t_run, "healthy": healthy}
    e.close()
out["incremental_over_static"] = out["incremental"]["us_per_tick"] / out["static"]["us_per_tick"]
print(json.dumps(out))
