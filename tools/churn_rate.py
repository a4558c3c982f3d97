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
k = int(frac * n)
rng = np.random.default_rng(0)
kills = [np.sort(rng.choice(n, k, replace=False)).astype(np.int32) for _ in range(ticks)]     # prepared outside the timed loops
news = [(np.arange(k) + t * k) % (8 * n) for t in range(ticks)]
new_s = [np.ascontiguousarray(pool[i]) for i in news]
new_q = [np.ascontiguousarray(pdq[i].reshape(-1, 3)) for i in news]
tail = np.arange(n - k, n, dtype=np.int32)
qoff = np.arange(k + 1, dtype=np.int64) * 4
vd5 = np.full(k, 5.0)
for label, inc, churn in (("static", True, False), ("incremental", True, True), ("one_call", True, True), ("host_mirror", False, True)):
    t_run = ticks if label != "host_mirror" else max(20, ticks // 20)
    e = Engine(parameters.default_pod("twod"), n)
    e.set_incremental(inc)
    e.add_agents(s0, 5.0)
    e.set_dest_queue(np.arange(n), off, dq, reset=True)
    e.step(300, sync=True)
    calls = 0.0
    t0 = time.perf_counter()
    for t in range(t_run):
        if churn and k:
            c0 = time.perf_counter()
            if label == "one_call":                                    # csf_replace_agents: leave + arrive + queues in one call
                e.replace_agents(kills[t], new_s[t], vd5, qoff, new_q[t])
            else:
                e.remove_agents(kills[t])
                e.add_agents(new_s[t], 5.0)
                e.set_dest_queue(tail, qoff, new_q[t], reset=True)
            calls += time.perf_counter() - c0
        e.step(1)
    e.sync()
    dt = time.perf_counter() - t0
    healthy = bool(np.isfinite(e.state()).all() and (e.status() == 0).all())
    out[label] = {"us_per_tick": dt / t_run * 1e6, "host_us_in_population_calls": calls / t_run * 1e6, "ticks": t_run,
                  "healthy": healthy}
    e.close()
out["incremental_over_static"] = out["incremental"]["us_per_tick"] / out["static"]["us_per_tick"]
out["one_call_over_static"] = out["one_call"]["us_per_tick"] / out["static"]["us_per_tick"]
print(json.dumps(out))
