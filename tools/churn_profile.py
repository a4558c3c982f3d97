#!/usr/bin/env python3
"""Where a tick with arrivals and departures spends its time: the population calls and csf_step on the host (wall clock
of each call), the pair kernel of every tick (time stamps of its dispatch), and the whole loop.  N = 16 384 TwoDBicycle.
usage: churn_profile.py TICKS FRACTION"""
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

n, box, ticks = 16384, 200.0, int(sys.argv[1]) if len(sys.argv) > 1 else 300
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
s0, off, dq = synthetic_population(n, box)
pool, _, pdq = synthetic_population(8 * n, box, seed=1)
pdq = pdq.reshape(-1, 4, 3)
k = int(frac * n)
rng = np.random.default_rng(0)
kills = [np.sort(rng.choice(n, k, replace=False)).astype(np.int32) for _ in range(ticks)]
news = [(np.arange(k) + t * k) % (8 * n) for t in range(ticks)]
new_s = [np.ascontiguousarray(pool[i]) for i in news]
new_q = [np.ascontiguousarray(pdq[i].reshape(-1, 3)) for i in news]
tail = np.arange(n - k, n, dtype=np.int32)
qoff = np.arange(k + 1, dtype=np.int64) * 4
e = Engine(parameters.default_pod("twod"), n)
e.set_incremental(True)
e.add_agents(s0, 5.0)
e.set_dest_queue(np.arange(n), off, dq, reset=True)
e.step(300, sync=True)
out = {"agents": n, "ticks": ticks, "churn_per_tick": frac}
for profiled in (False, True):
    if profiled:
        e.profile(1)
    t_rm = t_add = t_q = t_step = 0.0
    t0 = time.perf_counter()
    for t in range(ticks):
        a = time.perf_counter()
        e.remove_agents(kills[t])
        b = time.perf_counter()
        e.add_agents(new_s[t], 5.0)
        c = time.perf_counter()
        e.set_dest_queue(tail, qoff, new_q[t], reset=True)
        d = time.perf_counter()
        e.step(1)
        f = time.perf_counter()
        t_rm += b - a; t_add += c - b; t_q += d - c; t_step += f - d
    t1 = time.perf_counter()
    e.sync()
    t2 = time.perf_counter()
    r = {"loop_us_per_tick": (t2 - t0) / ticks * 1e6, "drain_us": (t2 - t1) * 1e6,
         "host_us": {"remove": t_rm / ticks * 1e6, "add": t_add / ticks * 1e6, "set_dest_queue": t_q / ticks * 1e6,
                     "step": t_step / ticks * 1e6}}
    if profiled:
        us = e.profile_samples()
        kern = e.profile_kernels()
        r["pair_us"] = {"mean": float(np.mean(us)), "min": float(np.min(us)), "max": float(np.max(us)),
                        "first_40": [round(float(x), 1) for x in us[:40]]}
        r["kernels"] = kern
    out["profiled" if profiled else "plain"] = r
print(json.dumps(out))
