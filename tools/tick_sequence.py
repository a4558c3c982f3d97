#!/usr/bin/env python3
"""Pair-kernel time of EVERY tick of a fresh process (bench.py population): why the first ticks after start-up run
slower than the steady state, and what sampling every tick with the kernels' own time stamps costs.
Prints one JSON object (committed as profiles/r2_tick_sequence.json)."""
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

n, box = 16384, 200.0
s0, off, dq = synthetic_population(n, box)
e = Engine(parameters.default_pod("twod"), n)
e.add_agents(s0, 5.0)
e.set_dest_queue(np.arange(n), off, dq, reset=True)
e.profile(1)
t0 = time.perf_counter()
e.step(1200, sync=True)
wall_first = time.perf_counter() - t0
us = e.profile_samples()
prof = e.profile_kernels()
cnt = prof["pair"][1]
out = {"agents": n, "ticks": int(cnt), "first_1200_ticks_wall_ms": wall_first * 1e3,
       "pair_us_tick_0_to_39": [round(float(x), 1) for x in us[:40]],
       "pair_us_mean_by_block_of_100": [round(float(us[k:k + 100].mean()), 1) for k in range(0, 1200, 100)],
       "agent_us_mean": prof["agent"][0] * 1e3 / max(prof["agent"][1], 1)}
# steady state: wall time per tick with and without per-tick sampling
for every in (0, 1, 8, 0, 1, 8):
    e.profile(every)
    t0 = time.perf_counter()
    e.step(600, sync=True)
    dt = time.perf_counter() - t0
    e.profile_kernels()
    out.setdefault(f"tick_us_profile_every_{every}", []).append(round(dt / 600 * 1e6, 2))
print(json.dumps(out))
