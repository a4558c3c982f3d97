#!/usr/bin/env python3
"""Soak run: N = 16 384 TwoDBicycle on routes that outlast the run, TICKS ticks in blocks of 1 000 with the block time and
the population's health after each; then the same with 0.2 % of the road users replaced every tick.  Prints one JSON line
per phase."""
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

ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 12000
n, box = 16384, 200.0
reach = tuple(50.0 * k for k in range(1, 16))
s0, off, dq = synthetic_population(n, box, reach=reach)
rows = len(reach) + 1
e = Engine(parameters.default_pod("twod"), n)
e.add_agents(s0, 5.0)
e.set_dest_queue(np.arange(n), off, dq, reset=True)
blocks = []
for b in range(ticks // 1000):
    t0 = time.perf_counter()
    e.step(1000, sync=True)
    blocks.append(round((time.perf_counter() - t0) * 1e3, 1))
    st = e.state()
    # (CSF_ST_SPLINE = 1 is no fault: a road user that has reached its last destination plans from coincident points and takes the
    # reference's fallback branch, vehicle.py:1416-1558 - past tick ~15 000 on these routes)
    assert np.isfinite(st).all() and ((e.status() & ~np.uint32(1)) == 0).all(), b
print(json.dumps({"phase": "static", "ticks": ticks, "us_per_tick_by_block_of_1000": blocks, "extent_m": float(np.ptp(st[:, 0]))}))
e.close()

pool, _, pdq = synthetic_population(4 * n, box, seed=1, reach=reach)
pdq = pdq.reshape(-1, rows, 3)
e = Engine(parameters.default_pod("twod"), n)
e.set_incremental(True)
e.add_agents(s0, 5.0)
e.set_dest_queue(np.arange(n), off, dq, reset=True)
rng = np.random.default_rng(0)
k = 32
tail = np.arange(n - k, n, dtype=np.int32)
qoff = np.arange(k + 1, dtype=np.int64) * rows
blocks = []
for b in range(ticks // 1000):
    t0 = time.perf_counter()
    for t in range(1000):
        kill = np.sort(rng.choice(n, k, replace=False)).astype(np.int32)
        new = rng.integers(0, 4 * n, k)
        e.remove_agents(kill)
        e.add_agents(pool[new], 5.0)
        e.set_dest_queue(tail, qoff, pdq[new].reshape(-1, 3), reset=True)
        e.step(1)
    e.sync()
    blocks.append(round((time.perf_counter() - t0) * 1e3, 1))
    st = e.state()
    assert np.isfinite(st).all() and e.n == n, b
    bad = int((e.status() != 0).sum())
    assert bad == 0, (b, bad)
print(json.dumps({"phase": "0.2 % replaced per tick", "ticks": ticks, "us_per_tick_by_block_of_1000": blocks}))
e.close()
