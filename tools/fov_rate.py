#!/usr/bin/env python3
"""Tick time at N = 16 384 TwoDBicycle for several fields of view (parameters.py: hfov)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synthetic_population  # noqa: E402
from cyclistsocialforce_amd import parameters  # noqa: E402
from cyclistsocialforce_amd.engine import Engine  # noqa: E402

n = 16384
s0, off, dq = synthetic_population(n, 200.0)
for hfov in (0.6, 2 * np.pi / 3, np.pi, 4.0, 2 * np.pi):
    e = Engine(parameters.default_pod("twod", hfov=float(hfov)), n)
    e.add_agents(s0, 5.0)
    e.set_dest_queue(np.arange(n), off, dq, reset=True)
    e.step(40, sync=True)
    K = 300
    t0 = time.perf_counter()
    e.step(K, sync=True)
    dt = time.perf_counter() - t0
    print(f"hfov {hfov:5.3f} rad: {dt / K * 1e6:7.1f} us per tick, {n * K / dt:.3e} agent-steps/s")
    e.close()
