#!/usr/bin/env python3
"""Tick time at N = 16 384 TwoDBicycle under both priority rules (intersection.py:739-741)."""
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
for rule, name in ((0, "unregulated"), (1, "p2r")):
    e = Engine(parameters.default_pod("twod", priority_rule=rule), n)
    e.add_agents(s0, 5.0)
    e.set_dest_queue(np.arange(n), off, dq, reset=True)
    e.step(40, sync=True)
    K = 400
    t0 = time.perf_counter()
    e.step(K, sync=True)
    dt = time.perf_counter() - t0
    print(f"{name:12s}: {dt / K * 1e6:7.1f} us per tick, {n * K / dt:.3e} agent-steps/s")
    e.close()
