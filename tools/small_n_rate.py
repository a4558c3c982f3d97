#!/usr/bin/env python3
"""Resident tick rate (Engine.step(K), no per-tick read-back) for small populations: the reference's own scale."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synthetic_population  # noqa: E402
from cyclistsocialforce_amd import parameters  # noqa: E402
from cyclistsocialforce_amd.engine import Engine  # noqa: E402

for model in ("twod", "invpend", "planarpoint", "bicycle"):
    for n in (3, 24, 128, 256, 512):
        s0, off, dq = synthetic_population(n, 40.0)
        if model == "invpend":
            s0 = np.c_[s0, np.zeros(n)]
        elif model == "planarpoint":
            s0 = s0[:, :4]
        e = Engine(parameters.default_pod(model), n)
        e.add_agents(s0, 5.0)
        e.set_dest_queue(np.arange(n), off, dq, reset=True)
        e.step(20, sync=True)
        K = 2000
        t0 = time.perf_counter()
        e.step(K, sync=True)
        dt = time.perf_counter() - t0
        print(f"{model:12s} N={n:4d}: {dt / K * 1e6:7.2f} us per tick")
        e.close()
