#!/usr/bin/env python3
"""Resident tick rate (Engine.step(K), no per-tick read-back) for small populations: the reference's own scale.
Three ways: the one-launch tick with all K ticks in ONE launch (csf_tick.hip: what step_n(K) does), the same kernel
with one launch per tick (what SocialForceIntersection.step() issues), and the general path (CSF_FUSED=0: pair kernel +
per-agent kernel)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synthetic_population  # noqa: E402
from cyclistsocialforce_amd import parameters  # noqa: E402
from cyclistsocialforce_amd.engine import Engine  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [3, 24, 128, 256, 512, 1024, 2048]
for model in ("twod", "invpend", "planarpoint", "bicycle"):
    for n in sizes:
        row = []
        for fused, per_tick in ((1, False), (1, True), (0, False)):
            os.environ["CSF_FUSED"] = str(fused)
            s0, off, dq = synthetic_population(n, max(40.0, (n / 0.41) ** 0.5))
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
            if per_tick:
                for _ in range(K):
                    e.step(1)
                e.sync()
            else:
                e.step(K, sync=True)
            dt = time.perf_counter() - t0
            row.append(dt / K * 1e6)
            name = e.count_pairs()[1]
            e.close()
        print(f"{model:12s} N={n:5d}: one launch for all ticks {row[0]:7.2f} us per tick | a launch per tick {row[1]:7.2f} | general path {row[2]:7.2f}")
