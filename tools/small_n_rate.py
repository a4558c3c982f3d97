#!/usr/bin/env python3
"""Resident tick time of small populations (GPU box): csf_step(K) of N road users at the headline's density, three models,
with the host's time to enqueue a tick and the per-launch times of the pair and the per-agent kernel.  One JSON line per (model, N)."""
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

K = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
for model in ("bicycle", "twod", "invpend", "planarpoint"):
    for n in (3, 128, 1024):
        box = max(10.0, float(np.sqrt(n / 0.41)))
        s0, off, dq = synthetic_population(n, box, reach=tuple(50.0 * k for k in range(1, 14)))
        if model == "invpend":
            s0 = np.c_[s0, np.zeros(n)]
        elif model == "planarpoint":
            s0 = s0[:, :4]
        e = Engine(parameters.default_pod(model), n)
        e.add_agents(s0, 5.0)
        e.set_dest_queue(np.arange(n), off, dq, reset=True)
        e.step(300, sync=True)
        t0 = time.perf_counter()
        e.step(K)                                             # returns when the ticks are enqueued
        t_enq = time.perf_counter() - t0
        e.sync()
        dt = time.perf_counter() - t0
        e.profile(8)
        e.step(256, sync=True)
        prof = {k: ms * 1e3 / max(c, 1) for k, (ms, c) in e.profile_kernels().items()}
        print(json.dumps({"model": model, "agents": n, "us_per_tick": dt / K * 1e6, "host_enqueue_us_per_tick": t_enq / K * 1e6, "pair_us": prof["pair"], "agent_us": prof["agent"],
                          "healthy": bool(np.isfinite(e.state()).all())}), flush=True)
        e.close()
