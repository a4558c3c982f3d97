#!/usr/bin/env python3
"""Tick time of N = 16 384 TwoDBicycle with ONE parameter set (the culling pair kernel), with the same set through the
plain all-pairs kernel (CSF_PAIR_VARIANT=1 in a second process is not needed: K = 2 identical sets take the same path),
and with four different sets (csf_set_param_classes): what a mixed population pays.  Prints one JSON object."""
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

n, box, ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 16384, 200.0, 200
s0, off, dq = synthetic_population(n, box)
sets = {"one_set": [{}],
        "two_equal_sets": [{}, dict(d_arrived_inter=2.0 + 1e-9)],
        "four_sets": [{}, dict(hfov=1.2 * np.pi, f_0=10.0, sigma_0=0.6, sigma_1=5.5), dict(hfov=1.0, e_0=0.9, e_1=0.4),
                      dict(hfov=2.5, f_0=4.0)]}
out = {"agents": n, "ticks": ticks}
for label, recipes in sets.items():
    pods = [parameters.default_pod("twod", **kw) for kw in recipes]
    e = Engine(pods[0], n)
    e.add_agents(s0, 5.0)
    e.set_dest_queue(np.arange(n), off, dq, reset=True)
    if len(pods) > 1:
        e.set_param_classes(pods, np.arange(n) % len(pods))
    e.step(40, sync=True)
    t0 = time.perf_counter()
    e.step(ticks, sync=True)
    dt = time.perf_counter() - t0
    out[label] = {"us_per_tick": dt / ticks * 1e6, "kernel": e.count_pairs()[1],
                  "healthy": bool(np.isfinite(e.state()).all() and (e.status() == 0).all())}
    e.close()
print(json.dumps(out))
