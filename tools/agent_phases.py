#!/usr/bin/env python3
"""Per-agent kernel time by phase at N = 16 384: destination force + combine (csf_calc_forces), integrate alone
(csf_apply_forces), and the whole tick (csf_step) - time stamps of the kernel's own dispatch."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synthetic_population  # noqa: E402
from cyclistsocialforce_amd import parameters  # noqa: E402
from cyclistsocialforce_amd.engine import Engine  # noqa: E402

out = {}
for model in ("twod", "invpend", "bicycle"):
    n = 16384
    s0, off, dq = synthetic_population(n, 200.0)
    if model == "invpend":
        s0 = np.c_[s0, np.zeros(n)]
    e = Engine(parameters.default_pod(model), n)
    e.add_agents(s0, 5.0)
    e.set_dest_queue(np.arange(n), off, dq, reset=True)
    e.step(100, sync=True)
    fx, fy = e.forces()
    res = {}
    e.profile(1)
    for label, call in (("dest+combine", lambda: e.calc_forces()), ("integrate", lambda: e.apply_forces(fx, fy)), ("tick", lambda: e.step(1))):
        e.profile_kernels()
        for _ in range(64):
            call()
        e.sync()
        k = e.profile_kernels()
        res[label] = round(k["agent"][0] / max(k["agent"][1], 1) * 1e3, 2)
    out[model] = res
    e.close()
print(json.dumps(out))
