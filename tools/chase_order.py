#!/usr/bin/env python3
"""Does the side-by-side tick depend on WHICH engine of a process runs it?  Engines are created in the given order of CSF_CHASE values
(default 2 0 2 0), each is then timed in turn, three rounds.    tools/chase_order.py [2 0 2 0] [--torch]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if "--torch" in sys.argv:
    import torch
    torch.cuda.set_device(0)
    torch.zeros(4, device="cuda")
from bench import synthetic_population  # noqa: E402
from cyclistsocialforce_amd import parameters  # noqa: E402
from cyclistsocialforce_amd.engine import Engine  # noqa: E402

modes = [a for a in sys.argv[1:] if a in ("0", "1", "2")] or ["2", "0", "2", "0"]
s0, off, dq = synthetic_population(16384, 200.0)
engines = []
for m in modes:
    os.environ["CSF_CHASE"] = m
    e = Engine(parameters.default_pod("twod"), 16384)
    e.add_agents(s0, 5.0)
    e.set_dest_queue(np.arange(16384), off, dq, reset=True)
    engines.append(e)
if "--profiled-bystander" in sys.argv:      # bench.py: the timed engine has its event pool and has run its first ticks before the scratch engine runs
    os.environ["CSF_CHASE"] = "0"
    by = Engine(parameters.default_pod("twod"), 16384)
    by.add_agents(s0, 5.0)
    by.set_dest_queue(np.arange(16384), off, dq, reset=True)
    by.profile(16)
    by.step(8, sync=True)
    if "--torch" in sys.argv:
        torch.cuda.synchronize()
for e in engines:
    e.step(8, sync=True)
    if "--window-state" in sys.argv:          # bench.py: window_state(scratch) - a read-back and the counting launch - before the warm-up
        e.state(); e.count_pairs(detail=True)
    e.step(192, sync=True)
for r in range(3):
    row = []
    for m, e in zip(modes, engines):
        e.step(64, sync=True)
        t0 = time.perf_counter()
        e.step(600, sync=True)
        row.append(round((time.perf_counter() - t0) / 600 * 1e6, 1))
    print(json.dumps({"created_with_CSF_CHASE": modes, "tick_us": row, "side_by_side_ticks": [e.chase_ticks() for e in engines]}), flush=True)
