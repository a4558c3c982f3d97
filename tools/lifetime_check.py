#!/usr/bin/env python3
"""Engines coming and going in one process (GPU box): device memory after every 50 of 300 create - populate - step - destroy rounds (a leak
shows as a slope), destroy with ticks still in flight, and a capacity no device can hold (NULL and a message, not an exception through
the C ABI).   tools/lifetime_check.py [rounds]"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synthetic_population  # noqa: E402
from cyclistsocialforce_amd import _ffi, parameters  # noqa: E402
from cyclistsocialforce_amd.engine import Engine  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n = 16384
s0, off, dq = synthetic_population(n, 200.0)
free = []
for r in range(rounds):
    e = Engine(parameters.default_pod(("twod", "invpend", "planarpoint")[r % 3]), n)
    st = s0 if r % 3 == 0 else (np.c_[s0, np.zeros(n)] if r % 3 == 1 else s0[:, :4])
    e.add_agents(st, 5.0)
    e.set_dest_queue(np.arange(n), off, dq, reset=True)
    e.step(70 if r % 10 == 0 else 3)          # (every tenth round past a re-binning; never waited for: destroy with ticks in flight)
    if r % 7 == 0:
        e.remove_agents(np.arange(0, n, 97))
        e.step(2)
    e.close()
    if r % 50 == 0 or r == rounds - 1:
        torch.cuda.synchronize()
        free.append(int(torch.cuda.mem_get_info()[0]))
L = _ffi.load()
pod = parameters.default_pod("twod")
huge = []
for cap in (1 << 33, 1 << 40, (1 << 62)):
    h = L.csf_create_v(C.byref(pod), C.sizeof(pod), _ffi.ABI_VERSION, cap, 0)
    huge.append([cap, bool(h), (L.csf_last_error(None) or b"").decode()[:120]])
    if h:
        L.csf_destroy(h)
e = Engine(pod, 64)      # and the library still works
e.add_agents(s0[:64], 5.0)
e.step(5, sync=True)
print(json.dumps({"rounds": rounds, "free_bytes_every_50_rounds": free, "lost_bytes_first_to_last": free[0] - free[-1] if free else None,
                  "lost_bytes_second_to_last": free[1] - free[-1] if len(free) > 1 else None, "huge_capacity": huge, "alive_after": bool(np.isfinite(e.state()).all())}))
