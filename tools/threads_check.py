#!/usr/bin/env python3
"""Engines on several host threads at once (GPU box): four threads, each with engines of its own (create - populate - ticks in calls of
varying length - population calls - read-backs - destroy, repeatedly), against the same work done on one thread: the states must be
equal bit for bit.  The C ABI is not re-entrant per HANDLE (include/csf.h); distinct handles on distinct threads must be.
    tools/threads_check.py [threads] [rounds]"""
import json
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synthetic_population  # noqa: E402
from cyclistsocialforce_amd import parameters  # noqa: E402
from cyclistsocialforce_amd.engine import Engine  # noqa: E402

threads = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
SIZES = (3, 700, 5000, 16384)


def work(seed):
    out = []
    for r in range(rounds):
        n = SIZES[(seed + r) % len(SIZES)]
        s0, off, dq = synthetic_population(n, 30.0 if n < 100 else 200.0 * (n / 16384) ** 0.5, seed=seed * 100 + r)
        e = Engine(parameters.default_pod(("twod", "planarpoint")[(seed + r) % 2]), n + 64)
        st = s0 if (seed + r) % 2 == 0 else s0[:, :4]
        e.add_agents(st, 5.0)
        e.set_dest_queue(np.arange(n), off, dq, reset=True)
        for k in (1, 70, 7, 130):
            e.step(k)
        if n > 100:
            e.remove_agents(np.arange(1, n, 53))
            e.add_agents(st[:5] + 0.5, 5.0)
            e.set_dest_queue(np.arange(e.n - 5, e.n), off[:6], dq[:20], reset=True)
        e.step(40)
        fx, fy = e.calc_forces()
        out.append((e.state().copy(), np.c_[fx, fy].copy(), e.status().copy()))
        e.close()
    return out


serial = [work(s) for s in range(threads)]
res = [None] * threads
errs = []


def run(s):
    try:
        res[s] = work(s)
    except Exception as ex:  # noqa: BLE001
        errs.append(repr(ex))


ts = [threading.Thread(target=run, args=(s,)) for s in range(threads)]
for t in ts:
    t.start()
for t in ts:
    t.join()
same = not errs and all(all(np.array_equal(a[k], b[k]) for k in range(3)) for s in range(threads) for a, b in zip(serial[s], res[s]))
print(json.dumps({"threads": threads, "engines_per_thread": rounds, "errors": errs, "same_as_one_thread": bool(same)}))
