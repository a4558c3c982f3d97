#!/usr/bin/env python3
"""The per-agent launch beside the pair launch (CSF_CHASE=1, csf_engine.hip: enqueue_chase_tick) against the two launches in turn
(CSF_CHASE=0) on one box: bit-identity of the states after `--check` ticks, then microseconds per tick of both, alternating, with the
kernels' medians.   tools/chase_ab.py [--agents N] [--box L] [--model twod] [--check 200] [--ticks 1000] [--rounds 3] [CSF_X=..]"""
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


def arg(name, dflt, cast=int):
    return cast(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else dflt


n, box, model = arg("--agents", 16384), arg("--box", 200.0, float), arg("--model", "twod", str)
check, ticks, rounds = arg("--check", 200), arg("--ticks", 1000), arg("--rounds", 3)
for kv in sys.argv[1:]:
    if kv.startswith("CSF_") and "=" in kv:
        os.environ[kv.split("=")[0]] = kv.split("=", 1)[1]
s0, off, dq = synthetic_population(n, box)
if model == "invpend":
    s0 = np.c_[s0, np.zeros(n)]
elif model == "planarpoint":
    s0 = s0[:, :4]


def make(chase):
    os.environ["CSF_CHASE"] = str(chase)
    e = Engine(parameters.default_pod(model), n)
    e.add_agents(s0, 5.0)
    e.set_dest_queue(np.arange(n), off, dq, reset=True)
    return e


solo = arg("--solo", -1)
if solo >= 0:                                              # one engine only (two engines = four streams on the runtime's four hardware queues)
    e = make(solo)
    e.step(130, sync=True)
    for r in range(rounds):
        t0 = time.perf_counter()
        e.step(ticks, sync=True)
        print(json.dumps({"CSF_CHASE": solo, "solo": True, "round": r, "tick_us": (time.perf_counter() - t0) / ticks * 1e6, "side_by_side_ticks": e.chase_ticks()}), flush=True)
    sys.exit(0)
a, b = make(0), make(2)
for k in range(0, check, 50):
    a.step(min(50, check - k), sync=True)
    b.step(min(50, check - k), sync=True)
sa, pa, za, _ = a.state(with_nav=True)
sb, pb, zb, _ = b.state(with_nav=True)
fa, fb = np.c_[a.forces()], np.c_[b.forces()]
same = bool(np.array_equal(sa, sb) and np.array_equal(pa, pb) and np.array_equal(za, zb) and np.array_equal(fa, fb))
print(json.dumps({"check_ticks": check, "bit_identical": same, "max_abs_state_diff": float(np.abs(sa - sb).max()), "max_abs_force_diff": float(np.abs(fa - fb).max()),
                  "side_by_side_ticks": b.chase_ticks(), "of": check, "status_flags": int((b.status() != 0).sum()), "near_dropped": b.near_dropped()}), flush=True)
res = {0: [], 1: []}
for r in range(rounds):
    for chase, e in ((0, a), (1, b)):
        e.step(64, sync=True)
        e.profile(16)
        t0 = time.perf_counter()
        e.step(ticks)
        t_enq = time.perf_counter() - t0                      # (the host's share: the call returns when everything is enqueued)
        e.sync()
        dt = time.perf_counter() - t0
        st = e.profile_stats()
        e.profile_kernels()
        e.profile(0)
        res[chase].append(dt / ticks * 1e6)
        print(json.dumps({"CSF_CHASE": chase, "round": r, "tick_us": dt / ticks * 1e6, "host_enqueue_us_per_tick": t_enq / ticks * 1e6, "pair_us": st["pair"] and st["pair"]["median"],
                          "agent_us": st["agent"] and st["agent"]["median"], "agent_us_min_max": st["agent"] and [st["agent"]["min"], st["agent"]["max"]]}), flush=True)
print(json.dumps({"agents": n, "model": model, "tick_us_in_turn_median": float(np.median(res[0])), "tick_us_side_by_side_median": float(np.median(res[1])),
                  "gain_us": float(np.median(res[0]) - np.median(res[1]))}))
