#!/usr/bin/env python3
"""Resident tick time of mid-size populations (GPU box): csf_step(K) of N TwoDBicycle road users, the tick as ONE launch
(csf_mid.hip) against a pair launch + a per-agent launch (CSF_FUSED_MID=0), and the host's time to enqueue a tick.  BASELINE config 2 (1 024 in 200 m x 200 m) and the headline's density.  One JSON line each."""
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

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
models = sys.argv[2].split(",") if len(sys.argv) > 2 else ["twod"]
REACH = tuple(50.0 * k for k in range(1, 14))


def run(tag, model, n, box, env):
    for k, v in env.items():
        os.environ[k] = v
    s0, off, dq = synthetic_population(n, box, reach=REACH)
    from cyclistsocialforce_amd import _ffi, engine as _engine
    width = _ffi.N_STATES[_engine.MODEL_IDS[model]]
    s0 = np.c_[s0, np.zeros((n, max(0, width - s0.shape[1])))][:, :width]
    e = Engine(parameters.default_pod(model), n)
    e.add_agents(s0, 5.0)
    e.set_dest_queue(np.arange(n), off, dq, reset=True)
    e.step(300, sync=True)
    best, enq = 1e9, 0.0
    for _ in range(3):
        t0 = time.perf_counter()
        e.step(K)
        t_enq = time.perf_counter() - t0
        e.sync()
        dt = time.perf_counter() - t0
        if dt < best:
            best, enq = dt, t_enq
    out = {"case": tag, "model": model, "agents": n, "box_m": round(box, 1), "env": env, "us_per_tick": best / K * 1e6,
           "host_enqueue_us_per_tick": enq / K * 1e6, "one_launch_ticks": e.mid_ticks(), "healthy": bool(np.isfinite(e.state()).all())}
    print(json.dumps(out), flush=True)
    e.close()
    for k in env:
        del os.environ[k]


for model in models:
    for n, box, tag in [(1024, 200.0, "config 2")] + [(n, float(np.sqrt(n / 0.41)), "headline density") for n in (64, 128, 256, 512, 1024, 2048, 3000)]:
        run(tag, model, n, box, {"CSF_MID_BELOW": "4000"})
        run(tag, model, n, box, {"CSF_FUSED_MID": "0"})
