#!/usr/bin/env python3
"""Device-clock stamps of the last 64 side-by-side ticks (CSF_CHASE_CLOCK; wall_clock64, 100 MHz): per tick the pair launch's first start
and last arrival, the gate's entry and exit, the per-agent kernel's first entry, last wave past its wait, last end - and how long after the
per-agent kernel's end the NEXT pair launch starts.     tools/chase_clock.py [--agents N] [--ticks 200] [--model twod] [CSF_X=..]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def arg(name, dflt, cast=int):
    return cast(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else dflt


n, box, ticks, model = arg("--agents", 16384), arg("--box", 200.0, float), arg("--ticks", 250), arg("--model", "twod", str)
for kv in sys.argv[1:]:
    if kv.startswith("CSF_") and "=" in kv:
        os.environ[kv.split("=")[0]] = kv.split("=", 1)[1]
os.environ["CSF_CHASE_CLOCK"] = "/tmp/chase_clock.bin"
import bench  # noqa: E402
from cyclistsocialforce_amd import parameters  # noqa: E402
from cyclistsocialforce_amd.engine import Engine  # noqa: E402

s0, off, dq = bench.synthetic_population(n, box)
if model == "invpend":
    s0 = np.c_[s0, np.zeros(n)]
e = Engine(parameters.default_pod(model), n)
e.add_agents(s0, 5.0)
e.set_dest_queue(np.arange(n), off, dq, reset=True)
e.step(66, sync=True)          # (past the re-binning at tick 64: the next 63 ticks are side by side without a break)
e.step(58, sync=True)
print("side-by-side ticks", e.chase_ticks())
e.close()
raw = np.fromfile("/tmp/chase_clock.bin", dtype=np.uint64)
last = int(raw[-1])
ck = raw[:1024].reshape(128, 8).astype(np.float64) / 100.0
rows = []
for r in range(last - 52, last - 1):                 # consecutive ticks of the last call
    a, b = ck[r & 127], ck[(r + 1) & 127]
    rows.append([a[1] - a[0], a[3] - a[1], a[4] - a[1], a[5] - a[1], a[6] - a[1], b[0] - a[6], b[0] - a[0], a[4] - a[3]])
m = np.median(np.array(rows), axis=0)
print(f"median over {len(rows)} consecutive ticks (us): pair first start -> last arrival {m[0]:.1f}; relative to that last arrival: gate exit {m[1]:+.1f}, "
      f"per-agent first entry {m[2]:+.1f} ({m[7]:.1f} after the gate's exit), last wave past its wait {m[3]:+.1f}, per-agent last end {m[4]:+.1f}; "
      f"next pair launch's first start {m[5]:.1f} after that; pair start -> pair start {m[6]:.1f}")
