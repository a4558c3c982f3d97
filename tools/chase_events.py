#!/usr/bin/env python3
"""Dispatch time stamps (HIP events of hipExtLaunchKernelGGL) of consecutive ticks, every tick sampled: where does a tick's time go with
the per-agent kernel beside the pair launch (CSF_CHASE=1) and behind it (CSF_CHASE=0)?   tools/chase_events.py [CSF_X=..]"""
import os
import re
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import bench
    from cyclistsocialforce_amd import parameters
    from cyclistsocialforce_amd.engine import Engine
    s0, off, dq = bench.synthetic_population(16384, 200.0)
    e = Engine(parameters.default_pod("twod"), 16384)
    e.add_agents(s0, 5.0)
    e.set_dest_queue(np.arange(16384), off, dq, reset=True)
    e.step(66, sync=True)
    e.profile(1)
    e.step(60, sync=True)
    k = e.profile_kernels()
    print("side-by-side ticks", e.chase_ticks(), "pair", k["pair"], "agent", k["agent"])
    e.close()
    sys.exit(0)
for chase in ("1", "0"):
    env = dict(os.environ, CSF_CHASE=chase, CSF_CHASE_CLOCK="/tmp/cc.bin")
    for kv in sys.argv[1:]:
        if "=" in kv:
            env[kv.split("=")[0]] = kv.split("=", 1)[1]
    r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
    rows = np.array([[float(x) for x in re.findall(r"-?\d+\.\d+", l)] for l in r.stderr.splitlines() if l.startswith("CHASE_EVENTS")])
    print(f"CSF_CHASE={chase}:", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:])
    if rows.size:
        m = np.median(rows, axis=0)
        print(f"   median of {len(rows)} ticks (us): per-agent kernel starts {m[0]:+.1f} / ends {m[1]:+.1f} relative to the pair launch's end; "
              f"next pair launch starts {m[2]:.1f} after the per-agent kernel's end; pair start -> pair start {m[3]:.1f}")
