#!/usr/bin/env python3
"""When does what happen in a tick whose per-agent kernel runs beside its pair launch (csf_engine.hip: enqueue_chase_tick)?
Wave stamps of BOTH kernels of the last tick (CSF_TRACE_BLOCKS: start / end of every pair wave; CSF_TRACE_AGENT: the per-agent
waves' eight stamps - entry, own loads, destination force, sums arrived, combine, integrate, stores issued, stores done), all on
wall_clock64 (100 MHz), printed relative to the first pair wave's start.    tools/chase_timeline.py [--agents N] [--ticks 40] [CSF_X=..]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def arg(name, dflt, cast=int):
    return cast(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else dflt


n, box, ticks = arg("--agents", 16384), arg("--box", 200.0, float), arg("--ticks", 40)
for kv in sys.argv[1:]:
    if kv.startswith("CSF_") and "=" in kv:
        os.environ[kv.split("=")[0]] = kv.split("=", 1)[1]
os.environ["CSF_TRACE_BLOCKS"] = "/tmp/chase_blocks.bin"
os.environ["CSF_TRACE_AGENT"] = "/tmp/chase_agent.bin"
import bench  # noqa: E402
from cyclistsocialforce_amd import parameters  # noqa: E402
from cyclistsocialforce_amd.engine import Engine  # noqa: E402

s0, off, dq = bench.synthetic_population(n, box)
e = Engine(parameters.default_pod("twod"), n)
e.add_agents(s0, 5.0)
e.set_dest_queue(np.arange(n), off, dq, reset=True)
e.step(ticks, sync=True)
print("side-by-side ticks", e.chase_ticks(), "of", ticks)
e.close()
blk = np.fromfile("/tmp/chase_blocks.bin", dtype=np.uint64).reshape(-1, 3)
blk = blk[blk[:, 0] != 0]
ag = np.fromfile("/tmp/chase_agent.bin", dtype=np.uint64)
ag = ag[: 8 * ((n + 63) // 64)].reshape(-1, 8).astype(np.int64)
t0 = int(blk[:, 0].min())
us = lambda x: (np.asarray(x, dtype=np.int64) - t0) / 100.0   # noqa: E731
pe = us(blk[:, 1])
print(f"pair waves: {len(blk)}; first start 0.0, last start {us(blk[:, 0]).max():.1f}, ends: 50 % {np.percentile(pe, 50):.1f}  85 % {np.percentile(pe, 85):.1f}  "
      f"95 % {np.percentile(pe, 95):.1f}  99 % {np.percentile(pe, 99):.1f}  last {pe.max():.1f} us")
names = ["entry", "own loads", "dest force", "sums arrived", "combine", "integrate", "stores issued", "stores done"]
print("per-agent waves (us after the first pair wave's start): min / median / max over the", ag.shape[0], "waves")
for k, nm in enumerate(names):
    col = us(ag[:, k])
    print(f"  {nm:14s} {col.min():8.1f} {np.median(col):8.1f} {col.max():8.1f}")
wait = (ag[:, 3] - ag[:, 2]) / 100.0
print(f"  waiting for the sums: median {np.median(wait):.1f}, max {wait.max():.1f} us;  last per-agent wave done {us(ag[:, 7]).max() - pe.max():+.1f} us after the last pair wave")
