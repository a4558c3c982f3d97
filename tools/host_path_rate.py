#!/usr/bin/env python3
"""Tick rate when every tick crosses the host boundary (PCIe-inclusive): the drop-in `SocialForceIntersection.step()`
path with its Python mirror, and the array-level `Engine.step(1)` + `state()` read-back, next to the resident
`Engine.step(K)` figure that bench.py reports."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synthetic_population  # noqa: E402
from cyclistsocialforce_amd import parameters  # noqa: E402
from cyclistsocialforce_amd.engine import Engine  # noqa: E402
from cyclistsocialforce_amd.intersection import SocialForceIntersection  # noqa: E402
from cyclistsocialforce_amd.vehicle import TwoDBicycle  # noqa: E402


def mirror_path(n, box, ticks):
    s0, off, dq = synthetic_population(n, box)
    vs = []
    for k in range(n):
        v = TwoDBicycle(tuple(s0[k]), id=str(k))
        v.setDestinations(dq[4 * k + 1:4 * k + 4, 0], dq[4 * k + 1:4 * k + 4, 1])
        vs.append(v)
    ins = SocialForceIntersection(vs, capacity=n)
    ins.step()
    t0 = time.perf_counter()
    for _ in range(ticks):
        ins.step()
    dt = time.perf_counter() - t0
    print(f"SocialForceIntersection.step() mirror path  N={n:6d}: {dt / ticks * 1e3:8.3f} ms/tick  {n * ticks / dt:12.0f} agent-steps/s")


def array_path(n, box, ticks):
    s0, off, dq = synthetic_population(n, box)
    e = Engine(parameters.default_pod("twod"), n)
    e.add_agents(s0, 5.0)
    e.set_dest_queue(np.arange(n), off, dq, reset=True)
    e.step(5, sync=True)
    t0 = time.perf_counter()
    for _ in range(ticks):
        e.step(1)
        e.state_by_component()
    dt = time.perf_counter() - t0
    print(f"Engine.step(1) + csf_get_state read-back    N={n:6d}: {dt / ticks * 1e3:8.3f} ms/tick  {n * ticks / dt:12.0f} agent-steps/s")
    t0 = time.perf_counter()
    for _ in range(ticks):
        e.step(1)
        e.tick_snapshot()
    dt = time.perf_counter() - t0
    print(f"Engine.step(1) + tick_snapshot() (1 transfer) N={n:6d}: {dt / ticks * 1e3:8.3f} ms/tick  {n * ticks / dt:12.0f} agent-steps/s")
    t0 = time.perf_counter()
    e.step(ticks, sync=True)
    dt = time.perf_counter() - t0
    print(f"Engine.step(K), state resident              N={n:6d}: {dt / ticks * 1e3:8.3f} ms/tick  {n * ticks / dt:12.0f} agent-steps/s")


if __name__ == "__main__":
    mirror_path(3, 30.0, 700)
    mirror_path(1024, 200.0, 100)
    mirror_path(16384, 200.0, 50)
    array_path(1024, 200.0, 500)
    array_path(16384, 200.0, 300)
