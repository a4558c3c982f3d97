#!/usr/bin/env python3
"""Host time to ENQUEUE a tick (csf_step returns before the GPU has run it) against the GPU time of the tick: the
engine stays ahead of the device only while the first is smaller.  Run with CSF_BENCH_FORCE_DIST-style rehearsal by
passing `dist` (1-rank communicator) and CSF_FAKE_SHARD=0/w for the per-rank load of a w-way shard."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cyclistsocialforce_amd import parameters  # noqa: E402
from cyclistsocialforce_amd.engine import Engine  # noqa: E402

n = 16384
s0, off, dq = bench.synthetic_population(n, 200.0)
eng = Engine(parameters.default_pod("twod"), n)
eng.add_agents(s0, 5.0)
eng.set_dest_queue(np.arange(n), off, dq, reset=True)
if len(sys.argv) > 1 and sys.argv[1] == "dist":
    eng.comm_init(Engine.comm_unique_id(), 0, 1)
eng.step(50, sync=True)
K = 2000
t0 = time.perf_counter()
eng.step(K)
t1 = time.perf_counter()
eng.sync()
t2 = time.perf_counter()
print(f"enqueue {1e6 * (t1 - t0) / K:.1f} us per tick, GPU {1e6 * (t2 - t0) / K:.1f} us per tick "
      f"({'sharded path' if len(sys.argv) > 1 else 'single GPU'}, CSF_FAKE_SHARD={os.environ.get('CSF_FAKE_SHARD', '-')})")
