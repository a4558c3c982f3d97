import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from bench import synthetic_population
from cyclistsocialforce_amd import parameters
from cyclistsocialforce_amd.engine import Engine
n = int(sys.argv[1]); box = float(sys.argv[2]); K = int(sys.argv[3])
s0, off, dq = synthetic_population(n, box, reach=tuple(50.0 * k for k in range(1, 14)))
e = Engine(parameters.default_pod("twod"), n)
e.add_agents(s0, 5.0)
e.set_dest_queue(np.arange(n), off, dq, reset=True)
e.step(100, sync=True)
t0 = time.perf_counter(); e.step(K, sync=True); dt = time.perf_counter() - t0
print(n, "us/tick", dt / K * 1e6, "mid", e.mid_ticks())
