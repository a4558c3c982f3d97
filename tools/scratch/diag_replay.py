import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from cyclistsocialforce_amd import engine, parameters
import bench
n = int(sys.argv[1]); T = int(sys.argv[2]); fix = int(sys.argv[3])
s0, off, dq = bench.synthetic_population(n, 1500.0, seed=34, reach=(50.0, 99.0, 100.0))
e = engine.Engine(parameters.default_pod("twod"), n)
e.add_agents(s0[:, :5], 5.0)
e.set_dest_queue(np.arange(n), off, dq, reset=True)
e.step(2)
st = e.state()
Fx = (5.0 * np.cos(st[:, 2]))[None, :].repeat(T, 0)
Fy = (5.0 * np.sin(st[:, 2]))[None, :].repeat(T, 0)
print("replay", n, T, fix, flush=True)
e.replay_forces(Fx, Fy, fix_speed=bool(fix), return_states=False)
print("ok", np.isfinite(e.state()).all(), flush=True)
