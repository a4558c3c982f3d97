import os, sys, time, json
import numpy as np
sys.path.insert(0, os.getcwd())
from bench import synthetic_population
from cyclistsocialforce_amd import parameters
from cyclistsocialforce_amd.engine import Engine
n, box, ticks, frac = 16384, 200.0, 600, float(sys.argv[1]) if len(sys.argv) > 1 else 0.05
s0, off, dq = synthetic_population(n, box)
pool, _, pdq = synthetic_population(8 * n, box, seed=1)
pdq = pdq.reshape(-1, 4, 3)
k = int(frac * n)
rng = np.random.default_rng(0)
kills = [np.sort(rng.choice(n, k, replace=False)).astype(np.int32) for _ in range(ticks)]
news = [(np.arange(k) + t * k) % (8 * n) for t in range(ticks)]
new_s = [np.ascontiguousarray(pool[i]) for i in news]
new_q = [np.ascontiguousarray(pdq[i].reshape(-1, 3)) for i in news]
tail = np.arange(n - k, n, dtype=np.int32)
qoff = np.arange(k + 1, dtype=np.int64) * 4
e = Engine(parameters.default_pod("twod"), n)
e.add_agents(s0, 5.0); e.set_dest_queue(np.arange(n), off, dq, reset=True); e.step(300, sync=True)
T = [0.0] * 4
t0 = time.perf_counter()
for t in range(ticks):
    a = time.perf_counter(); e.remove_agents(kills[t])
    b = time.perf_counter(); e.add_agents(new_s[t], 5.0)
    c = time.perf_counter(); e.set_dest_queue(tail, qoff, new_q[t], reset=True)
    d = time.perf_counter(); e.step(1)
    f = time.perf_counter()
    T[0] += b - a; T[1] += c - b; T[2] += d - c; T[3] += f - d
e.sync()
dt = time.perf_counter() - t0
print(json.dumps({"k": k, "us_per_tick": dt / ticks * 1e6, "remove": T[0] / ticks * 1e6, "add": T[1] / ticks * 1e6, "queue": T[2] / ticks * 1e6, "step_call": T[3] / ticks * 1e6}))
e.profile(1); 
for t in range(64):
    e.remove_agents(kills[t]); e.add_agents(new_s[t], 5.0); e.set_dest_queue(tail, qoff, new_q[t], reset=True); e.step(1)
e.sync()
print({k_: ms * 1e3 / max(c, 1) for k_, (ms, c) in e.profile_kernels().items()})
