"""Host time of the three population calls and of csf_step per tick at 5 % churn (N = 16 384; another fraction as the first argument), then the kernels' own times over 100 such ticks (GPU box)."""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
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
e.add_agents(s0, 5.0)
e.set_dest_queue(np.arange(n), off, dq, reset=True)
e.step(300, sync=True)
T = dict(remove=0.0, add=0.0, queue=0.0, step=0.0)
t00 = time.perf_counter()
for t in range(ticks):
    a = time.perf_counter(); e.remove_agents(kills[t])
    b = time.perf_counter(); e.add_agents(new_s[t], 5.0)
    c = time.perf_counter(); e.set_dest_queue(tail, qoff, new_q[t], reset=True)
    d = time.perf_counter(); e.step(1)
    f = time.perf_counter()
    T["remove"] += b - a; T["add"] += c - b; T["queue"] += d - c; T["step"] += f - d
e.sync()
tot = time.perf_counter() - t00
print(os.environ.get("CSF_HOLE_REUSE", "1"), {k2: round(v / ticks * 1e6, 1) for k2, v in T.items()}, "total us/tick", round(tot / ticks * 1e6, 1), flush=True)
# device-side: the same ticks with profiling of kernels
e.profile(1); e.profile_kernels()
for t in range(100):
    e.remove_agents(kills[t]); e.add_agents(new_s[t], 5.0); e.set_dest_queue(tail, qoff, new_q[t], reset=True); e.step(1)
e.sync()
kk = e.profile_kernels()
print("   kernels us/launch:", {a: round(v[0] / max(v[1], 1) * 1e3, 1) for a, v in kk.items() if v[1]})
