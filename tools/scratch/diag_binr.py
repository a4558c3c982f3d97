import os, sys
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
from oracle import csf_oracle as orc
from cyclistsocialforce_amd import engine, parameters
from test_gpu_parity import synthetic_population
n, box = 65536, 400.0
x, y, psi, v, off, dq = synthetic_population(n, box)
s0 = np.c_[x, y, psi, v, np.zeros(n)]
def rep(env):
    for k, val in env.items(): os.environ[k] = val
    e = engine.Engine(parameters.default_pod("twod"), n)
    e.add_agents(s0, 1e6); e.set_dest_queue(np.arange(n), off, dq, reset=True)
    e.calc_forces()
    _, _, rx, ry = e.force_parts()
    nd = e.near_dropped()
    e.close()
    for k in env: del os.environ[k]
    return rx, ry, nd
bx, by, nd1 = rep({})
sx, sy, nd2 = rep({"CSF_RECV_BINNED": "0"})
ex, ey, nd3 = rep({"CSF_FAR_EPS": "0"})
print("near_dropped", nd1, nd2, nd3)
d = np.hypot(bx - ex, by - ey)
worst = np.argsort(d)[-40:]
recv = np.concatenate([worst, np.arange(0, n, 4099)])
ox, oy = orc.column_sums(orc.default_params("twod"), x, y, psi, v, recv)
scale = np.hypot(ox, oy).max()
for name, (ax, ay) in (("binned", (bx, by)), ("slot order", (sx, sy)), ("every pair", (ex, ey))):
    err = np.hypot(ax[recv] - ox, ay[recv] - oy)
    print(f"{name:12s} vs oracle: max {err.max():.3e} (rel {err.max() / scale:.2e}), at the 40 worst of binned-vs-every-pair {err[:40].max():.3e}, elsewhere {err[40:].max():.3e}")
print("scale", scale, "worst |F|", np.hypot(ox, oy)[:5])
