import os, sys
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
from oracle import csf_oracle as orc
from cyclistsocialforce_amd import engine, parameters
import bench

def clamped(ox, oy, fdx, fdy):
    lim, r = np.hypot(fdx, fdy), np.maximum(np.hypot(ox, oy), 1e-300)
    sc = np.minimum(1.0, lim / r)
    return ox * sc, oy * sc

def run(tag):
    n, box, grow = 24576, 1500.0, int(os.environ.get("GROW", "700"))
    s0, off, dq = bench.synthetic_population(n + 2048, box, seed=33, reach=(50.0, 99.0, 100.0))
    dq3 = dq.reshape(-1, 4, 3)
    p = orc.default_params("twod")
    e = engine.Engine(parameters.default_pod("twod"), n + grow)
    e.set_incremental(True)
    e.add_agents(s0[:n, :5], 5.0)
    e.set_dest_queue(np.arange(n), np.arange(n + 1) * 4, dq3[:n].reshape(-1, 3), reset=True)
    e.step(3)
    rng = np.random.default_rng(9)
    k = 300
    kill = np.sort(rng.choice(n, k, replace=False))
    e.remove_agents(kill)
    new = np.arange(n, n + k + grow)
    e.add_agents(s0[new, :5], 5.0)
    e.set_dest_queue(np.arange(n - k, n + grow), np.arange(k + grow + 1) * 4, dq3[new].reshape(-1, 3), reset=True)
    for tick in range(2):
        e.step(1)
        e.calc_forces()
        fdx, fdy, frx, fry = e.force_parts()
        st = e.state()
        for name, recv in (("old", np.arange(0, n - k, 307)), ("tail", np.arange(n - k, n, 5)), ("fresh", np.arange(n, n + grow, 5))):
            if recv.size == 0: continue
            ox, oy = orc.column_sums(p, st[:, 0], st[:, 1], st[:, 2], st[:, 3], recv)
            cx, cy = clamped(ox, oy, fdx[recv], fdy[recv])
            d = np.maximum(np.abs(frx[recv] - cx), np.abs(fry[recv] - cy)) / max(np.hypot(cx, cy).max(), 1.0)
            print(f"{tag} tick {tick} {name}: max {d.max():.2e}, wrong {int((d > 1e-4).sum())}/{recv.size}", flush=True)
    e.close()

run(os.environ.get("TAG", "default"))
