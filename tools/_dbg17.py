import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
os.environ["CSF_PAIR_VARIANT"] = "0"
import numpy as np
import test_gpu_large as T
from cyclistsocialforce_amd import engine, parameters
seed, n0, box = 17, 3300, 700.0
rng = np.random.default_rng(seed)
cap = n0 + 900
pool, _, pdq = T.population(cap + 6000, box, seed=seed + 50)
pdq = pdq.reshape(-1, 4, 3)
pods = [parameters.default_pod("twod")]
cls_of = rng.integers(0, 1, pool.shape[0])
engines = []
for inc in (True, False):
    e = engine.Engine(pods[0], cap)
    e.set_incremental(inc)
    e.add_agents(pool[:n0], 5.0)
    e.set_dest_queue(np.arange(n0), np.arange(n0 + 1) * 4, pdq[:n0].reshape(-1, 3), reset=True)
    e.step(2)
    engines.append(e)
fresh, n = n0, n0
# replicate: extend, add, step, step  (same rng draws as the test)
ops = []
for it in range(4):
    op = rng.choice(["step", "step", "remove", "add", "add", "replace", "edit", "extend", "vdes", "push"],
                    p=[0.22, 0.1, 0.14, 0.12, 0.08, 0.1, 0.08, 0.08, 0.05, 0.03])
    ops.append(str(op))
    if op == "step":
        k = int(rng.integers(1, 4))
        for j in range(k):
            for e in engines: e.step(1)
            w = 3058
            fa = engines[0].force_parts(); fb = engines[1].force_parts()
            A = engines[0].state(); B = engines[1].state()
            print("   tick", j, "agent", w, "dest A", fa[0][w], fa[1][w], "B", fb[0][w], fb[1][w], "rep A", fa[2][w], fa[3][w], "B", fb[2][w], fb[3][w], "dv", A[w,3]-B[w,3], "ddelta", A[w,4]-B[w,4], "dev all", np.abs(A-B).max())
        print("step", k)
    elif op == "add":
        k = int(rng.integers(1, 60)); new = np.arange(fresh, fresh + k); fresh += k
        for e in engines:
            e.add_agents(pool[new], 5.0)
            e.set_dest_queue(np.arange(n, n + k), np.arange(k + 1) * 4, pdq[new].reshape(-1, 3), reset=True)
        n += k; print("add", k)
    elif op == "extend":
        k = int(rng.integers(1, 30)); idx = np.sort(rng.choice(n, k, replace=False)); src = rng.integers(0, pdq.shape[0], k)
        rows, off, mode = pdq[src][:, 2:].reshape(-1, 3), np.arange(k + 1) * 2, 0
        for e in engines: e.set_dest_queue(idx, off, rows, reset=mode)
        print("extend", idx)
    A, pa, za, _ = engines[0].state(with_nav=True); B, pb, zb, _ = engines[1].state(with_nav=True)
    d = np.abs(A - B).max(axis=1); w = int(d.argmax())
    print(it, op, "max dev", d.max(), "at", w, "ptr", pa[w], pb[w], "zn", za[w], zb[w])
    fa = engines[0].force_parts(); fb = engines[1].force_parts()
    print("   dest A", fa[0][w], fa[1][w], "B", fb[0][w], fb[1][w], " rep A", fa[2][w], fa[3][w], "B", fb[2][w], fb[3][w])
    print("   state A", A[w], "B", B[w])
