import sys, time, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from tests.test_gpu_small import crowd
from cyclistsocialforce_amd import parameters
from cyclistsocialforce_amd.engine import Engine
from cyclistsocialforce_amd.intersection import SocialForceIntersection
from cyclistsocialforce_amd.vehicle import TwoDBicycle
n = 3
x, y, psi, v, off, dq = crowd(n, seed=3)
def mk():
    e = Engine(parameters.default_pod("twod"), n)
    e.add_agents(np.c_[x, y, psi, v, np.zeros(n)], 5.0); e.set_dest_queue(np.arange(n), off, dq, reset=True); e.step(50, sync=True)
    return e
K = 5000
e = mk(); t0 = time.perf_counter()
for _ in range(K): e.step(1); e.tick_snapshot()
print("step(1) + tick_snapshot(): %.2f us" % ((time.perf_counter() - t0) / K * 1e6))
e = mk(); t0 = time.perf_counter()
for _ in range(K): e.step_snapshot(1)
print("step_snapshot(1):          %.2f us" % ((time.perf_counter() - t0) / K * 1e6))
e = mk(); t0 = time.perf_counter()
for _ in range(K): e.step(1); e.sync()
print("step(1) + sync():          %.2f us" % ((time.perf_counter() - t0) / K * 1e6))
vs = []
D = dq.reshape(n, 5, 3)
for k in range(n):
    b = TwoDBicycle((x[k], y[k], psi[k], v[k], 0.0), id=str(k)); b.setDestinations(D[k, 1:, 0], D[k, 1:, 1]); vs.append(b)
ins = SocialForceIntersection(vs, capacity=n)
ins.step(); t0 = time.perf_counter()
for _ in range(K): ins.step()
print("SocialForceIntersection.step(): %.2f us" % ((time.perf_counter() - t0) / K * 1e6))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(2000): ins.step()
pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(12)
