"""How often does BASELINE config 3 (16 384 InvPendulumBicycle, 200 m x 200 m) meet the cut of vehicle.py:1832?  CPU only: the
oracle runs the population and counts, per tick, the riders whose force direction lies within `band` rad of +-pi (band = the
direction error of an fp32 pair sum, 1e-5 relative: a rider inside it may get psi_d = +pi - eps from one program and -pi + eps
from another) and the riders with a swing under way (|psi_d - unwrapped yaw| > pi).

usage: python tools/invpend_cut_count.py [n=16384] [box=200] [ticks=200]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import csf_oracle as orc  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
box = float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
ticks = int(sys.argv[3]) if len(sys.argv) > 3 else 200
s0, off, dq = bench.synthetic_population(n, box, seed=0)
s6 = np.zeros((n, 6)); s6[:, :5] = s0
pop = orc.Population(orc.default_params("invpend"), s6, 5.0, off, dq)
bands = (1e-3, 1e-4, 1e-5)
hits = {b: 0 for b in bands}
swing_ticks, swingers, walkers = 0, set(), 0
for t in range(ticks):
    pop.calc_forces_range(0, n)
    fx, fy = pop.forces()
    xl, zr = pop.lti()
    psid = np.arctan2(fy, fx)
    d = np.pi - np.abs(psid)
    riding = zr[:, 0]
    for b in bands:
        hits[b] += int(((d < b) & riding).sum())
    sw = (np.abs(psid - xl[:, 4]) > np.pi) & riding
    swing_ticks += int(sw.sum())
    swingers.update(np.nonzero(sw)[0].tolist())
    walkers += int((~riding).sum())
    pop.apply_forces(fx, fy)
    if t % 20 == 19:
        print(f"tick {t + 1}: rider-ticks within {bands} rad of the cut: {[hits[b] for b in bands]}; swing rider-ticks {swing_ticks} "
              f"({len(swingers)} riders); walking rider-ticks {walkers}", flush=True)
print(f"{n} InvPendulum riders, {box:g} m, {ticks} ticks = {n * ticks} rider-ticks")
for b in bands:
    print(f"  force direction within {b:g} rad of +-pi while riding: {hits[b]} rider-ticks")
print(f"  commanded yaw more than pi from the loop's unwrapped yaw: {swing_ticks} rider-ticks, {len(swingers)} riders")
