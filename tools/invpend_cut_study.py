"""Where does an InvPendulum crowd leave its own shadow?  (CPU only: the oracle against itself.)

vehicle.py:1832 feeds psi_d = arctan2(Fy, Fx) in (-pi, pi] to a yaw loop whose state vehicle.x[4] is the UNWRAPPED yaw
(:1835-1846): a rider whose force direction crosses the cut at +-pi sees its commanded yaw jump by 2 pi, and the loop answers
with a full swing.  Which tick the jump falls on hangs on the sign of Fy where Fy ~ 0, Fx < 0.  This script

  1. runs the oracle free and lists, per tick, the riders whose force direction is within `band` rad of the cut, and the
     riders whose commanded yaw is more than pi away from the unwrapped yaw of the loop (a swing in progress);
  2. runs a second oracle whose forces are perturbed by `eps` relative (what an fp32 pair sum does) in the shadow windows
     of tests/conftest.py: shadow_run and reports in which window, for which rider, the two part.

usage: python tools/invpend_cut_study.py [n=32] [seed=320] [box=30] [hfov=4.0] [ticks=400] [window=10] [eps=1e-5]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import csf_oracle as orc  # noqa: E402


def crowd(n, seed, box):
    rng = np.random.default_rng(seed)
    x, y = rng.uniform(0, box, n), rng.uniform(0, box, n)
    psi, v = rng.uniform(-np.pi, np.pi, n), rng.uniform(3, 6, n)
    reach = np.array([8.0, 25.0, 60.0, 61.0])
    dq = np.zeros((n, 5, 3))
    dq[:, 0, 0], dq[:, 0, 1] = x, y
    dq[:, 1:, 0] = x[:, None] + reach[None, :] * np.cos(psi)[:, None]
    dq[:, 1:, 1] = y[:, None] + reach[None, :] * np.sin(psi)[:, None]
    dq[:, 4, 2] = 1.0
    return x, y, psi, v, np.arange(n + 1) * 5, dq.reshape(-1, 3)


def main():
    a = sys.argv[1:]
    n = int(a[0]) if len(a) > 0 else 32
    seed = int(a[1]) if len(a) > 1 else 320
    box = float(a[2]) if len(a) > 2 else 30.0
    hfov = float(a[3]) if len(a) > 3 else 4.0
    ticks = int(a[4]) if len(a) > 4 else 400
    window = int(a[5]) if len(a) > 5 else 10
    eps = float(a[6]) if len(a) > 6 else 1e-5
    x, y, psi, v, off, dq = crowd(n, seed, box)
    s0 = np.zeros((n, 6)); s0[:, 0] = x; s0[:, 1] = y; s0[:, 2] = psi; s0[:, 3] = v
    P = orc.default_params("invpend", hfov=hfov)

    # 1. the free run
    pop = orc.Population(P, s0, 5.0, off, dq)
    near_cut, swings = [], {}
    for t in range(ticks):
        pop.calc_forces_range(0, n)
        fx, fy = pop.forces()
        xl, zr = pop.lti()
        psid = np.arctan2(fy, fx)
        d = np.pi - np.abs(psid)
        for r in np.nonzero((d < 1e-3) & zr[:, 0])[0]:
            near_cut.append((t, int(r), fx[r], fy[r], d[r]))
        for r in np.nonzero((np.abs(psid - xl[:, 4]) > np.pi) & zr[:, 0])[0]:
            swings.setdefault(int(r), []).append(t)
        pop.apply_forces(fx, fy)
    print(f"free run, {n} InvPendulum riders, seed {seed}, box {box} m, hfov {hfov}, {ticks} ticks")
    print(f"  rider-ticks with the force direction within 1e-3 rad of the cut: {len(near_cut)}")
    for t, r, fx_, fy_, d_ in near_cut[:12]:
        print(f"    tick {t:4d} rider {r:3d}  F = ({fx_:+.6e}, {fy_:+.6e})  pi - |psi_d| = {d_:.3e}")
    print(f"  riders whose commanded yaw is > pi from the loop's unwrapped yaw (a swing under way), first / last tick / count:")
    for r, ts in sorted(swings.items()):
        print(f"    rider {r:3d}: ticks {ts[0]} .. {ts[-1]} ({len(ts)})")

    # 2. the shadow: B follows A (forces perturbed by eps) re-anchored every `window` ticks the way shadow_run does it
    rng = np.random.default_rng(1)
    for complete in (False, True):
        A = orc.Population(P, s0, 5.0, off, dq)
        B = orc.Population(P, s0, 5.0, off, dq)

        def stepA(k):
            for _ in range(k):
                A.calc_forces_range(0, n)
                fx, fy = A.forces()
                sc = np.hypot(fx, fy).max()
                A.apply_forces(fx + eps * sc * rng.standard_normal(n), fy + eps * sc * rng.standard_normal(n))

        def anchor():
            s = A.state(); ptr, zn, i, _ = A.nav()
            B.push_state(s, ptr, zn.astype(np.uint8), col=i)
            if complete:
                B.set_lti(*A.lti())

        tick, first = 0, None
        rows = []
        while tick < ticks:
            end = min(tick + window, ticks)
            if tick > 0:
                anchor(); stepA(1); B.step(1); anchor()
            k = end - tick - (0 if end == ticks else 1)
            stepA(k); B.step(k)
            sa, sb = A.state(), B.state()
            dev = np.hypot(sa[:, 0] - sb[:, 0], sa[:, 1] - sb[:, 1])
            xa, _ = A.lti(); xb, _ = B.lti()
            rows.append((end, dev.max(), int(dev.argmax()), np.abs(xa[:, 4] - xb[:, 4]).max(), np.abs(xa[:, [1, 3]] - xb[:, [1, 3]]).max()))
            if first is None and dev.max() > 1e-4 * box:
                first = rows[-1]
            tick = end
        print(f"shadow of the perturbed run (eps = {eps:g}), windows of {window}, anchor = s, ptr, znav, column"
              + (" + vehicle.x, zrid" if complete else " (shadow_run's)") + ":")
        print(f"  worst deviation at a window end: {max(r[1] for r in rows):.3e} m;  first window beyond 1e-4 of the box: {first}")
        bad = [r for r in rows if r[1] > 1e-4 * box]
        for r in bad[:6]:
            print(f"    window ending {r[0]:4d}: {r[1]:.3e} m (rider {r[2]}), |d psi_unwrapped| max {r[3]:.3e}, |d rates| max {r[4]:.3e}")


if __name__ == "__main__":
    main()
