"""The per-agent launch BESIDE the pair launch that feeds it (include/csf.h: csf_chase_ticks; csf_engine.hip: enqueue_chase_tick): every
wave of the per-agent kernel waits for the arrival counter of its 64 road users while the rest of the pair launch drains, on the engine's
second stream.  The two orders must give the same states to the last bit (same kernels' arithmetic, same order of additions); the hand-over
of undecidable pairs (intersection.py:690-745 decided in fp64 by the per-agent kernel) and the double-buffered records are what could
break.  SocialForceIntersection.step: intersection.py:866-896."""
import numpy as np
import pytest

from oracle import csf_oracle as orc
from test_gpu_parity import amd  # noqa: F401  (fixture)

pytestmark = [pytest.mark.gpu]


def crowd(n, box, seed, ns):
    rng = np.random.default_rng(seed)
    s0 = np.zeros((n, ns))
    s0[:, 0] = rng.uniform(0, box, n); s0[:, 1] = rng.uniform(0, box, n)
    s0[:, 2] = rng.uniform(-np.pi, np.pi, n); s0[:, 3] = rng.uniform(3, 6, n)
    d = np.array([50.0, 99.0, 100.0])
    dq = np.zeros((n, 4, 3))
    dq[:, 0, 0] = s0[:, 0]; dq[:, 0, 1] = s0[:, 1]
    dq[:, 1:, 0] = s0[:, 0, None] + d[None, :] * np.cos(s0[:, 2])[:, None]
    dq[:, 1:, 1] = s0[:, 1, None] + d[None, :] * np.sin(s0[:, 2])[:, None]
    return s0, np.arange(n + 1) * 4, dq.reshape(-1, 3)


def engine(amd, monkeypatch, chase, model, s0, off, dq, rule=0, **kw):
    monkeypatch.setenv("CSF_CHASE", str(chase))
    e = amd.Engine(amd.pod(model, priority_rule=rule, **kw), s0.shape[0] + 256)
    e.add_agents(s0, 5.0)
    e.set_dest_queue(np.arange(s0.shape[0]), off, dq, reset=True)
    return e


def same(a, b):
    sa, pa, za, ta = a.state(with_nav=True)
    sb, pb, zb, tb = b.state(with_nav=True)
    assert ta == tb and np.array_equal(sa, sb) and np.array_equal(pa, pb) and np.array_equal(za, zb)
    assert np.array_equal(np.c_[a.forces()], np.c_[b.forces()]) and np.array_equal(np.c_[a.force_parts()], np.c_[b.force_parts()])
    xa, ya, ra = a.integrator_state(); xb, yb, rb = b.integrator_state()
    assert np.array_equal(xa, xb) and np.array_equal(ya, yb) and np.array_equal(ra, rb)
    assert (a.status() == 0).all() and (b.status() == 0).all() and b.near_dropped() == 0


@pytest.mark.parametrize("model,n,box,ticks,rule,hfov", [("twod", 16384, 200.0, 200, 0, None), ("invpend", 8192, 140.0, 150, 0, None),
                                                         ("planarpoint", 8192, 140.0, 150, 1, None), ("twod", 4096, 45.0, 150, 0, 4.0)])
def test_side_by_side_ticks_are_the_ticks_in_turn_to_the_last_bit(amd, monkeypatch, model, n, box, ticks, rule, hfov):
    """... across re-binnings (ticks 64, 128, 192: those ticks take the launches in turn and the halves are made equal again), at the
    headline size, with the other rider classes, priority to the right, and in a crowd dense enough for near pairs and hand-overs"""
    ns = orc.N_STATES[orc.MODEL_IDS[model]]
    s0, off, dq = crowd(n, box, 11, ns)
    kw = {} if hfov is None else {"hfov": hfov}
    a = engine(amd, monkeypatch, 0, model, s0, off, dq, rule, **kw)
    b = engine(amd, monkeypatch, 2, model, s0, off, dq, rule, **kw)
    for k in range(0, ticks, 50):
        a.step(50); b.step(50)
        same(a, b)
    assert a.chase_ticks() == 0 and b.chase_ticks() >= ticks - 8, b.chase_ticks()
    assert b.count_pairs()[1] == "pair_cull_kernel"
    a.close(); b.close()


def test_side_by_side_ticks_between_other_calls(amd, monkeypatch):
    """calc_forces / apply_forces, pushed states, per-kernel profiling, per-tick calls (which take the launches in turn), departures and
    arrivals between stretches of side-by-side ticks: every hand-over between the paths leaves both engines in the same state, and the
    forces stay the oracle's"""
    n, box = 6000, 120.0
    s0, off, dq = crowd(n, box, 5, 5)
    a = engine(amd, monkeypatch, 0, "twod", s0, off, dq)
    b = engine(amd, monkeypatch, 2, "twod", s0, off, dq)
    for e in (a, b):
        e.step(20)
        fx, fy = e.calc_forces()
        e.apply_forces(fx, fy)
        e.step(10)
        s = e.state(); s[::7, 3] *= 0.5
        e.push_state(np.arange(n), s)
        e.step(12)
        e.profile(2); e.step(9); e.profile_kernels(); e.profile(0)
        for _ in range(3):
            e.step(1)
        e.step(2); e.step(30)
    same(a, b)
    assert b.chase_ticks() >= 20 + 10 + 12 + 9 + 30 - 12
    kill = np.arange(3, n, 37, dtype=np.int32)
    new, noff, ndq = crowd(150, box, 9, 5)
    for e in (a, b):
        e.remove_agents(kill)
        e.add_agents(new, 5.0)
        m = e.n
        e.set_dest_queue(np.arange(m - 150, m), noff, ndq, reset=True)
        e.step(40)
    same(a, b)
    st = b.state()
    b.calc_forces()
    fdx, fdy, frx, fry = b.force_parts()
    recv = np.arange(0, st.shape[0], 5)
    ox, oy = orc.column_sums(orc.default_params("twod"), st[:, 0], st[:, 1], st[:, 2], st[:, 3], recv)
    lim, r = np.hypot(fdx[recv], fdy[recv]), np.maximum(np.hypot(ox, oy), 1e-300)
    sc = np.minimum(1.0, lim / r)
    assert max(np.abs(frx[recv] - ox * sc).max(), np.abs(fry[recv] - oy * sc).max()) < 1e-4 * max(np.hypot(ox * sc, oy * sc).max(), 1.0)
    a.close(); b.close()


def test_the_engine_measures_which_order_is_faster(amd, monkeypatch):
    """CSF_CHASE=1 (the default): three periods between re-binnings - in turn, side by side, in turn -, then the faster order - the states are the in-turn engine's
    whatever it decides"""
    n, box = 16384, 200.0
    s0, off, dq = crowd(n, box, 2, 5)
    a = engine(amd, monkeypatch, 0, "twod", s0, off, dq)
    b = engine(amd, monkeypatch, 1, "twod", s0, off, dq)
    for e in (a, b):
        e.step(10); e.step(250); e.step(400)
    same(a, b)
    decided, us = b.chase_calibration()
    assert decided in (1, -1), decided
    if us[0] > 0:                                   # (0: an earlier engine of this kind in this process has measured, and this one took that over)
        assert 50.0 < us[0] < 1000.0 and 50.0 < us[1] < 1000.0 and (decided == -1 or us[1] < 0.99 * us[0]), (decided, us)
    a.step(100); b.step(100)
    same(a, b)
    assert (b.chase_ticks() >= 90) == (decided == 1), (b.chase_ticks(), decided)
    print(f"in turn {us[0]:.1f} us per tick, side by side {us[1]:.1f}: {'side by side' if decided == 1 else 'in turn'}")
    # what one engine found serves the next of its kind: no measurement, the same decision from the first eligible tick on
    c = engine(amd, monkeypatch, 1, "twod", s0, off, dq)
    c.step(40)
    assert c.chase_calibration() == (decided, [0.0, 0.0]) and (c.chase_ticks() >= 30) == (decided == 1)
    a.close(); b.close(); c.close()
