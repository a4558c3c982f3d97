"""The tick of mid-size populations as ONE launch (csf_mid.hip; include/csf.h: csf_mid_ticks): between 33 and ~3 000 road users
of one parameter set on one device, the pair sums of a receiver group and the per-agent tick of its road users share a grid,
and the next tick's records go to the other half of a double buffer.  Against the oracle (forces every tick, trajectories in
shadow windows), against the two-launch path, across re-binnings, population changes, a road, every rider model and both
priority rules.  SocialForceIntersection.step: intersection.py:866-896."""
import numpy as np
import pytest

from oracle import csf_oracle as orc
from conftest import shadow_run
from test_gpu_parity import MODELS, amd, make_engine  # noqa: F401  (amd: fixture)
from test_gpu_small import crowd

pytestmark = [pytest.mark.gpu, pytest.mark.auto_variant]


def anchor(e, pop):
    """the oracle takes the engine's state (two free runs are 1e-7 m apart after a few ticks, and with hundreds of road users
    some pair then sits on the other side of a field-of-view edge: the programs would be evaluating different populations)"""
    s, ptr, zn, tk = e.state(with_nav=True)
    pop.push_state(s, ptr, zn, col=tk % 3000)
    x, _, zrid = e.integrator_state()
    pop.set_lti(x, zrid)


def start_state(model, x, y, psi, v):
    n = len(x)
    s0 = np.zeros((n, orc.N_STATES[orc.MODEL_IDS[model]])); s0[:, 0] = x; s0[:, 1] = y; s0[:, 2] = psi; s0[:, 3] = v
    return s0


@pytest.mark.parametrize("model,n,rule,hfov,box", [("twod", 33, 0, None, 30.0), ("twod", 100, 1, None, 40.0), ("twod", 257, 0, 4.0, 50.0),
                                                   ("invpend", 64, 0, None, 40.0), ("invpend", 200, 1, 2.5, 60.0),
                                                   ("planarpoint", 90, 0, None, 40.0), ("planarpoint", 500, 1, 2.0, 70.0),
                                                   ("bicycle", 48, 0, None, 35.0), ("bicycle", 300, 1, 4.0, 60.0),
                                                   ("planarbike", 70, 0, None, 40.0),
                                                   ("twod", 1024, 0, None, 90.0), ("twod", 1279, 1, None, 100.0), ("twod", 2100, 0, None, 130.0)])
def test_mid_crowds_vs_oracle(amd, model, n, rule, hfov, box):
    """dense crowds: total forces against the oracle every tick for 20 ticks (every receiver, 1e-4 of the largest force; the
    oracle re-anchored on the engine's state before every tick where the crowd is large - with 3 000 road users some pair sits
    within the 1e-7 m that two free runs differ by of a field-of-view edge on most ticks, and the two programs then evaluate
    different populations); trajectories, destination pointers and navigation states over 140 ticks across two re-binnings
    (shadow windows of 10)"""
    x, y, psi, v, off, dq = crowd(n, seed=7 * n + rule, box=box)
    s0 = start_state(model, x, y, psi, v)
    over = {} if hfov is None else {"hfov": hfov}
    e = make_engine(amd, model, s0, 5.0, off, dq, rule, **over)
    pop = orc.Population(orc.default_params(model, priority_rule=rule, **over), s0, 5.0, off, dq)
    for t in range(20):
        if n > 300 and t > 0:
            anchor(e, pop)
        e.step(1); pop.step(1)
        fx, fy = e.forces(); ofx, ofy = pop.forces()
        scale = max(np.hypot(ofx, ofy).max(), 1e-3)
        assert max(np.abs(fx - ofx).max(), np.abs(fy - ofy).max()) < 1e-4 * scale, (t, n)
    assert e.mid_ticks() == 20 and e.small_ticks() == 0
    e2 = make_engine(amd, model, s0, 5.0, off, dq, rule, **over)
    pop2 = orc.Population(orc.default_params(model, priority_rule=rule, **over), s0, 5.0, off, dq)
    ticks = 140 if n <= 1024 else 70
    worst, _, got, ref = shadow_run(e2, pop2, ticks, 10)
    assert e2.mid_ticks() == ticks and (e2.status() == 0).all() and e2.near_dropped() == 0
    extent = max(np.ptp(ref[:, 0]), np.ptp(ref[:, 1]), 14.0)
    assert worst < 1e-4 * extent
    _, ptr, zn, _ = e2.state(with_nav=True)
    optr, ozn, _, _ = pop2.nav()
    np.testing.assert_array_equal(ptr, optr)
    np.testing.assert_array_equal(np.asarray(zn).reshape(n, 3).astype(bool), ozn)


@pytest.mark.parametrize("group", ["0", "4", "16", "32"])
def test_one_launch_agrees_with_two(amd, monkeypatch, group):
    """the same crowd through csf_mid.hip (road users per workgroup: the engine's choice, 4, 16, 32) and through pair launch +
    per-agent launch: the same per-pair code and per-agent code, the fp32 partial sums cut at other places - forces of the
    first tick to 1e-5 of the largest, the states after 60 ticks (one re-binning) to 1e-5 m"""
    x, y, psi, v, off, dq = crowd(700, seed=11, box=80.0)
    s0 = start_state("twod", x, y, psi, v)
    monkeypatch.setenv("CSF_MID_GROUP", group)
    a = make_engine(amd, "twod", s0, 5.0, off, dq)
    monkeypatch.setenv("CSF_FUSED_MID", "0")
    b = make_engine(amd, "twod", s0, 5.0, off, dq)
    monkeypatch.delenv("CSF_FUSED_MID")
    a.step(1); b.step(1)
    fa, fb = np.c_[a.forces()], np.c_[b.forces()]
    assert np.abs(fa - fb).max() < 1e-5 * np.abs(fb).max()
    a.step(59); b.step(59)
    assert np.abs(a.state()[:, :2] - b.state()[:, :2]).max() < 1e-5
    assert a.mid_ticks() == 60 and b.mid_ticks() == 0


def test_one_launch_and_other_paths_in_turn(amd):
    """ticks of the fused path between csf_calc_forces / csf_apply_forces, a state pushed from the host, per-kernel profiling
    (which takes the two launches) and population changes: every hand-over between the paths re-synchronises the double
    buffers; the oracle follows the same sequence"""
    n = 400
    x, y, psi, v, off, dq = crowd(n, seed=5, box=60.0)
    s0 = start_state("twod", x, y, psi, v)
    e = make_engine(amd, "twod", s0, 5.0, off, dq, capacity=n + 64)
    pop = orc.Population(orc.default_params("twod"), s0, 5.0, off, dq)

    def check(tag):
        fx, fy = e.forces(); ofx, ofy = pop.forces()
        scale = max(np.hypot(ofx, ofy).max(), 1e-3)
        assert max(np.abs(fx - ofx).max(), np.abs(fy - ofy).max()) < 1e-4 * scale, tag
        assert np.abs(e.state()[:, :2] - pop.state()[:, :2]).max() < 1e-4 * 60.0, tag

    e.step(3); pop.step(3); check("fused")
    fx, fy = e.calc_forces(); pop.calc_forces_range(0, n)
    e.apply_forces(fx, fy); pop.apply_forces(*pop.forces()); check("calc + apply")
    e.step(2); pop.step(2); check("fused again")
    e.profile(1); e.step(2); pop.step(2); e.profile(0); check("profiled (two launches)")
    assert e.profile_kernels()["pair"][1] == 2
    e.step(2); pop.step(2); check("fused after profiling")
    s = e.state(); s[::7, 3] *= 0.5
    _, ptr, zn, t = e.state(with_nav=True)
    e.push_state(np.arange(n), s); pop.push_state(s)
    e.step(2); pop.step(2); check("after a pushed state")
    assert e.mid_ticks() == 3 + 2 + 2 + 2 and (e.status() == 0).all()
    # population changes: the engine against itself through the host mirror (CSF_FUSED_MID has no say there)
    kill = np.arange(5, n, 41, dtype=np.int32)
    e.remove_agents(kill)
    new = np.c_[np.linspace(2, 58, 20), np.full(20, 61.0), np.full(20, -np.pi / 2), np.full(20, 4.0), np.zeros(20)]
    e.add_agents(new, 5.0)
    m = e.n
    rows = np.zeros((20, 5, 3))                                 # (the start row + four points straight ahead, as crowd() lays them out)
    rows[:, :, 0] = new[:, 0:1]
    rows[:, :, 1] = 61.0 - np.array([0.0, 8.0, 25.0, 60.0, 61.0])[None, :]
    e.set_dest_queue(np.arange(m - 20, m), np.arange(21) * 5, rows.reshape(-1, 3), reset=True)
    e.step(4)
    assert e.n == n - kill.size + 20 and np.isfinite(e.state()).all() and (e.status() == 0).all()
    st = e.state()
    e.calc_forces()
    fdx, fdy, frx, fry = e.force_parts()
    ox, oy = orc.column_sums(orc.default_params("twod"), st[:, 0], st[:, 1], st[:, 2], st[:, 3], np.arange(m))
    lim, r = np.hypot(fdx, fdy), np.maximum(np.hypot(ox, oy), 1e-300)
    sc = np.minimum(1.0, lim / r)
    assert max(np.abs(frx - ox * sc).max(), np.abs(fry - oy * sc).max()) < 1e-4 * max(np.hypot(ox * sc, oy * sc).max(), 1.0)


def test_one_launch_with_road_edges(amd):
    """intersection.py:226-242 beside the fused tick: the road launch in front of it, the road term in its combine phase"""
    n = 150
    x, y, psi, v, off, dq = crowd(n, seed=23, box=40.0)
    s0 = start_state("twod", x, y, psi, v)
    xs = np.linspace(-20.0, 70.0, 900)
    verts = np.r_[np.c_[xs, np.full(900, -3.0)], np.c_[xs, np.full(900, 43.0)]]
    roff, F0, sg = np.array([0, 900, 1800]), np.array([0.15, 0.2]), np.array([2.0, 2.0])
    e = make_engine(amd, "twod", s0, 5.0, off, dq)
    e.set_road(roff, verts, F0, sg)
    pop = orc.Population(orc.default_params("twod"), s0, 5.0, off, dq)
    pop.set_road(roff, verts, F0, sg)
    for t in range(40):
        if t > 0:
            anchor(e, pop)
        e.step(1); pop.step(1)
        fx, fy = e.forces(); ofx, ofy = pop.forces()
        scale = max(np.hypot(ofx, ofy).max(), 1e-3)
        assert max(np.abs(fx - ofx).max(), np.abs(fy - ofy).max()) < 1e-4 * scale, t
    assert e.mid_ticks() == 40 and (e.status() == 0).all()


def test_undecidable_pairs_on_the_one_launch_path(amd):
    """receivers planted within rounding of a field-of-view edge and of the line ahead of a source (np.sign(phi),
    vehicle.py:1625): the pair workgroups hand those pairs over with the source's position from the tick's snapshot
    (Dev::src64) while other groups already integrate - the forces match the oracle for every receiver over 12 ticks"""
    rng = np.random.default_rng(3)
    n = 320
    x, y = rng.uniform(0, 60, n), rng.uniform(0, 60, n)
    psi, v = rng.uniform(-np.pi, np.pi, n), rng.uniform(3, 6, n)
    hf = 2 * np.pi / 3
    for k in range(0, 160, 2):                                  # receiver k + 1 on the edge of the field of view of ... as seen from source k
        r, sgn = rng.uniform(3, 20), (-1) ** k
        if k % 4 == 0:                                          # receiver k+1 sees source k exactly at +- hfov / 2 (+- 1e-9 .. 1e-7 rad)
            ang = psi[k + 1] + sgn * (hf / 2 + rng.choice([-1, 1]) * 10 ** rng.uniform(-9, -7))
            x[k], y[k] = x[k + 1] + r * np.cos(ang), y[k + 1] + r * np.sin(ang)
        else:                                                   # receiver k+1 exactly ahead of source k (+- 1e-9 .. 1e-7 m off its line)
            off_ = rng.choice([-1, 1]) * 10 ** rng.uniform(-9, -7)
            x[k + 1] = x[k] + r * np.cos(psi[k]) - off_ * np.sin(psi[k])
            y[k + 1] = y[k] + r * np.sin(psi[k]) + off_ * np.cos(psi[k])
            psi[k + 1] = psi[k] + np.pi + rng.uniform(-0.3, 0.3)   # (facing the source: it is in the receiver's field of view)
    reach = np.array([8.0, 25.0, 60.0, 61.0])
    dq = np.zeros((n, 5, 3)); dq[:, 0, 0], dq[:, 0, 1] = x, y
    dq[:, 1:, 0] = x[:, None] + reach[None, :] * np.cos(psi)[:, None]
    dq[:, 1:, 1] = y[:, None] + reach[None, :] * np.sin(psi)[:, None]
    off = np.arange(n + 1) * 5
    s0 = start_state("twod", x, y, psi, v)
    e = make_engine(amd, "twod", s0, 5.0, off, dq.reshape(-1, 3))
    pop = orc.Population(orc.default_params("twod"), s0, 5.0, off, dq.reshape(-1, 3))
    for t in range(12):
        e.step(1); pop.step(1)
        fx, fy = e.forces(); ofx, ofy = pop.forces()
        scale = max(np.hypot(ofx, ofy).max(), 1e-3)
        assert max(np.abs(fx - ofx).max(), np.abs(fy - ofy).max()) < 1e-4 * scale, t
        anchor(e, pop)                                         # (the planted geometry is rounding-sensitive by construction)
    assert e.mid_ticks() == 12 and e.near_dropped() == 0


def test_population_outgrows_the_one_launch_tick_on_the_other_half_of_its_records(amd, monkeypatch):
    """An ODD number of one-launch ticks leaves the records in the second half of the double buffer; the population then grows
    past the plain kernels' range with a second parameter set, and the class-segmented order pads its runs with the permanent
    sentinel record - which has to exist in that half too (a zero record there is a road user at the scene's origin)."""
    monkeypatch.setenv("CSF_SEGMENTS", "1")
    rng = np.random.default_rng(3)
    n0, n1, box = 600, 3300, 170.0
    pods = [amd.pod("twod"), amd.pod("twod", hfov=1.2 * np.pi, f_0=10.0, sigma_0=0.6)]

    def people(n):
        s = np.zeros((n, 5))
        s[:, 0] = rng.uniform(0.3 * box, box, n); s[:, 1] = rng.uniform(0.3 * box, box, n)     # (nobody near the origin but a phantom)
        s[:, 2] = rng.uniform(-np.pi, np.pi, n); s[:, 3] = rng.uniform(3, 4.8, n)
        d = np.array([40.0, 79.0, 80.0])
        dq = np.zeros((n, 4, 3))
        dq[:, 0, 0] = s[:, 0]; dq[:, 0, 1] = s[:, 1]
        dq[:, 1:, 0] = s[:, 0, None] + d[None, :] * np.cos(s[:, 2])[:, None]
        dq[:, 1:, 1] = s[:, 1, None] + d[None, :] * np.sin(s[:, 2])[:, None]
        return s, dq.reshape(-1, 3)

    s0, dq0 = people(n0)
    e = amd.Engine(pods[0], 4096)
    e.add_agents(s0, 4.5)
    e.set_dest_queue(np.arange(n0), np.arange(n0 + 1) * 4, dq0, reset=True)
    e.step(3)
    assert e.mid_ticks() == 3
    s1, dq1 = people(n1 - n0)
    e.set_param_classes(pods)
    e.add_agents(s1, 4.5)
    e.set_dest_queue(np.arange(n0, n1), np.arange(n1 - n0 + 1) * 4, dq1, reset=True)
    cls = (np.arange(n1) % 2).astype(np.int32)
    e.set_agent_class(np.arange(n1), cls)
    e.step(2)
    assert e.mid_ticks() == 3 and e.count_pairs()[1] == "pair_cull_kernel"
    st = e.state()
    e.calc_forces()
    fdx, fdy, frx, fry = e.force_parts()
    recv = np.arange(0, n1, 3)
    classes = [orc.Params.from_buffer_copy(bytes(p)) for p in pods]
    ox, oy = orc.column_sums(classes, st[:, 0], st[:, 1], st[:, 2], st[:, 3], recv, cls=cls.astype(np.uint8))
    lim, mag = np.hypot(fdx[recv], fdy[recv]), np.hypot(ox, oy)
    sc = np.where(mag > lim, lim / np.maximum(mag, 1e-300), 1.0)
    err = max(np.abs(frx[recv] - ox * sc).max(), np.abs(fry[recv] - oy * sc).max()) / max(np.hypot(ox * sc, oy * sc).max(), 1.0)
    assert err < 1e-4 and (e.status() == 0).all(), err
    e.close()
