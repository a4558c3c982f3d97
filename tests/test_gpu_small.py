"""The one-wave tick of small populations (csf_agent.hip: small_tick_kernel; include/csf.h: csf_small_ticks): up to 32 road
users of one parameter set are ticked by ONE wave, every tick of a csf_step call in one launch; field-of-view decisions
and np.sign(phi) are taken on the fp64 difference of the two positions - in fp32 with the band of its rounding, and inside
the band by the reference's own fp64 chain.  Against the golden trajectories of the literal
reference, against the oracle on random small crowds, against the general path (pair launch + per-agent launch), and the
conditions under which the engine leaves the path."""
import numpy as np
import pytest

from oracle import csf_oracle as orc
from conftest import shadow_run
from test_gpu_parity import MODELS, amd, make_engine  # noqa: F401  (amd: fixture)

pytestmark = [pytest.mark.gpu, pytest.mark.auto_variant]


@pytest.mark.parametrize("prefix,model", [("demo_twod", "twod"), ("demo_planarpoint", "planarpoint"), ("demo_invpend", "invpend"), ("demo_bicycle", "bicycle"),
                                          ("road_pp", "planarpoint")])
@pytest.mark.parametrize("per_call", [10, 1])
def test_demo_trajectories_golden_through_the_one_wave_kernel(amd, golden, prefix, model, per_call):
    """intersection.py:866-896 for the reference's three-cyclist demo: 1e-4 of the scene extent, as everywhere"""
    g = golden("trajectories")
    e = make_engine(amd, model, g[f"{prefix}_s0"], g[f"{prefix}_vdes"], g[f"{prefix}_off"], g[f"{prefix}_dq"], 0)
    if f"{prefix}_verts" in g.files:                          # the curve scenario's road edges (scenarios/curve-scenario.py): 1 530 vertices
        e.set_road(g[f"{prefix}_roff"], g[f"{prefix}_verts"], g[f"{prefix}_F0"], g[f"{prefix}_sigma"])
    S = g[f"{prefix}_S"]
    assert g[f"{prefix}_s0"].shape[0] <= 32
    extent = max(np.ptp(S[..., 0]), np.ptp(S[..., 1]), 1.0)
    ticks = 0
    for k in range(1, S.shape[0]):
        for _ in range(10 // per_call):
            e.step(per_call)
        ticks += 10
        got = e.state()
        np.testing.assert_allclose(got[:, :2], S[k][:, :2], rtol=0, atol=1e-4 * extent, err_msg=f"{prefix} sample {k}")
        np.testing.assert_allclose(got[:, 3], S[k][:, 3], rtol=0, atol=2e-3, err_msg=f"{prefix} speed sample {k}")
    assert (e.status() == 0).all() and e.small_ticks() == ticks


def crowd(n, seed, box=14.0):
    rng = np.random.default_rng(seed)
    x, y = rng.uniform(0, box, n), rng.uniform(0, box, n)
    psi, v = rng.uniform(-np.pi, np.pi, n), rng.uniform(3, 6, n)
    reach = np.array([8.0, 25.0, 60.0, 61.0])
    dq = np.zeros((n, 5, 3))
    dq[:, 0, 0], dq[:, 0, 1] = x, y
    dq[:, 1:, 0] = x[:, None] + reach[None, :] * np.cos(psi)[:, None]
    dq[:, 1:, 1] = y[:, None] + reach[None, :] * np.sin(psi)[:, None]
    dq[:, 4, 2] = 1.0                                      # the last destination is a stop
    return x, y, psi, v, np.arange(n + 1) * 5, dq.reshape(-1, 3)


@pytest.mark.parametrize("model,n,rule,hfov", [("twod", 8, 0, None), ("twod", 5, 1, None), ("twod", 2, 0, 4.0), ("twod", 1, 0, None),
                                               ("invpend", 6, 0, None), ("planarpoint", 8, 1, 2.0), ("planarpoint", 3, 0, None),
                                               ("twod", 16, 0, None), ("twod", 13, 1, 2.0), ("invpend", 16, 0, 4.0), ("planarpoint", 11, 0, None), ("twod", 16, 0, 4.0),
                                               ("twod", 32, 0, None), ("planarpoint", 27, 1, 2.0), ("invpend", 32, 0, None), ("invpend", 32, 0, 4.0), ("twod", 19, 0, 4.0),
                                               ("bicycle", 7, 0, None), ("bicycle", 32, 1, None), ("bicycle", 12, 0, 4.0),
                                               ("balancingrider", 5, 0, None), ("balancingrider", 14, 1, 2.5), ("balancingrider", 32, 0, None)])
def test_small_crowds_vs_oracle(amd, model, n, rule, hfov):
    """a dense handful (14 m box: every pair matters, fields of view cut through the crowd), forces every tick for 30 ticks,
    trajectories over 400 - against the oracle, and against the engine's general path"""
    box = 14.0 if n <= 8 else (22.0 if n <= 16 else 30.0)
    if model == "balancingrider":
        box *= 2.0             # (the pole placement's gains grow as a rider is slowed to walking pace: tests/test_gpu_balancingrider.py)
    x, y, psi, v, off, dq = crowd(n, seed=10 * n + rule, box=box)
    s0 = np.zeros((n, orc.N_STATES[MODELS[model]])); s0[:, 0] = x; s0[:, 1] = y; s0[:, 2] = psi; s0[:, 3] = v
    over = {} if hfov is None else {"hfov": hfov}
    e = make_engine(amd, model, s0, 5.0, off, dq, rule, **over)
    pop = orc.Population(orc.default_params(model, priority_rule=rule, **over), s0, 5.0, off, dq)
    for t in range(30):
        e.step(1); pop.step(1)
        fx, fy = e.forces(); ofx, ofy = pop.forces()
        scale = max(np.hypot(ofx, ofy).max(), 1e-3)
        assert max(np.abs(fx - ofx).max(), np.abs(fy - ofy).max()) < 1e-4 * scale, (t, n)
    # trajectories: a handful runs free for 400 ticks; a dense crowd of up to 32 is chaotic on that horizon (conftest.shadow_run:
    # the oracle shadows the uninterrupted engine in windows of 10 ticks)
    e2 = make_engine(amd, model, s0, 5.0, off, dq, rule, **over)
    pop2 = orc.Population(orc.default_params(model, priority_rule=rule, **over), s0, 5.0, off, dq)
    ticks = 400
    worst, _, got, ref = shadow_run(e2, pop2, ticks, 400 if n <= 8 else 10)
    assert e2.small_ticks() == ticks and (e2.status() == 0).all()
    extent = max(np.ptp(ref[:, 0]), np.ptp(ref[:, 1]), 14.0)
    assert worst < 1e-4 * extent
    _, ptr, zn, _ = e2.state(with_nav=True)
    optr, ozn, _, _ = pop2.nav()
    np.testing.assert_array_equal(ptr, optr)                  # destination pointers and navigation states: the same decisions
    np.testing.assert_array_equal(np.asarray(zn).reshape(n, 3).astype(bool), ozn)    # (one-hot, as the reference keeps it)


def test_general_path_agrees_and_is_taken_when_asked(amd, monkeypatch):
    x, y, psi, v, off, dq = crowd(7, seed=3)
    s0 = np.c_[x, y, psi, v, np.zeros(7)]
    a = make_engine(amd, "twod", s0, 5.0, off, dq)
    monkeypatch.setenv("CSF_FUSED_SMALL", "0")
    b = make_engine(amd, "twod", s0, 5.0, off, dq)
    monkeypatch.delenv("CSF_FUSED_SMALL")
    a.step(300); b.step(300)
    assert a.small_ticks() == 300 and b.small_ticks() == 0
    assert np.abs(a.state()[:, :2] - b.state()[:, :2]).max() < 2e-5      # (fp32 pair sums on fp32 records against fp64 differences)


def test_the_engine_leaves_the_path_when_it_does_not_apply(amd):
    """a thirty-third road user, a road, profiling: the general path"""
    x, y, psi, v, off, dq = crowd(32, seed=4, box=30.0)
    s0 = np.c_[x, y, psi, v, np.zeros(32)]
    e = make_engine(amd, "twod", s0, 5.0, off, dq, capacity=48)
    e.step(5)
    assert e.small_ticks() == 5
    e.add_agents(np.array([[40.0, 40.0, 0.0, 4.0, 0.0]]), 5.0)
    e.set_dest_queue(np.array([32]), np.array([0, 2]), np.array([[40.0, 40.0, 0.0], [90.0, 40.0, 0.0]]), reset=True)
    e.step(5)
    assert e.small_ticks() == 5 and e.n == 33 and np.isfinite(e.state()).all()
    e.remove_agents(np.array([32], dtype=np.int32))
    e.step(5)
    assert e.small_ticks() == 5 and e.n == 32 and (e.status() == 0).all()   # (its slot stays behind, dead: the general path skips it)
    x, y, psi, v, off, dq = crowd(8, seed=4)
    s0 = np.c_[x, y, psi, v, np.zeros(8)]
    f = make_engine(amd, "twod", s0, 5.0, off, dq)
    f.profile(1)
    f.step(3)
    assert f.small_ticks() == 0 and f.profile_kernels()["pair"][1] == 3     # sampled launches are the general path's
    f.profile(0)
    f.step(2)
    assert f.small_ticks() == 2 and (f.status() == 0).all()
    g = make_engine(amd, "twod", s0, 5.0, off, dq)                           # a road the wave cannot stage (> 2 048 vertices)
    big = np.c_[np.linspace(-5.0, 300.0, 3000), np.full(3000, -6.0)]
    g.set_road(np.array([0, 3000]), big, np.array([0.15]), np.array([2.0]))
    g.step(4)
    assert g.small_ticks() == 0 and np.isfinite(g.state()).all()


@pytest.mark.parametrize("model,n,sigma", [("twod", 5, 2.0), ("planarpoint", 1, 3.0), ("invpend", 3, 2.5), ("bicycle", 8, 2.0)])
def test_small_crowds_between_road_edges_vs_oracle(amd, model, n, sigma):
    """intersection.py:226-242 in the one-wave kernel: two edges of 700 vertices each beside the crowd (integer and fractional
    sigma: the power of rsq and the exp2 / log2 form), total forces against the oracle every tick for 60 ticks"""
    x, y, psi, v, off, dq = crowd(n, seed=40 + n)
    s0 = np.zeros((n, orc.N_STATES[MODELS[model]])); s0[:, 0] = x; s0[:, 1] = y; s0[:, 2] = psi; s0[:, 3] = v
    xs = np.linspace(-20.0, 50.0, 700)
    verts = np.r_[np.c_[xs, np.full(700, -3.0)], np.c_[xs, np.full(700, 17.0)]]
    roff, F0, sg = np.array([0, 700, 1400]), np.array([0.15, 0.2]), np.array([sigma, sigma])
    e = make_engine(amd, model, s0, 5.0, off, dq)
    e.set_road(roff, verts, F0, sg)
    pop = orc.Population(orc.default_params(model), s0, 5.0, off, dq)
    pop.set_road(roff, verts, F0, sg)
    for t in range(60):
        e.step(1); pop.step(1)
        fx, fy = e.forces(); ofx, ofy = pop.forces()
        scale = max(np.hypot(ofx, ofy).max(), 1e-3)
        assert max(np.abs(fx - ofx).max(), np.abs(fy - ofy).max()) < 1e-4 * scale, (t, n)
    assert e.small_ticks() == 60 and (e.status() == 0).all()
    assert np.abs(e.state()[:, :2] - pop.state()[:, :2]).max() < 1e-4 * 70.0


@pytest.mark.parametrize("n,road", [(3, False), (16, False), (5, True), (40, False)])
def test_step_and_read_back_in_one_call(amd, n, road):
    """csf_step_get_tick = csf_step + csf_get_tick, tick for tick: on the one-wave path (the kernel packs the read-back itself)
    and on the general path (40 road users), after single ticks and after several"""
    x, y, psi, v, off, dq = crowd(n, seed=70 + n, box=14.0 if n <= 8 else 35.0)
    s0 = np.c_[x, y, psi, v, np.zeros(n)]
    a, b = make_engine(amd, "twod", s0, 5.0, off, dq), make_engine(amd, "twod", s0, 5.0, off, dq)
    if road:
        xs = np.linspace(-20.0, 40.0, 300)
        for e in (a, b):
            e.set_road(np.array([0, 300]), np.c_[xs, np.full(300, -3.0)], np.array([0.15]), np.array([2.0]))
    for k in (1, 1, 1, 7, 1, 30):
        sa, pa, za, fxa, fya, ta = a.step_snapshot(k)
        b.step(k)
        sb, pb, zb, fxb, fyb, tb = b.tick_snapshot()
        assert ta == tb
        np.testing.assert_array_equal(sa, sb)
        np.testing.assert_array_equal(pa, pb)
        np.testing.assert_array_equal(za, zb)
        np.testing.assert_array_equal(fxa, fxb)
        np.testing.assert_array_equal(fya, fyb)
    assert a.small_ticks() == (41 if n <= 32 else 0) and b.small_ticks() == a.small_ticks()
    s2, _, _, _, _, _ = a.step_snapshot(2, forces=False)       # (outputs may be left out)
    b.step(2)
    np.testing.assert_array_equal(s2, b.state())
