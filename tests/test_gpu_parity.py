"""Parity of the HIP path (through the C ABI, libcsf_hip.so) against the golden vectors of the literal
reference and against the CPU oracle on identical seeded inputs.  All tests need a real MI355X.

Tolerances (stated per BASELINE.json north_star: "trajectories within 1e-4 rel of the CPU reference"):
  * O(N) per-agent work runs in fp64 on the GPU: 2e-7 against the reference (its own FITPACK / lm noise);
  * the all-pairs sum runs in fp32: single pair forces 2e-5 of f_0, trajectories 1e-4 relative to the
    extent of the scene.
"""
import numpy as np
import pytest

from oracle import csf_oracle as orc

pytestmark = pytest.mark.gpu

MODELS = {"bicycle": 0, "twod": 1, "invpend": 2, "planarpoint": 3, "planarbike": 4, "balancingrider": 6}


@pytest.fixture(scope="module")
def amd():
    from cyclistsocialforce_amd import engine, parameters

    class NS:
        pass

    ns = NS()
    ns.Engine = engine.Engine
    ns.pod = parameters.default_pod
    return ns


def make_engine(amd, model, s0, vdes, off, dq, rule=0, capacity=None, **over):
    p = amd.pod(model, priority_rule=rule, **over)
    s0 = np.asarray(s0, dtype=float)
    n = s0.shape[0]
    e = amd.Engine(p, capacity or max(n, 1))
    e.add_agents(s0, vdes)
    # golden queues already hold the (x0, y0, 0) start row the constructor creates: replace
    e.set_dest_queue(np.arange(n), off, dq, reset=True)
    return e


def test_pair_field_twod_golden(amd, golden):
    """vehicle.py:1560-1648 through the pair kernel's device function."""
    g = golden("pair_fields")
    e = amd.Engine(amd.pod("twod"), 4)
    fx = np.zeros(g["x"].size); fy = np.zeros(g["x"].size)
    for k in range(g["x"].size):
        a, b = e.pair_force(np.r_[g["src"][k], 5.0], g["x"][k:k + 1], g["y"][k:k + 1], g["psi"][k:k + 1])
        fx[k], fy[k] = a[0], b[0]
    f0 = 7.0
    # the field's tangential part jumps at phi = 0 (np.sign(phi), vehicle.py:1625): receivers placed within
    # 1e-6 rad of straight ahead are decided by rounding in ANY precision, the reference's fp64 included
    bearing = np.arctan2(g["y"] - g["src"][:, 1], g["x"] - g["src"][:, 0]) - g["src"][:, 2]
    ok = np.abs(np.sin(bearing)) > 1e-6
    assert ok.sum() >= g["x"].size - 8
    np.testing.assert_allclose(fx[ok], g["twod_fx"][ok], rtol=2e-4, atol=2e-5 * f0)
    np.testing.assert_allclose(fy[ok], g["twod_fy"][ok], rtol=2e-4, atol=2e-5 * f0)


def test_pair_field_bicycle_golden(amd, golden):
    """vehicle.py:1054-1147."""
    g = golden("pair_fields")
    e = amd.Engine(amd.pod("bicycle"), 4)
    fx = np.zeros(g["x"].size); fy = np.zeros(g["x"].size)
    for k in range(g["x"].size):
        a, b = e.pair_force(np.r_[g["src"][k], g["src_v"][k]], g["x"][k:k + 1], g["y"][k:k + 1], g["psi"][k:k + 1])
        fx[k], fy[k] = a[0], b[0]
    np.testing.assert_allclose(fx, g["bicycle_fx"], rtol=2e-4, atol=2e-5 * 6.0)
    np.testing.assert_allclose(fy, g["bicycle_fy"], rtol=2e-4, atol=2e-5 * 6.0)


def test_pair_field_far_and_coincident_are_zero(amd):
    """SURVEY finding 4: far pairs are exactly 0 (never NaN); coincident agents contribute 0."""
    e = amd.Engine(amd.pod("twod"), 4)
    fx, fy = e.pair_force([0, 0, 0, 5], [-3000.0, 0.0, 1e6], [8000.0, 0.0, 1e6], [0.0, 0.3, 1.0])
    assert np.all(fx == 0.0) and np.all(fy == 0.0)


@pytest.mark.parametrize("tag", ["n3", "n16", "n32", "n16_p2r"])
def test_fov_mask_golden(amd, golden, tag):
    """intersection.py:690-745: the kernel's dot-product test reproduces get_untracked_foes."""
    g = golden("masks_totals")
    rule = int(g[f"{tag}_p2r"])
    e = amd.Engine(amd.pod("twod", priority_rule=rule), 4)
    s0 = g[f"{tag}_s0"]
    U = g[f"{tag}_untracked0"]
    n = s0.shape[0]
    for i in range(n):
        fx, fy = e.pair_force(np.r_[s0[i, :3], 5.0], s0[:, 0], s0[:, 1], s0[:, 2], apply_fov=True)
        fx0, fy0 = e.pair_force(np.r_[s0[i, :3], 5.0], s0[:, 0], s0[:, 1], s0[:, 2], apply_fov=False)
        zero = (fx == 0) & (fy == 0)
        live = (np.hypot(fx0, fy0) > 0)          # pairs whose unmasked force is representable
        np.testing.assert_array_equal(zero[live], U[i][live], err_msg=f"source {i}")
        assert zero[i]


@pytest.mark.parametrize("tag", ["n2", "n3", "n16", "n32", "n16_p2r"])
def test_calc_forces_golden(amd, golden, tag):
    """intersection.py:747-864 after one warm-up tick (spline branch of the destination force)."""
    g = golden("masks_totals")
    rule = int(g[f"{tag}_p2r"])
    e = make_engine(amd, "twod", g[f"{tag}_s0"], g[f"{tag}_vdes"], g[f"{tag}_off"], g[f"{tag}_dq"], rule)
    e.step(1)
    np.testing.assert_allclose(e.state(), g[f"{tag}_s1"], rtol=1e-5, atol=1e-5)
    fx, fy = e.calc_forces()
    np.testing.assert_allclose(fx, g[f"{tag}_Fx1"], rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(fy, g[f"{tag}_Fy1"], rtol=2e-4, atol=2e-4)


def test_control_move_golden(amd, golden):
    """vehicle.py:1218-1272 through csf_apply_forces (one engine per case class, batched)."""
    g = golden("control_move")
    for is_bike, model in ((False, "twod"), (True, "bicycle")):
        sel = np.where(g["is_bicycle"] == is_bike)[0]
        s = g["s"][sel]
        n = len(sel)
        e = amd.Engine(amd.pod(model), n)
        e.add_agents(s, 5.0)
        rows, off = [], [0]
        for k in sel:
            d = g["dest"][k]
            rows.append([d[0], d[1], 0.0])
            if d[2] == 0:                     # not the last destination: a second row keeps isLastDest False
                rows.append([d[0], d[1], 0.0])
            off.append(len(rows))
        e.set_dest_queue(np.arange(n), off, np.array(rows), reset=True)
        e.apply_forces(g["F"][sel, 0], g["F"][sel, 1])
        np.testing.assert_allclose(e.state(), g["s_next"][sel], rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("tag", ["demo_a", "turn", "stop_last", "stop_mid", "pp_turn"])
def test_dest_force_runs_golden(amd, golden, tag):
    """vehicle.py:354-457, 545-594, 1416-1558: single agent in closed loop on its destination force."""
    g = golden("dest_force_runs")
    model = str(g[f"{tag}_model"])
    S = g[f"{tag}_s"]
    dq = g[f"{tag}_dq"]
    e = make_engine(amd, model, S[:1], float(g[f"{tag}_vdes"]), [0, dq.shape[0]], dq)
    T = g[f"{tag}_Fdest"].shape[0]
    tol = 2e-7
    for t in range(T):
        fx, fy = e.dest_force()
        s, ptr, zn, _ = e.state(with_nav=True)
        np.testing.assert_allclose([fx[0], fy[0]], g[f"{tag}_Fdest"][t], rtol=tol, atol=tol, err_msg=f"tick {t}")
        assert ptr[0] == g[f"{tag}_ptr"][t], f"tick {t}"
        np.testing.assert_array_equal(zn[0], g[f"{tag}_znav"][t], err_msg=f"tick {t}")
        e.apply_forces(fx, fy)
        np.testing.assert_allclose(e.state()[0], S[t + 1], rtol=tol, atol=tol, err_msg=f"tick {t}")
    assert e.status()[0] == 0


def test_planarpoint_steps_golden(amd, golden):
    """dynamics.py:996-1079 (closed-form implicit midpoint)."""
    g = golden("planarpoint_steps")
    s0 = g["s01"][:, :4]
    e = amd.Engine(amd.pod("planarpoint"), s0.shape[0])
    e.add_agents(s0, 5.0)
    e.apply_forces(g["F01"][:, 0], g["F01"][:, 1])
    np.testing.assert_allclose(e.state(), g["s01"][:, 4:], rtol=1e-7, atol=1e-8)
    e.apply_forces(g["F01"][:, 2], g["F01"][:, 3])
    np.testing.assert_allclose(e.state(), g["s2"], rtol=1e-7, atol=1e-8)


def test_road_edges_golden(amd, golden):
    """intersection.py:118-242 on the curve-scenario geometry: total - dest - repulsive = road term."""
    g = golden("road_edges")
    n = g["x"].size
    s0 = np.c_[g["x"], g["y"], np.zeros(n), np.full(n, 4.0)]
    e = amd.Engine(amd.pod("planarpoint", f_0=0.0), n)   # f_0 = 0 switches the agent-agent field off
    e.add_agents(s0, 5.0)
    e.set_road(g["off"], g["verts"], g["F0"], g["sigma"])
    fx, fy = e.calc_forces()
    fdx, fdy, frx, fry = e.force_parts()
    assert np.all(frx == 0) and np.all(fry == 0)
    scale = np.maximum(np.hypot(g["Fx"], g["Fy"]), 1e-3)
    assert np.max(np.abs(fx - fdx - g["Fx"]) / scale) < 5e-5
    assert np.max(np.abs(fy - fdy - g["Fy"]) / scale) < 5e-5


TRAJ = [
    ("demo_twod", "twod", 0), ("demo_bicycle", "bicycle", 0), ("demo_planarpoint", "planarpoint", 0),
    ("demo_invpend", "invpend", 0), ("dense_twod", "twod", 0), ("dense_bicycle", "bicycle", 0),
    ("dense_planarpoint", "planarpoint", 0), ("dense_invpend", "invpend", 0), ("p2r_twod", "twod", 1),
    ("road_pp", "planarpoint", 0), ("lap_twod", "twod", 0),
]


@pytest.mark.parametrize("prefix,model,rule", TRAJ)
def test_population_trajectories_golden(amd, golden, prefix, model, rule):
    """intersection.py:866-896 end to end against the literal reference (1e-4 of the scene extent)."""
    g = golden("trajectories")
    e = make_engine(amd, model, g[f"{prefix}_s0"], g[f"{prefix}_vdes"], g[f"{prefix}_off"], g[f"{prefix}_dq"], rule)
    if f"{prefix}_verts" in g.files:
        e.set_road(g[f"{prefix}_roff"], g[f"{prefix}_verts"], g[f"{prefix}_F0"], g[f"{prefix}_sigma"])
    S = g[f"{prefix}_S"]
    every = {"lap_twod": 50}.get(prefix, 10)
    extent = max(np.ptp(S[..., 0]), np.ptp(S[..., 1]), 1.0)
    worst = 0.0
    for k in range(1, S.shape[0]):
        e.step(every)
        got = e.state()
        worst = max(worst, np.abs(got[:, :2] - S[k][:, :2]).max() / extent)
        np.testing.assert_allclose(got[:, :2], S[k][:, :2], rtol=0, atol=1e-4 * extent, err_msg=f"{prefix} sample {k}")
        np.testing.assert_allclose(got[:, 3], S[k][:, 3], rtol=0, atol=2e-3, err_msg=f"{prefix} speed sample {k}")
    assert (e.status() == 0).all()
    print(f"{prefix}: worst position deviation / extent = {worst:.3e}")


@pytest.mark.auto_variant
@pytest.mark.parametrize("every_n", [None, 1])
@pytest.mark.parametrize("prefix,model,rule", TRAJ)
def test_population_trajectories_golden_own_choice(amd, golden, prefix, model, rule, every_n):
    """The same runs on the kernels the engine chooses itself, which is what a drop-in user of these population sizes gets:
    the plain all-pairs kernel (csf_engine.hip: pair_variant_for) - the ticks between two samples in one call, or
    (every_n = 1) a call per tick as SocialForceIntersection.step() issues them."""
    g = golden("trajectories")
    e = make_engine(amd, model, g[f"{prefix}_s0"], g[f"{prefix}_vdes"], g[f"{prefix}_off"], g[f"{prefix}_dq"], rule)
    if f"{prefix}_verts" in g.files:
        e.set_road(g[f"{prefix}_roff"], g[f"{prefix}_verts"], g[f"{prefix}_F0"], g[f"{prefix}_sigma"])
    S = g[f"{prefix}_S"]
    every = {"lap_twod": 50}.get(prefix, 10)
    if every_n == 1 and prefix == "lap_twod":
        pytest.skip("3100 single-tick calls: covered by the other samples")
    extent = max(np.ptp(S[..., 0]), np.ptp(S[..., 1]), 1.0)
    worst = 0.0
    for k in range(1, S.shape[0]):
        if every_n is None:
            e.step(every)
        else:
            for _ in range(every):
                e.step(1)
        got = e.state()
        worst = max(worst, np.abs(got[:, :2] - S[k][:, :2]).max() / extent)
        np.testing.assert_allclose(got[:, :2], S[k][:, :2], rtol=0, atol=1e-4 * extent, err_msg=f"{prefix} sample {k}")
        np.testing.assert_allclose(got[:, 3], S[k][:, 3], rtol=0, atol=2e-3, err_msg=f"{prefix} speed sample {k}")
    assert (e.status() == 0).all() and e.count_pairs()[1] == "pair_kernel" and e.near_dropped() == 0
    print(f"{prefix}: worst position deviation / extent = {worst:.3e}")


@pytest.mark.auto_variant
@pytest.mark.parametrize("model,n,kernel", [("twod", 1024, "pair_kernel"), ("twod", 2900, "pair_kernel"), ("twod", 3100, "pair_cull_kernel"),
                                             ("bicycle", 2900, "pair_kernel"), ("bicycle", 3100, "pair_bike_kernel"),
                                             ("invpend", 300, "pair_kernel"), ("planarpoint", 70, "pair_kernel")])
def test_pair_kernel_chosen_by_population_size(amd, monkeypatch, model, n, kernel):
    """plain below 3 072 road users, cull-first (binned) from there - each against the oracle"""
    box = float(np.sqrt(n / 0.2))
    x, y, psi, v, off, dq = synthetic_population(n, box, seed=4)
    s0 = np.zeros((n, orc.N_STATES[MODELS[model]])); s0[:, 0] = x; s0[:, 1] = y; s0[:, 2] = psi; s0[:, 3] = v
    e = make_engine(amd, model, s0, 5.0, off, dq)
    pop = orc.Population(orc.default_params(model), s0, 5.0, off, dq)
    e.step(40); pop.step(40)
    assert e.count_pairs()[1] == kernel
    assert np.abs(e.state()[:, :2] - pop.state()[:, :2]).max() < 1e-4 * box and (e.status() == 0).all()
    # a population that shrinks below the threshold changes kernel at the next re-binning
    if n == 3100:
        e.remove_agents(np.arange(0, 400))
        e.step(70)                                               # (past the next re-binning: the engine re-bins every 64 ticks)
        assert e.count_pairs()[1] == "pair_kernel" and np.isfinite(e.state()).all()


def synthetic_population(n, box, seed=0):
    """SURVEY.md §8(d) synthetic inputs."""
    rng = np.random.default_rng(seed)
    x = rng.uniform(0, box, n); y = rng.uniform(0, box, n)
    psi = rng.uniform(-np.pi, np.pi, n); v = rng.uniform(3, 6, n)
    d = np.array([50.0, 99.0, 100.0])
    dq = np.zeros((n, 4, 3))
    dq[:, 0, 0] = x; dq[:, 0, 1] = y
    dq[:, 1:, 0] = x[:, None] + d[None, :] * np.cos(psi)[:, None]
    dq[:, 1:, 1] = y[:, None] + d[None, :] * np.sin(psi)[:, None]
    off = np.arange(n + 1) * 4
    return x, y, psi, v, off, dq.reshape(-1, 3)


@pytest.mark.parametrize("model,n,box,ticks", [("twod", 1024, 200.0, 200), ("invpend", 512, 120.0, 100),
                                                ("bicycle", 512, 120.0, 100), ("planarpoint", 512, 120.0, 100)])
def test_random_population_vs_oracle(amd, model, n, box, ticks):
    """BASELINE config 2 shape (random placement in an open square) against the CPU oracle."""
    x, y, psi, v, off, dq = synthetic_population(n, box)
    ns = orc.N_STATES[MODELS[model]]
    s0 = np.zeros((n, ns)); s0[:, 0] = x; s0[:, 1] = y; s0[:, 2] = psi; s0[:, 3] = v
    e = make_engine(amd, model, s0, 5.0, off, dq)
    pop = orc.Population(orc.default_params(model), s0, 5.0, off, dq)
    e.step(ticks); pop.step(ticks)
    got, ref = e.state(), pop.state()
    rel = np.abs(got[:, :2] - ref[:, :2]).max() / box
    print(f"{model} N={n}: max |dpos| / box after {ticks} ticks = {rel:.3e}")
    assert rel < 1e-4
    assert np.abs(got[:, 3] - ref[:, 3]).max() < 5e-3
    fx, fy = e.forces(); ox, oy = pop.forces()
    assert np.abs(np.c_[fx - ox, fy - oy]).max() < 5e-3
    assert (e.status() == 0).all()


def test_full_size_properties_16k(amd):
    """N = 16384 (BASELINE metric size): properties that need no oracle run.
    (a) bit-reproducible; (b) invariant under a permutation of the agents; (c) superposition of source
    sets before the clamp; (d) a strided sample of receivers against the oracle's single-receiver sum."""
    n, box = 16384, 200.0
    x, y, psi, v, off, dq = synthetic_population(n, box)
    s0 = np.c_[x, y, psi, v, np.zeros(n)]
    big = 1e6                                     # |F_dest| = v_desired on tick 0: clamp never active
    e1 = make_engine(amd, "twod", s0, big, off, dq)
    e1.calc_forces()
    _, _, rx1, ry1 = e1.force_parts()
    e2 = make_engine(amd, "twod", s0, big, off, dq)
    e2.calc_forces()
    _, _, rx2, ry2 = e2.force_parts()
    assert np.array_equal(rx1, rx2) and np.array_equal(ry1, ry2)                      # (a)
    perm = np.random.default_rng(1).permutation(n)
    dq4 = dq.reshape(n, 4, 3)
    e3 = make_engine(amd, "twod", s0[perm], big, off, dq4[perm].reshape(-1, 3))
    e3.calc_forces()
    _, _, rx3, ry3 = e3.force_parts()
    scale = np.hypot(rx1, ry1).max()
    assert np.abs(rx3 - rx1[perm]).max() < 2e-5 * scale                               # (b)
    assert np.abs(ry3 - ry1[perm]).max() < 2e-5 * scale
    # (c) receivers 0..63 feel the source sets A = [64, n/2) and B = [n/2, n) additively; every sub-engine also
    #     contains the 64 receivers themselves (set D), so  F(D+A) + F(D+B) - F(D) = F(D+A+B)
    half = n // 2
    def sub(lo, hi):
        idx = np.r_[np.arange(64), np.arange(lo, hi)]
        ee = make_engine(amd, "twod", s0[idx], big, np.arange(idx.size + 1) * 4, dq4[idx].reshape(-1, 3))
        ee.calc_forces()
        return [a[:64] for a in ee.force_parts()[2:]]
    ax, ay = sub(64, half); bx, by = sub(half, n); cx, cy = sub(64, n); dx_, dy_ = sub(64, 64)
    assert np.abs(ax + bx - dx_ - cx).max() < 2e-5 * scale and np.abs(ay + by - dy_ - cy).max() < 2e-5 * scale
    assert np.abs(cx - rx1[:64]).max() < 2e-5 * scale and np.abs(cy - ry1[:64]).max() < 2e-5 * scale
    # (d) sample of receivers against the oracle's pair function
    p = orc.default_params("twod")
    for j in range(0, n, 1024):
        U = np.array([orc.lib().csfo_untracked(p.hfov, 0, i, j, x[i], y[i], x[j], y[j], psi[j]) for i in range(n)], dtype=bool)
        src = np.where(~U)[0]
        fx = fy = 0.0
        for i in src:
            a, b = orc.pair_twod(p, (x[i], y[i], psi[i]), x[j:j + 1], y[j:j + 1], psi[j:j + 1])
            fx += a[0]; fy += b[0]
        assert abs(fx - rx1[j]) < 1e-4 * max(1.0, np.hypot(fx, fy)), j
        assert abs(fy - ry1[j]) < 1e-4 * max(1.0, np.hypot(fx, fy)), j


def test_edge_cases(amd):
    """Empty population, single agent (intersection.py:849-851), ragged queues, add/remove mid-run."""
    p = amd.pod("twod")
    e = amd.Engine(p, 8)
    e.step(3)                                           # n == 0: intersection.py:888
    assert e.n == 0 and e.tick == 3
    e.add_agents(np.array([[0.0, 0, 0, 5, 0]]), 5.0)
    e.set_dest_queue([0], [0, 2], [[50.0, 0, 0], [100.0, 0, 0]])
    e.step(10)
    s = e.state()
    assert s.shape == (1, 5) and s[0, 0] > 0.3 and abs(s[0, 1]) < 1e-9
    # ragged queues + removal keeps order
    e.add_agents(np.array([[0.0, 10, 0, 5, 0], [0.0, 20, 0, 5, 0], [0.0, 30, 0, 5, 0]]), [4.0, 5.0, 6.0])
    e.set_dest_queue([1, 3], [0, 1, 4], [[60.0, 10, 0], [20.0, 30, 0], [40.0, 31, 0], [60.0, 30, 1]])
    e.step(5)
    before = e.state()
    e.remove_agents([2])
    after = e.state()
    np.testing.assert_array_equal(after, before[[0, 1, 3]])
    e.step(5)
    assert e.n == 3 and np.isfinite(e.state()).all()
    with pytest.raises(Exception):
        e.add_agents(np.zeros((9, 5)), 5.0)             # capacity
    with pytest.raises(Exception):
        e.remove_agents([7])


def test_history_and_push_state(amd):
    p = amd.pod("twod")
    e = amd.Engine(p, 4)
    e.add_agents(np.array([[0.0, 0, 0, 5, 0], [0.0, 50, 0, 4, 0]]), 5.0)
    e.set_dest_queue([0, 1], [0, 1, 2], [[100.0, 0, 0], [100.0, 50, 0]])
    e.enable_history(stride=2, capacity=16)
    states = []
    for _ in range(10):
        e.step(1)
        states.append(e.state())
    h = e.history(0, 5)
    for k in range(5):
        np.testing.assert_array_equal(h[k], states[2 * k + 1])
    s = e.state()
    s[0, 0] += 1.0
    e.push_state([0], s[0:1])
    np.testing.assert_array_equal(e.state()[0], s[0])


@pytest.mark.cull_variant
def test_single_rank_communicator_path(amd):
    """world = 1 with a real RCCL communicator: the sharded tick (split agent phases, all-gather on the comm
    stream, event choreography) must reproduce the unsharded engine bit for bit."""
    n, box = 1500, 120.0
    x, y, psi, v, off, dq = synthetic_population(n, box, seed=4)
    s0 = np.c_[x, y, psi, v, np.zeros(n)]
    ref = make_engine(amd, "twod", s0, 5.0, off, dq)
    ref.step(40)
    e = make_engine(amd, "twod", s0, 5.0, off, dq)
    uid = amd.Engine.comm_unique_id()
    assert len(uid) == 128
    e.comm_init(uid, 0, 1)
    assert e.shard_range() == (0, n)
    e.step(40)
    np.testing.assert_array_equal(e.state(), ref.state())
    # the population changes with the communicator in place (every rank would make the same calls: csf_engine.hip gather_population)
    for eng in (ref, e):
        eng.remove_agents(np.arange(100, 140))
        eng.add_agents(s0[:25] + np.array([0.3, 0.2, 0.0, 0.0, 0.0]), 4.0)
        eng.set_dest_queue(np.arange(n - 40, n - 15), np.arange(26) * 4, dq.reshape(-1, 4, 3)[:25].reshape(-1, 3), reset=True)
        eng.step(12)
    assert e.n == ref.n == n - 15 and e.shard_range() == (0, n - 15)
    # (the unsharded engine takes arrivals on the device, this one through its host mirror: other slots, another summation order)
    np.testing.assert_allclose(e.state(), ref.state(), rtol=0, atol=1e-4)


@pytest.mark.parametrize("rpb", [8, 16, 32])
@pytest.mark.parametrize("hfov,rule", [(0.6, 0), (np.pi * 2 / 3, 1), (np.pi, 0), (4.0, 0), (4.0, 1), (2 * np.pi, 0)])
def test_field_of_view_variants_vs_oracle(amd, monkeypatch, hfov, rule, rpb):
    """Narrow, half-plane, wide and full-circle fields of view, with and without priority-to-the-right
    (intersection.py:733-741), on a binned population (N >= 1024: batch classification where it applies)."""
    monkeypatch.setenv("CSF_RPB", str(rpb))      # receivers per workgroup (32 is chosen from 8192 receivers up)
    n, box = 1600, 90.0
    x, y, psi, v, off, dq = synthetic_population(n, box, seed=11)
    s0 = np.c_[x, y, psi, v, np.zeros(n)]
    e = make_engine(amd, "twod", s0, 1e6, off, dq, rule=rule, hfov=float(hfov))
    e.calc_forces()
    _, _, rx, ry = e.force_parts()
    p = orc.default_params("twod", priority_rule=rule, hfov=float(hfov))
    pop = orc.Population(p, s0, 1e6, off, dq)
    pop.calc_forces_range(0, n)
    _, _, ox, oy = pop.force_parts()
    scale = max(np.hypot(ox, oy).max(), 1.0)
    err = np.abs(np.c_[rx - ox, ry - oy]).max() / scale
    print(f"hfov={hfov:.3f} rule={rule}: max |dF_rep| / max|F_rep| = {err:.2e}")
    assert err < 5e-5


@pytest.mark.parametrize("n,box,binr", [(4096, 500.0, None), (16384, 200.0, None), (8192, 700.0, "1"), (65536, 400.0, None), (65536, 1600.0, None)])
def test_far_field_cull_bound(amd, monkeypatch, n, box, binr):
    """Batches beyond the far-field radius are skipped (include/csf.h: csf_far_radius) and, inside it, the per-pair reach
    test drops every pair whose contribution is provably below eps * f_0 / n (csf_engine.hip: update_far_radius; at
    N = 16 384 in 200 m: four of five pairs inside the field of view).  What is left out of a receiver's column sum must
    stay below eps * f_0; with eps = 0 every pair is evaluated.  With receivers in binned order and candidate tile lists (65 536
    slots and more - the headline's density, and a sparse scene; 8 192 in 700 m by request) the radius comes from the sources
    a receiver can meet instead of from all n (csf_engine.hip: rebin): smaller, the same promise."""
    if binr is not None:
        monkeypatch.setenv("CSF_RECV_BINNED", binr)
    x, y, psi, v, off, dq = synthetic_population(n, box)
    s0 = np.c_[x, y, psi, v, np.zeros(n)]
    f0 = amd.pod("twod").f_0

    def rep(eps):
        if eps is None:
            monkeypatch.delenv("CSF_FAR_EPS", raising=False)
        else:
            monkeypatch.setenv("CSF_FAR_EPS", repr(eps))
        e = make_engine(amd, "twod", s0, 1e6, off, dq)       # |F_dest| = 1e6: the clamp never acts
        e.calc_forces()
        r = e.far_radius()
        _, _, rx, ry = e.force_parts()
        return r, rx, ry

    r0, x0, y0 = rep(0.0)
    assert np.isinf(r0)
    r1, x1, y1 = rep(None)                                    # default: eps = 2^-24
    assert 100.0 < r1 < box * 2 ** 0.5                        # the cull is active in this scene
    r_plain = np.log(n * 2.0 ** 24) / (np.log(16384 * 2.0 ** 24) / 154.7)    # ln(n / eps) / kappa (kappa from the headline's 154.7 m)
    if n >= 65536 or binr == "1":
        monkeypatch.setenv("CSF_FAR_TIGHT", "0")
        r1p, x1p, y1p = rep(None)
        monkeypatch.delenv("CSF_FAR_TIGHT")
        print(f"far radius from the sources within reach {r1:.1f} m, from all n {r1p:.1f} m")
        # (a receiver of 8 192 in 700 m, or of the dense 400 m scene, can meet most of the population: nothing to gain there)
        assert abs(r1p - r_plain) < 1.0 and (r1 < r1p - 4.0 if (n >= 65536 and box > 500.0) else r1 <= r1p + 1e-3)
    else:
        assert abs(r1 - r_plain) < 1.0
    scale = np.hypot(x0, y0).max()
    bound = 2.0 ** -24 * f0 + 8 * np.finfo(np.float32).eps * scale
    if n >= 65536 or binr == "1":
        # the engine without the cull takes its receivers in slot order (scene coordinates), the others in binned order
        # (coordinates relative to their group): another arithmetic - both against the fp64 oracle on a sample of receivers,
        # 1e-4 of the largest sum; the two radii share their arithmetic: the bound itself, for every receiver
        recv = np.arange(0, n, max(1, n // 400))
        ox, oy = orc.column_sums(orc.default_params("twod"), x, y, psi, v, recv)
        assert max(np.abs(x1p[recv] - ox).max(), np.abs(y1p[recv] - oy).max()) <= 1e-4 * scale
        assert max(np.abs(x1[recv] - ox).max(), np.abs(y1[recv] - oy).max()) <= 1e-4 * scale
        assert np.abs(x1 - x1p).max() <= bound and np.abs(y1 - y1p).max() <= bound
        x0, y0 = x1p, y1p
    assert np.abs(x1 - x0).max() <= bound and np.abs(y1 - y0).max() <= bound
    if n >= 65536:
        return
    r2, x2, y2 = rep(1e-3)                                    # a coarse eps: visibly different, still bounded
    assert r2 < r1
    d2 = np.hypot(x2 - x0, y2 - y0).max()
    print(f"far radius {r1:.1f} m (eps 2^-24), {r2:.1f} m (eps 1e-3); max omitted {np.hypot(x1 - x0, y1 - y0).max():.2e} / {d2:.2e}")
    assert d2 <= 1e-3 * f0 + 8 * np.finfo(np.float32).eps * scale


def test_tick_snapshot_equals_separate_readbacks(amd):
    """csf_get_tick (one transfer) returns exactly what csf_get_state + csf_get_forces return."""
    for model, n in (("twod", 300), ("invpend", 65), ("planarpoint", 1)):
        x, y, psi, v, off, dq = synthetic_population(n, 60.0)
        ns = orc.N_STATES[MODELS[model]]
        s0 = np.zeros((n, ns)); s0[:, 0] = x; s0[:, 1] = y; s0[:, 2] = psi; s0[:, 3] = v
        e = make_engine(amd, model, s0, 5.0, off, dq)
        s, ptr, zn, fx, fy, tick = e.tick_snapshot()          # before any tick: the uploaded state
        assert tick == 0 and np.array_equal(s, e.state_by_component())
        e.step(7)
        s, ptr, zn, fx, fy, tick = e.tick_snapshot()
        s2, ptr2, zn2, tick2 = e.state_by_component(with_nav=True)
        fx2, fy2 = e.forces()
        assert tick == tick2 == 7
        assert np.array_equal(s, s2) and np.array_equal(ptr, ptr2) and np.array_equal(zn, zn2)
        assert np.array_equal(fx, fx2) and np.array_equal(fy, fy2)
        s3, _, _, fx3, fy3, _ = e.tick_snapshot(forces=False)
        assert fx3 is None and fy3 is None and np.array_equal(s3, s2)
    empty = amd.Engine(amd.pod("twod"), 4)
    assert empty.tick_snapshot()[0].shape == (0, 5)


@pytest.mark.parametrize("sigmas", [(2.0, 2.0, 2.0), (3.0, 3.0, 3.0), (1.0, 1.0, 1.0), (5.0, 5.0, 5.0),
                                    (2.0, 3.0, 2.5), (0.5, 4.0, 6.0)])
def test_road_force_exponent_paths_vs_oracle(amd, sigmas):
    """RoadEdge.calcRepulsiveForce (intersection.py:226-242): the road kernel takes r^-(sigma+1) as a power of
    rsq(r^2) when every edge shares one integer sigma and through log2/exp2 otherwise; both against the oracle,
    with vertex counts that leave odd and even numbers of 64-vertex batches and a receiver exactly on a vertex."""
    rng = np.random.default_rng(5)
    counts = (70, 64, 131)
    verts = np.vstack([np.c_[np.linspace(0, 40, c), 8.0 * k + rng.normal(0, 0.05, c)] for k, c in enumerate(counts)])
    off = np.r_[0, np.cumsum(counts)]
    F0 = np.array([0.15, 0.05, 0.3])
    sg = np.array(sigmas)
    x = rng.uniform(-5, 45, 400); y = rng.uniform(-4, 22, 400)
    # fp32 records resolve 4e-6 m here and r^-(sigma+1) amplifies that by (sigma+1)/r: keep receivers >= 0.5 m away
    clear = np.min(np.hypot(x[:, None] - verts[None, :, 0], y[:, None] - verts[None, :, 1]), axis=1) >= 0.5
    x, y = x[clear][:96], y[clear][:96]
    n = x.size
    x[0], y[0] = verts[10]                                       # r = 0 for one vertex: that vertex adds nothing
    s0 = np.c_[x, y, np.zeros(n), np.full(n, 4.0)]
    e = amd.Engine(amd.pod("planarpoint", f_0=0.0), n)
    e.add_agents(s0, 5.0)
    e.set_road(off, verts, F0, sg)
    fx, fy = e.calc_forces()
    fdx, fdy, _, _ = e.force_parts()
    keep = np.ones(len(verts), bool); keep[10] = False
    rx, ry = orc.road_force(verts, off, F0, sg, x[1:], y[1:])
    scale = np.maximum(np.hypot(rx, ry), 1e-6)
    assert np.max(np.abs(fx[1:] - fdx[1:] - rx) / scale) < 5e-5
    assert np.max(np.abs(fy[1:] - fdy[1:] - ry) / scale) < 5e-5
    # the receiver on vertex 10: the oracle over the other vertices of that edge and the other edges
    v2 = verts[keep]; off2 = off.copy(); off2[1:] -= 1
    r0x, r0y = orc.road_force(v2, off2, F0, sg, x[:1], y[:1])
    # (its two neighbours on the edge, 0.58 m away, pull with O(1) each and nearly cancel: absolute tolerance)
    assert np.isfinite(fx[0]) and abs(fx[0] - fdx[0] - r0x[0]) < 2e-5
    assert np.isfinite(fy[0]) and abs(fy[0] - fdy[0] - r0y[0]) < 2e-5


@pytest.mark.parametrize("model", ["twod", "invpend"])
def test_full_size_ticks_vs_oracle(amd, model):
    """BASELINE configs 1 and 3 at their full size (16 384 agents, 200 m box): a few whole ticks against the CPU oracle,
    which evaluates every pair in fp64 (the engine runs with its far-field cull on).  Forces: 1e-4 of the largest force
    for EVERY receiver - sources within fp32 rounding of a field-of-view edge are decided as the reference decides them
    (csf_field.h: keep_x2 -> csf_pair.hip: near_drain -> csf_agent.hip: COMBINE), pairs closer than 1 m are evaluated
    from the precise records."""
    n, box, ticks = 16384, 200.0, 3
    x, y, psi, v, off, dq = synthetic_population(n, box)
    ns = orc.N_STATES[MODELS[model]]
    s0 = np.zeros((n, ns)); s0[:, 0] = x; s0[:, 1] = y; s0[:, 2] = psi; s0[:, 3] = v
    e = make_engine(amd, model, s0, 5.0, off, dq)
    assert np.isfinite(e.far_radius()) and e.far_radius() < box * 2 ** 0.5
    p = orc.default_params(model)
    pop = orc.Population(p, s0, 5.0, off, dq)
    e.step(ticks); pop.step(ticks)
    got, ref = e.state(), pop.state()
    moved = np.abs(ref[:, :2] - s0[:, :2]).max()
    err = np.abs(got[:, :2] - ref[:, :2]).max()
    print(f"{model} N={n}: max |dpos| after {ticks} ticks = {err:.3e} m (agents moved up to {moved:.3f} m)")
    assert err < 1e-4 * moved                                   # 1e-4 relative to the distance covered
    fx, fy = e.forces(); ox, oy = pop.forces()
    scale = np.hypot(ox, oy).max()
    df = np.abs(np.c_[fx - ox, fy - oy]).max(axis=1)
    print(f"   force error / max force: median {np.median(df) / scale:.2e}, 99.9 % {np.percentile(df, 99.9) / scale:.2e}, "
          f"max {df.max() / scale:.2e} (receiver {int(df.argmax())})")
    assert np.median(df) < 2e-6 * scale and np.percentile(df, 99.9) < 2e-5 * scale
    assert df.max() < 1e-4 * scale                              # every one of the 16 384 receivers
    assert (e.status() == 0).all() and e.near_dropped() == 0


def test_binned_receivers_and_far_tile_skip_are_exact(amd, monkeypatch):
    """The cull-first kernel takes its receivers by place of the binned order; large populations (CSF_RECV_BINNED=1
    forces it here) also skip tiles of sources that lie beyond the far-field radius of a whole receiver group without
    loading them (csf_pair.hip, SKIP); a rank of a sharded run does the same with the places of ITS receivers.  The skip
    does not change which terms enter a column sum."""
    n, box = 8192, 700.0
    x, y, psi, v, off, dq = synthetic_population(n, box)
    s0 = np.c_[x, y, psi, v, np.zeros(n)]

    def rep(binned, shard=None):
        monkeypatch.setenv("CSF_RECV_BINNED", "1" if binned else "0")
        if shard is None:
            monkeypatch.delenv("CSF_FAKE_SHARD", raising=False)
        else:
            monkeypatch.setenv("CSF_FAKE_SHARD", shard)
        e = make_engine(amd, "twod", s0, 1e6, off, dq)       # |F_dest| = 1e6: the clamp never acts
        e.calc_forces()
        lo, hi = e.shard_range()
        _, _, rx, ry = e.force_parts()
        return lo, hi, rx, ry

    _, _, x0, y0 = rep(False)
    assert np.abs(x0).max() > 0
    _, _, x1, y1 = rep(True)
    # (in binned order the pairs are formed relative to the receiver group's origin, in slot order in scene coordinates
    # with the near pairs corrected: the same terms to rounding)
    sc0 = np.hypot(x0, y0).max()
    dd = np.maximum(np.abs(x0 - x1), np.abs(y0 - y1))
    assert np.median(dd) < 1e-6 * sc0 and np.percentile(dd, 99.5) < 2e-5 * sc0 and dd.max() < 1e-4 * sc0
    for shard in ("0/4", "3/4", "2/3"):
        lo, hi, xs, ys = rep(True, shard)
        assert 0 <= lo < hi <= n and hi - lo < n
        # (a shard splits the sources into a different number of chunks: the same terms, another fp32 summation order)
        scale = np.hypot(x0, y0).max()
        dd = np.maximum(np.abs(xs[lo:hi] - x0[lo:hi]), np.abs(ys[lo:hi] - y0[lo:hi]))
        assert np.median(dd) < 1e-6 * scale and np.percentile(dd, 99.5) < 2e-5 * scale and dd.max() < 1e-4 * scale, shard


@pytest.mark.parametrize("hfov,rule,rpb", [(np.pi * 2 / 3, 0, 16), (np.pi * 2 / 3, 1, 16), (4.0, 0, 16), (2 * np.pi, 0, 16),
                                           (np.pi * 2 / 3, 0, 32), (np.pi * 2 / 3, 1, 32)])
def test_bicycle_field_on_binned_records_vs_oracle(amd, monkeypatch, hfov, rule, rpb):
    """The older elliptic field (vehicle.py:1054-1147) at N >= 1024: binned records, batches outside the field of view
    skipped whole (pair_bike_kernel); column sums of one evaluation and a short run against the oracle."""
    monkeypatch.setenv("CSF_REBIN_TICKS", "32")             # (the engine re-bins every 64 ticks; this case is written around 32)
    monkeypatch.setenv("CSF_RPB", str(rpb))      # receivers per workgroup (32 is chosen from 8192 receivers up)
    n, box = 2048, 120.0
    x, y, psi, v, off, dq = synthetic_population(n, box, seed=3)
    s0 = np.c_[x, y, psi, v, np.zeros(n)]
    e = make_engine(amd, "bicycle", s0, 1e6, off, dq, rule=rule, hfov=float(hfov))
    e.calc_forces()
    _, _, rx, ry = e.force_parts()
    p = orc.default_params("bicycle", priority_rule=rule, hfov=float(hfov))
    pop = orc.Population(p, s0, 1e6, off, dq)
    pop.calc_forces_range(0, n)
    _, _, ox, oy = pop.force_parts()
    scale = max(np.hypot(ox, oy).max(), 1.0)
    err = np.abs(np.c_[rx - ox, ry - oy]).max() / scale
    print(f"bicycle hfov={hfov:.3f} rule={rule}: max |dF_rep| / max|F_rep| = {err:.2e}")
    assert err < 5e-5
    e2 = make_engine(amd, "bicycle", s0, 5.0, off, dq, rule=rule, hfov=float(hfov))
    pop2 = orc.Population(p, s0, 5.0, off, dq)
    e2.step(40); pop2.step(40)                                    # crosses a re-binning (every 32 ticks)
    dev = np.abs(e2.state()[:, :2] - pop2.state()[:, :2]).max(axis=1)
    assert dev.max() < 1e-4 * box                                 # every road user
    assert (e2.status() == 0).all() and e2.near_dropped() == 0


def test_no_receiver_leaves_the_band_at_a_field_of_view_edge(amd, monkeypatch):
    """Round 3 documented a receiver of this run (Bicycle field, hfov = 4.0: an edge at +-2 rad, 2 048 road users) whose
    force differed from the oracle's by 1.3e-2 of the largest force on one tick, because one source 6.9 m away had its
    bearing 2.3e-8 rad from the edge and the fp32 test decided it the other way.  Sources within rounding of an edge are
    now decided as the reference decides them: with the oracle re-anchored on the engine's state before every tick (so that
    every tick compares ONE force evaluation on identical states) no receiver's force differs by more than the 1e-4 of the
    fp32 sums on any of the 40 ticks; with the band switched off (CSF_FOV_BAND=0: the fp32 decision, round 3's kernel) the
    old case is back."""
    monkeypatch.setenv("CSF_RPB", "16")
    hfov, n, box = 4.0, 2048, 120.0
    x, y, psi, v, off, dq = synthetic_population(n, box, seed=3)
    s0 = np.c_[x, y, psi, v, np.zeros(n)]
    p = orc.default_params("bicycle", hfov=hfov)

    def worst_tick(band):
        if band is None:
            monkeypatch.delenv("CSF_FOV_BAND", raising=False)
        else:
            monkeypatch.setenv("CSF_FOV_BAND", band)
        e = make_engine(amd, "bicycle", s0, 5.0, off, dq, hfov=hfov)
        pop = orc.Population(p, s0, 5.0, off, dq)
        worst = 0.0
        for tick in range(40):
            st, ptr, zn, _ = e.state(with_nav=True)
            pop.push_state(st, ptr=ptr, znav=zn)                 # both evaluate this tick's forces on the same state
            e.step(1); pop.step(1)
            fx, fy = e.forces(); ox, oy = pop.forces()
            scale = max(np.hypot(ox, oy).max(), 1.0)
            worst = max(worst, float((np.maximum(np.abs(fx - ox), np.abs(fy - oy)) / scale).max()))
        assert e.near_dropped() == 0
        e.close()
        return worst

    w = worst_tick(None)
    print(f"  exact field-of-view decisions: worst force difference of 40 ticks {w:.1e} of the largest force")
    assert w < 1e-4
    w0 = worst_tick("0")
    print(f"  fp32 decisions (CSF_FOV_BAND=0): {w0:.1e}")
    assert w0 > 1e-3, "the edge case this run is known for has disappeared: choose another seed"


@pytest.mark.auto_variant
@pytest.mark.parametrize("n,kernel", [(1500, "pair_kernel"), (4096, "pair_cull_kernel")])
def test_sign_of_phi_where_the_receiver_sits_on_the_line_ahead_of_a_source(amd, monkeypatch, n, kernel):
    """vehicle.py:1625 takes np.sign(phi): the tangential part of the TwoD field jumps (its direction by up to 25 degrees) where
    the receiver sits exactly ahead of the source.  300 receivers of this crowd are put 3 - 40 m ahead of a source, off its
    heading line by +-1e-9 ... +-1e-6 m - far below what fp32 resolves at coordinates of ~300 m (1.5e-5 m) - and every
    force must still be the oracle's: such a pair is flagged by the fast path (|sin phi| within its rounding band), taken from the
    precise records, and where even they cannot tell (a heading is known to 2^-24 there) the per-agent kernel decides by the
    reference's own chain - acos, limitAngle, sign - on the fp64 states (csf_dev.h: sign_phi_exact).  With the bands switched off
    (CSF_FOV_BAND=0) the fp32 sign is back, and about half of the planted receivers are off."""
    rng = np.random.default_rng(77)
    box, base = 150.0, 220.0                                       # coordinates 220 ... 370 m
    x = base + rng.uniform(0, box, n); y = base + rng.uniform(0, box, n)
    psi = rng.uniform(-np.pi, np.pi, n); v = rng.uniform(3, 6, n)
    m = 300
    src = rng.choice(n // 2, m, replace=False)                     # sources among the first half, receivers among the second
    rcv = n // 2 + rng.choice(n - n // 2, m, replace=False)
    dist = rng.uniform(3.0, 40.0, m)
    offs = np.r_[1e-9, -1e-9, 1e-8, -1e-8, 1e-7, -1e-7, 1e-6, -1e-6][np.arange(m) % 8]     # (phi >= 2.5e-11: far above the reference's own rounding)
    x[rcv] = x[src] + dist * np.cos(psi[src]) - offs * np.sin(psi[src])
    y[rcv] = y[src] + dist * np.sin(psi[src]) + offs * np.cos(psi[src])
    psi[rcv] = psi[src] + np.pi + rng.uniform(-1.0, 1.0, m)        # (facing the source, more or less: it is inside their field of view)
    s0 = np.c_[x, y, psi, v, np.zeros(n)]
    dq = np.zeros((n, 2, 3)); dq[:, 0, 0] = x; dq[:, 0, 1] = y
    dq[:, 1, 0] = x + 60 * np.cos(psi); dq[:, 1, 1] = y + 60 * np.sin(psi)
    off = np.arange(n + 1, dtype=np.int64) * 2
    p = orc.default_params("twod")
    ox, oy = orc.column_sums(p, x, y, psi, v, np.arange(n))

    def planted_off(band):
        if band is None:
            monkeypatch.delenv("CSF_FOV_BAND", raising=False)
        else:
            monkeypatch.setenv("CSF_FOV_BAND", band)
        e = make_engine(amd, "twod", s0, 5.0, off, dq.reshape(-1, 3))
        assert e.count_pairs()[1] == kernel
        e.calc_forces()
        fdx, fdy, rx, ry = e.force_parts()
        lim, mag = np.hypot(fdx, fdy), np.maximum(np.hypot(ox, oy), 1e-300)
        sc = np.minimum(1.0, lim / mag)
        scale = max(np.hypot(ox * sc, oy * sc).max(), 1.0)
        err = np.maximum(np.abs(rx - ox * sc), np.abs(ry - oy * sc)) / scale
        assert e.near_dropped() == 0
        e.close()
        return err

    err = planted_off(None)
    print(f"  {kernel}: worst of all {n} receivers {err.max():.1e}, of the {m} planted ones {err[rcv].max():.1e}")
    assert err.max() < 1e-4
    err0 = planted_off("0")
    print(f"  fp32 sign (CSF_FOV_BAND=0): {(err0[rcv] > 1e-4).sum()} of the {m} planted receivers beyond 1e-4, worst {err0[rcv].max():.1e}")
    assert (err0[rcv] > 1e-4).sum() > m // 10, "the planted pairs no longer show the jump: the test has lost its teeth"


@pytest.mark.parametrize("model,hfov,rule", [("twod", 2 * np.pi / 3, 0), ("twod", 2 * np.pi / 3, 1), ("bicycle", 4.0, 0),
                                             ("twod", np.pi, 0)])
def test_untracked_matrix_is_the_oracles_at_4096(amd, model, hfov, rule):
    """get_untracked_foes (intersection.py:690-745) for 4 096 road users in the headline box, after a few ticks (generic
    fp64 positions and headings): all 16.8 million decisions - fp64, atan2 -> limitAngle -> angleDifference on the fp64
    state, like the reference - equal the oracle's.  The bar for boolean work is exact."""
    n, box = 4096, 200.0
    x, y, psi, v, off, dq = synthetic_population(n, box, seed=11)
    s0 = np.c_[x, y, psi, v, np.zeros(n)]
    e = make_engine(amd, model, s0, 5.0, off, dq, rule=rule, hfov=float(hfov))
    e.step(3)
    st = e.state()
    U = e.untracked()
    want = orc.untracked_matrix(float(hfov), rule, st[:, 0], st[:, 1], st[:, 2])
    assert U.shape == want.shape == (n, n)
    diff = int((U != want).sum())
    print(f"{model} hfov={hfov:.3f} rule={rule}: {diff} of {n * n} decisions differ; {int((~want).sum())} tracked pairs")
    assert diff == 0
    e.close()
