"""BalancingRiderBicycle (vehicle.py:1953-1990; dynamics.py:295-705; parameters.py:1214-1411) on the engine: the linearised
Whipple-Carvallo bicycle under full-state feedback whose gains follow the rider's speed, stepped with the implicit midpoint
rule - against the literal reference (tests/golden/balancingrider.npz, make_golden_balancingrider.py) and against the oracle,
alone, in crowds, mixed with the other vehicle classes, and through the host mirror's classes."""
import os

import numpy as np
import pytest

from oracle import csf_oracle as orc
from conftest import shadow_run
from test_gpu_parity import amd, make_engine  # noqa: F401  (amd: fixture)

pytestmark = pytest.mark.gpu


def frame(s):
    """vehicle.s -> the integrator's state (dynamics.py:350-371): roll, steer, their rates, yaw; steer and yaw mirrored"""
    s = np.atleast_2d(s)
    return np.c_[s[:, 5], -s[:, 4], s[:, 7], -s[:, 6], -s[:, 2]]


def test_prescribed_forces_against_the_reference(amd, golden):
    """one vehicle on 120 prescribed forces (dynamics.py:664-705), state for state: the reference solves the midpoint equations
    with MINPACK to ~1e-8, the engine solves the 5 x 5 linear system they are, in fp64"""
    g = golden("balancingrider")
    S, F, X = g["steps_S"], g["steps_F"], g["steps_X"]
    e = make_engine(amd, "balancingrider", S[:1], 5.0, np.array([0, 1]), np.array([[S[0, 0], S[0, 1], 0.0]]))
    assert e.ns == 8
    for t in range(F.shape[0]):
        e.apply_forces(F[t:t + 1, 0], F[t:t + 1, 1])
        np.testing.assert_allclose(e.state()[0], S[t + 1], rtol=0, atol=1e-9, err_msg=f"step {t}")
    x, v_gain, _ = e.integrator_state()
    np.testing.assert_allclose(x[0], X[-1][:5], rtol=0, atol=1e-9)           # (unwrapped yaw included)
    assert abs(v_gain[0] - 0.5 * (S[-1, 3] + S[-2, 3])) < 1e-12 or S[-1, 3] == S[-2, 3]      # dynamics.py:671-673
    assert (e.status() == 0).all()
    e.close()


@pytest.mark.parametrize("tag", ["demo", "dense"])
def test_population_trajectories_against_the_reference(amd, golden, tag):
    """three and sixteen BalancingRiderBicycles through SocialForceIntersection.step (TwoD field, direct-approach destination
    force): 300 / 200 ticks of the literal reference, every tenth state.  Positions to 1e-4 of the extent (the pair forces are
    fp32 sums), the other states to 2e-3."""
    g = golden("balancingrider")
    s0, S = g[f"{tag}_s0"], g[f"{tag}_S"]
    e = make_engine(amd, "balancingrider", s0, g[f"{tag}_vdes"], g[f"{tag}_off"], g[f"{tag}_dq"])
    extent = max(np.ptp(S[..., 0]), np.ptp(S[..., 1]), 1.0)
    for k in range(1, S.shape[0]):
        e.step(10)
        got = e.state()
        np.testing.assert_allclose(got[:, :2], S[k][:, :2], rtol=0, atol=1e-4 * extent, err_msg=f"sample {k}")
        np.testing.assert_allclose(got[:, 2:], S[k][:, 2:], rtol=0, atol=2e-3, err_msg=f"sample {k}")
    assert (e.status() == 0).all()
    e.close()


def crowd(n, box, seed):
    rng = np.random.default_rng(seed)
    s0 = np.zeros((n, 8))
    s0[:, 0] = rng.uniform(0, box, n); s0[:, 1] = rng.uniform(0, box, n)
    s0[:, 2] = rng.uniform(-np.pi, np.pi, n); s0[:, 3] = rng.uniform(2.0, 6.0, n)
    s0[:, 4] = rng.normal(0, 0.02, n); s0[:, 5] = rng.normal(0, 0.02, n)      # steer and roll angles, their rates
    s0[:, 6] = rng.normal(0, 0.05, n); s0[:, 7] = rng.normal(0, 0.05, n)
    d = np.array([30.0, 69.0, 70.0])
    dq = np.zeros((n, 4, 3))
    dq[:, 0, 0] = s0[:, 0]; dq[:, 0, 1] = s0[:, 1]
    dq[:, 1:, 0] = s0[:, 0, None] + d[None, :] * np.cos(s0[:, 2])[:, None]
    dq[:, 1:, 1] = s0[:, 1, None] + d[None, :] * np.sin(s0[:, 2])[:, None]
    return s0, np.arange(n + 1) * 4, dq.reshape(-1, 3)


@pytest.mark.parametrize("n,box,rule,hfov", [(12, 15.0, 0, None), (200, 60.0, 1, 2.5), (1500, 150.0, 0, None), (5000, 260.0, 0, 3.0)])
def test_crowds_vs_oracle(amd, n, box, rule, hfov):
    """crowds of balancing riders against the oracle (which reproduces the reference's trajectories to 1e-14,
    tests/test_oracle_golden.py): the forces of the first tick for every receiver, then 60 ticks in shadow windows of 10 -
    positions, every state of vehicle.s, the integrator's state with its unwrapped yaw, destination pointers"""
    s0, off, dq = crowd(n, box, seed=n)
    over = {} if hfov is None else {"hfov": hfov}
    vdes = np.random.default_rng(1).uniform(3.0, 5.5, n)
    e = make_engine(amd, "balancingrider", s0, vdes, off, dq, rule, **over)
    pop = orc.Population(orc.default_params("balancingrider", priority_rule=rule, **over), s0, vdes, off, dq)
    e.calc_forces(); pop.calc_forces_range(0, n)
    fx, fy = e.forces(); ox, oy = pop.forces()
    assert max(np.abs(fx - ox).max(), np.abs(fy - oy).max()) < 1e-4 * max(np.hypot(ox, oy).max(), 1.0)
    worst, devs, got, ref = shadow_run(e, pop, 60, 10)
    assert worst < 1e-4 * box and (e.status() == 0).all() and e.near_dropped() == 0
    # within a window of 10 ticks: headings to 2e-3 rad, speeds, steer and roll angles to 5e-3, their rates to 5e-2 - for the
    # riders that ride.  Riders the crowd has slowed to walking pace are another matter in this model: the gains of the
    # pole placement grow as the speed falls (dynamics.py:600-615), the handlebar of the LINEAR bicycle then swings at tens of
    # rad/s, in the reference as here, and what two programs agree on is a fraction of that swing (2 %, over the window).
    dd = np.abs(got[:, 2:] - ref[:, 2:]); dd[:, 0] = np.abs((dd[:, 0] + np.pi) % (2 * np.pi) - np.pi)
    swing = np.maximum(np.abs(ref[:, 6]), 10 * np.abs(ref[:, 7]))
    calm = swing < 5.0
    assert calm.sum() > 0.5 * n or n > 1000
    assert (dd[calm] < np.array([2e-3, 5e-3, 5e-3, 5e-3, 5e-2, 5e-2])).all(), dd[calm].max(axis=0)
    assert (dd[~calm] < 2e-2 * np.maximum(swing[~calm], 5.0)[:, None] * np.array([0.1, 0.1, 0.1, 0.1, 1.0, 1.0])).all()
    x, _, _ = e.integrator_state()
    ox_ = pop.lti()[0]
    assert np.abs(x[calm, :4] - ox_[calm, :4]).max() < 5e-2 and np.abs(x[calm, 4] - ox_[calm, 4]).max() < 2e-3
    # vehicle.s is the integrator's state, mirrored, the angles wrapped (dynamics.py:337-348; in a crowd this dense a few
    # riders of the LINEAR bicycle model are pushed through whole turns of the handlebar, in the reference as here)
    np.testing.assert_allclose(frame(got)[:, 2:4], x[:, 2:4], rtol=0, atol=0)
    wrapped = (frame(got)[:, [0, 1, 4]] - x[:, [0, 1, 4]] + np.pi) % (2 * np.pi) - np.pi
    assert np.abs(wrapped).max() < 1e-9
    _, ptr, zn, _ = e.state(with_nav=True)
    optr, ozn, _, _ = pop.nav()
    np.testing.assert_array_equal(ptr, optr)
    e.close()


def test_fixed_poles_and_fixed_gains(amd):
    """parameters.py:1309-1316: a rider with poles given (no pole model: the same poles at every speed) and one with gains given
    (no pole placement at all, dynamics.py:382-391) - against the oracle on 150 ticks of a small crowd"""
    from cyclistsocialforce_amd.parameters import BalancingRiderBicycleParameters

    s0, off, dq = crowd(10, 25.0, seed=3)
    for kw in (dict(poles=(-9.0, -1.5 + 2.0j, -1.5 - 2.0j, -2.5 + 6.0j, -2.5 - 6.0j)), dict(gains=(-12.0, 2.5, -7.5, -0.2, -6.0))):
        pod = BalancingRiderBicycleParameters(**kw).to_pod(6)
        e = amd.Engine(pod, 10)
        e.add_agents(s0, 4.5)
        e.set_dest_queue(np.arange(10), off, dq, reset=True)
        pop = orc.Population(orc.Params.from_buffer_copy(bytes(pod)), s0, 4.5, off, dq)
        worst, _, got, ref = shadow_run(e, pop, 150, 50)
        assert worst < 2.5e-3 and (e.status() == 0).all(), (kw, worst)
        assert np.abs(got[:, 3:6] - ref[:, 3:6]).max() < 5e-3
        e.close()


def test_state_written_between_ticks(amd):
    """vehicle.s written by the user (csf_push_state): the integrator restarts from it (dynamics.py:350-371 - the rates are
    states of vehicle.s, nothing else is hidden), and csf_set_integrator_state puts an unwrapped yaw back"""
    s0, off, dq = crowd(40, 40.0, seed=8)
    e = make_engine(amd, "balancingrider", s0, 4.5, off, dq)
    pop = orc.Population(orc.default_params("balancingrider"), s0, 4.5, off, dq)
    e.step(30); pop.step(30)
    s = e.state()
    s[::2, 4] += 0.05; s[::2, 7] -= 0.1; s[1::2, 2] = ((s[1::2, 2] + 0.3 + np.pi) % (2 * np.pi)) - np.pi
    e.push_state(np.arange(40), s)
    _, ptr, zn, tk = e.state(with_nav=True)
    pop.push_state(s, ptr, zn, col=tk % 3000)
    x, _, _ = e.integrator_state()
    np.testing.assert_allclose(x[:, :4], frame(s)[:, :4], rtol=0, atol=0)
    turns = (x[:, 4] - frame(s)[:, 4]) / (2 * np.pi)                              # (the yaw keeps its winding number)
    np.testing.assert_allclose(turns, np.round(turns), rtol=0, atol=1e-12)
    x[:, 4] += 2 * np.pi                                                         # a winding number: the dynamics does not see it
    e.set_integrator_state(np.arange(40), x=x)
    e.step(20); pop.step(20)
    got, ref = e.state(), pop.state()
    assert np.hypot(got[:, 0] - ref[:, 0], got[:, 1] - ref[:, 1]).max() < 2e-3
    assert np.abs(e.integrator_state()[0][:, 4] - pop.lti()[0][:, 4] - 2 * np.pi).max() < 2e-3
    e.close()


def test_mixed_with_the_other_classes(amd):
    """balancing riders among TwoD, inverted-pendulum and Bicycle road users (rows eight states wide): forces and 50 ticks in
    shadow windows against the oracle; road users leave and arrive in between"""
    rng = np.random.default_rng(17)
    n = 600
    order = ["balancingrider", "twod", "invpend", "bicycle", "balancingrider"]
    pods = [amd.pod(m) for m in order[:4]] + [amd.pod("balancingrider", hfov=2.0, k_p_v=4.0)]
    s0, off, dq = crowd(n, 90.0, seed=21)
    cls = rng.integers(0, len(pods), n).astype(np.uint8)
    for k, m in enumerate(order):                                 # (columns a class does not have stay zero)
        s0[cls == k, orc.N_STATES[orc.MODEL_IDS[m]]:] = 0.0
    e = amd.Engine(pods[0], n + 8)
    e.set_param_classes(pods)
    e.add_agents(s0, 4.5)
    e.set_dest_queue(np.arange(n), off, dq, reset=True)
    e.set_agent_class(np.arange(n), cls)
    assert e.ns == 8
    classes = [orc.Params.from_buffer_copy(bytes(p)) for p in pods]
    pop = orc.Population(classes[0], s0, 4.5, off, dq, ns=8)
    pop.set_classes(classes, cls)
    e.calc_forces(); pop.calc_forces_range(0, n)
    fx, fy = e.forces(); ox, oy = pop.forces()
    assert max(np.abs(fx - ox).max(), np.abs(fy - oy).max()) < 1e-4 * max(np.hypot(ox, oy).max(), 1.0)
    worst, devs, got, ref = shadow_run(e, pop, 50, 10)
    assert worst < 1e-4 * 90.0 and (e.status() == 0).all() and e.near_dropped() == 0
    br = (cls == 0) | (cls == 4)
    assert np.abs(got[br, 3:6] - ref[br, 3:6]).max() < 5e-3
    e.close()


def test_through_the_mirror_classes(golden):
    """BalancingRiderBicycle objects in a SocialForceIntersection, as a script written against the reference builds them: the
    dense crowd of the golden file, vehicle.s eight states long, vehicle.traj [8, 3000]; one rider's roll angle written between
    ticks (it comes back), a TwoDBicycle joining later"""
    from cyclistsocialforce_amd.intersection import SocialForceIntersection
    from cyclistsocialforce_amd.vehicle import BalancingRiderBicycle, TwoDBicycle

    g = golden("balancingrider")
    s0, S, off, dq = g["dense_s0"], g["dense_S"], g["dense_off"], g["dense_dq"]
    bikes = []
    for k in range(s0.shape[0]):
        b = BalancingRiderBicycle(tuple(s0[k]), id=f"b{k}")
        assert b.params.v_desired_default == g["dense_vdes"][k]
        rows = dq[off[k] + 1:off[k + 1]]
        b.setDestinations(rows[:, 0], rows[:, 1])
        bikes.append(b)
    ins = SocialForceIntersection(bikes)
    extent = max(np.ptp(S[..., 0]), np.ptp(S[..., 1]), 1.0)
    for k in range(1, 11):
        for _ in range(10):
            ins.step()
        for j, b in enumerate(bikes):
            assert b.s.shape == (8,) and b.traj.shape == (8, 3000) and b.i == 10 * k
            np.testing.assert_allclose(b.s[:2], S[k][j, :2], rtol=0, atol=1e-4 * extent)
            np.testing.assert_allclose(b.s[2:], S[k][j, 2:], rtol=0, atol=2e-3)
            np.testing.assert_allclose(b.traj[:, 10 * k], b.s, rtol=0, atol=0)
    before = bikes[3].s.copy()
    bikes[3].s[5] += 0.05                                           # a push on the roll angle
    ins.step()
    assert abs(bikes[3].s[5] - before[5]) > 0.01 and abs(bikes[3].s[7]) > abs(before[7])       # (it is felt: a roll rate)
    late = TwoDBicycle((5.0, 5.0, 0.0, 4.0, 0.0), id="late")
    late.setDestinations((40.0, 79.0, 80.0), (5.0, 5.0, 5.0))
    ins.add_road_user(late)
    for _ in range(30):
        ins.step()
    assert late.s.shape == (5,) and late.s[0] > 5.8 and all(np.isfinite(b.s).all() for b in bikes)
    assert abs(bikes[3].s[5]) < 0.05                                # (the rider has caught the roll)


def test_arrivals_and_departures_on_the_device(amd):
    """balancing riders arriving and leaving between ticks through the device-side path (the spawn record carries all eight states;
    patch_kernel starts the integrator from them, dynamics.py:306-307, 350-371) against the same sequence through the host mirror
    (csf_set_incremental(0): download, change, upload): states, integrator states and destination pointers after every call"""
    n0, box = 3300, 420.0
    pool, off, dq = crowd(n0 + 900, box, seed=77)
    dq3 = dq.reshape(-1, 4, 3)
    engines = []
    for inc in (True, False):
        e = amd.Engine(amd.pod("balancingrider"), n0 + 400)
        e.set_incremental(inc)
        e.add_agents(pool[:n0], 4.5)
        e.set_dest_queue(np.arange(n0), np.arange(n0 + 1) * 4, dq3[:n0].reshape(-1, 3), reset=True)
        e.step(3)
        engines.append(e)
    rng = np.random.default_rng(5)
    n, fresh = n0, n0
    for it in range(24):
        op = ("remove", "add", "step")[it % 3]
        if op == "remove":
            idx = np.sort(rng.choice(n, int(rng.integers(5, 40)), replace=False))
            for e in engines:
                e.remove_agents(idx)
            n -= idx.size
        elif op == "add":
            k = int(rng.integers(5, 40))
            new = np.arange(fresh, fresh + k); fresh += k
            for e in engines:
                e.add_agents(pool[new], 4.5)
                e.set_dest_queue(np.arange(n, n + k), np.arange(k + 1) * 4, dq3[new].reshape(-1, 3), reset=True)
            n += k
        else:
            k = int(rng.integers(1, 4))
            for e in engines:
                e.step(k)
        a, b = engines[0].state(with_nav=True), engines[1].state(with_nav=True)
        assert a[0].shape == (n, 8)
        # (a sparse population: few pairs interact, the two engines' fp32 sums are cut at other places - rounding level, growing
        # slowly over the calls: the two are never re-anchored)
        # (... which the rider's feedback gains multiply on their way into the steer and roll rates)
        tol = np.array([5e-5, 5e-5, 5e-5, 1e-3, 5e-4, 5e-5, 1e-2, 1e-3])     # (the speed follows |F| with gain k_p_v t_s = 0.1 per tick)
        assert (np.abs(a[0] - b[0]) < tol).all(), (it, op, np.abs(a[0] - b[0]).max(axis=0))
        np.testing.assert_array_equal(a[1], b[1])
        xa, va, _ = engines[0].integrator_state(); xb, vb, _ = engines[1].integrator_state()
        assert (np.abs(xa - xb) < np.array([5e-5, 5e-4, 1e-3, 1e-2, 5e-5])).all()
        np.testing.assert_allclose(va, vb, rtol=0, atol=1e-3)
    assert all((e.status() == 0).all() for e in engines)
    for e in engines:
        e.close()


@pytest.mark.auto_variant
@pytest.mark.parametrize("tag", ["demo", "dense"])
def test_golden_populations_through_the_one_wave_kernel(amd, golden, tag):
    """the reference's trajectories again on the engine's own choice for 3 and 16 riders: all ticks of a call in one launch of one
    wave (csf_agent.hip: small_tick_kernel), in calls of 10 ticks and of 1"""
    g = golden("balancingrider")
    s0, S = g[f"{tag}_s0"], g[f"{tag}_S"]
    extent = max(np.ptp(S[..., 0]), np.ptp(S[..., 1]), 1.0)
    for per_call in (10, 1):
        e = make_engine(amd, "balancingrider", s0, g[f"{tag}_vdes"], g[f"{tag}_off"], g[f"{tag}_dq"])
        for k in range(1, S.shape[0]):
            for _ in range(10 // per_call):
                e.step(per_call)
            got = e.state()
            np.testing.assert_allclose(got[:, :2], S[k][:, :2], rtol=0, atol=1e-4 * extent, err_msg=f"sample {k}")
            np.testing.assert_allclose(got[:, 2:], S[k][:, 2:], rtol=0, atol=2e-3, err_msg=f"sample {k}")
        assert e.small_ticks() == 10 * (S.shape[0] - 1) and (e.status() == 0).all()
        e.close()


@pytest.mark.parametrize("path", [pytest.param("one wave", marks=pytest.mark.auto_variant), pytest.param("two launches", marks=pytest.mark.cull_variant)])
def test_riders_between_road_edges_against_the_reference(amd, golden, path):
    """three riders on the curve scenario's road (1 530 vertices of eight edges): the road term of intersection.py:853-857 on this
    class, 300 ticks of the literal reference - in the one-wave kernel (the road staged in LDS) and on the general path"""
    g = golden("balancingrider")
    S = g["road_S"]
    e = make_engine(amd, "balancingrider", g["road_s0"], g["road_vdes"], g["road_off"], g["road_dq"])
    e.set_road(g["road_roff"], g["road_verts"], g["road_F0"], g["road_sigma"])
    extent = max(np.ptp(S[..., 0]), np.ptp(S[..., 1]), 1.0)
    for k in range(1, S.shape[0]):
        e.step(10)
        got = e.state()
        np.testing.assert_allclose(got[:, :2], S[k][:, :2], rtol=0, atol=1e-4 * extent, err_msg=f"sample {k}")
        np.testing.assert_allclose(got[:, 2:], S[k][:, 2:], rtol=0, atol=2e-3, err_msg=f"sample {k}")
    assert (e.status() == 0).all() and (e.small_ticks() == 300) == (path == "one wave")
    e.close()


@pytest.mark.auto_variant
@pytest.mark.parametrize("tag,fname", [("pop1", "BR1_ImRe5GivenV_pole-model-params.yaml"), ("pop0", "BR0_ImRe5GivenV_pole-model-params.yaml")])
def test_riders_whose_poles_are_drawn_anew_against_the_reference(tag, fname):
    """`stochastic_control_behavior=True` (parameters.py:1380-1396): every rider draws its poles from its pole model's mixture when
    it is made and again whenever its speed has moved 0.8333 m/s - from NumPy's global generator, so that np.random.seed(77) in front of
    the same construction gives the mirror the reference's draws, tick for tick (the tick is split: forces, draws on the host, then
    the integration - cyclistsocialforce_amd/intersection.py: _resample_poles).  400 ticks of the literal reference
    (tests/golden/make_golden_balancingrider.py stochastic): the poles of every rider after every tick, the trajectory."""
    from cyclistsocialforce_amd import parameters as P
    from cyclistsocialforce_amd.intersection import SocialForceIntersection
    from cyclistsocialforce_amd.vehicle import BalancingRiderBicycle

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "balancingrider_stochastic.npz"))
    s0, vdes, off, dq, S, Pl = (g[f"{tag}_{k}"] for k in ("s0", "vdes", "off", "dq", "S", "poles"))
    np.random.seed(77)
    vs = []
    for a in range(s0.shape[0]):
        prm = P.BalancingRiderBicycleParameters(controlparam_filename=fname, stochastic_control_behavior=True)
        prm.v_desired_default = float(vdes[a])
        b = BalancingRiderBicycle(tuple(s0[a]), id=f"s{a}", params=prm)
        rows = dq[off[a] + 1:off[a + 1]]
        b.setDestinations(rows[:, 0], rows[:, 1], rows[:, 2])
        vs.append(b)
    np.testing.assert_allclose(np.array([v.params.poles for v in vs]), g[f"{tag}_poles0"], rtol=1e-10, atol=1e-11)   # the constructors' draws
    ins = SocialForceIntersection(vs)
    extent = max(np.ptp(S[..., 0]), np.ptp(S[..., 1]), 1.0)
    draws = 0
    for tk in range(Pl.shape[0]):
        before = [np.array(v.params.poles) for v in vs]
        ins.step()
        now = np.array([v.params.poles for v in vs])
        draws += sum(not np.array_equal(x, y) for x, y in zip(before, now))
        np.testing.assert_allclose(now, Pl[tk], rtol=1e-6, atol=1e-7, err_msg=f"poles after tick {tk + 1}")   # (a draw is conditioned on the speed, which carries the fp32 pair sums: 4e-9)
        if (tk + 1) % 10 == 0:
            got = np.array([v.s for v in vs])
            np.testing.assert_allclose(got[:, :2], S[(tk + 1) // 10][:, :2], rtol=0, atol=1e-4 * extent, err_msg=f"tick {tk + 1}")
            np.testing.assert_allclose(got[:, 3], S[(tk + 1) // 10][:, 3], rtol=0, atol=2e-3)
    assert draws >= 3, draws          # (the reference drew anew 5 / 3 times in these runs: the test would be empty without)
