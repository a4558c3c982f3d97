"""The reference's own loop, unchanged in shape, on the engine (needs a MI355X): vehicle constructors ->
SocialForceIntersection -> Scenario.run, checked against the golden trajectories of the literal reference
(demo/demoCSFstandalone.py:101-146 geometry; scenarios/curve-scenario.py road)."""
import numpy as np
import pytest

from cyclistsocialforce_amd.intersection import (CurvedRoadSegment, RoadSegmentCollection,
                                                 SocialForceIntersection, StraightRoadSegment)
from cyclistsocialforce_amd.parameters import RoadElementParameters
from cyclistsocialforce_amd.scenario import Scenario
from cyclistsocialforce_amd.vehicle import (Bicycle, InvPendulumBicycle, PlanarPointBicycle, TwoDBicycle)

pytestmark = pytest.mark.gpu

CLASSES = {"twod": TwoDBicycle, "bicycle": Bicycle, "planarpoint": PlanarPointBicycle, "invpend": InvPendulumBicycle}


def demo_bikes(cls):
    """demoCSFstandalone.py:101-118"""
    a = cls((-23 + 17, 0, 0, 5, 0, 0, 0, 0), id="a", saveForces=True)
    a.params.v_desired_default = 4.5
    b = cls((0 + 15, -20, np.pi / 2, 5, 0, 0, 0, 0), id="b", saveForces=True)
    b.params.v_desired_default = 5.0
    c = cls((-2 + 15, -20, np.pi / 2, 5, 0, 0, 0, 0), id="c", saveForces=True)
    c.params.v_desired_default = 5.0
    a.setDestinations((35, 64, 65), (0, 0, 0))
    b.setDestinations((15, 15, 15), (20, 49, 50))
    c.setDestinations((13, 13, 13), (20, 49, 50))
    return a, b, c


@pytest.mark.parametrize("model", ["twod", "bicycle", "planarpoint", "invpend"])
def test_standalone_demo_through_scenario(golden, model):
    g = golden("trajectories")
    bikes = demo_bikes(CLASSES[model])
    ins = SocialForceIntersection(bikes)

    class Demo(Scenario):                        # demoCSFstandalone.py:94-141
        def __init__(self):
            self.intersection = ins
            Scenario.__init__(self, self._step_func, t_r=0, verbose=False)

        def _step_func(self):
            self.intersection.step()

    scn = Demo()
    scn.run(7)                                   # demoCSFstandalone.py:144-146
    S = g[f"demo_{model}_S"]
    F = g[f"demo_{model}_F"]
    for k, v in enumerate(bikes):
        assert v.i == 700
        np.testing.assert_allclose(v.s, S[-1][k], rtol=0, atol=2e-4)
        np.testing.assert_allclose(v.traj[:2, 10:701:10].T, S[1:, k, :2], rtol=0, atol=2e-4)   # vehicle.traj history
        np.testing.assert_allclose(v.trajF[:, 700], F[-1][k], rtol=0, atol=2e-3)               # saveForces
        np.testing.assert_allclose(v.force, F[-1][k], rtol=0, atol=2e-3)
        assert len(v.F) == 700
    assert ins.hist_n_vecs == [3] * 700
    np.testing.assert_allclose(ins.vehicleX[:, 0], S[-1][:, 0], atol=2e-4)
    if model == "twod":                          # SURVEY.md §8(c) known answers
        np.testing.assert_allclose(bikes[0].s[:2], [21.366756374020543, -0.52069184088739151], atol=2e-4)


def test_planarpoint_on_curve_road(golden):
    """scenarios/curve-scenario.py:63-81 road built with the road classes; agents + static-obstacle forces."""
    g = golden("trajectories")
    rp = RoadElementParameters(sigma=2.0, F_0=0.15)
    s1 = StraightRoadSegment(np.array((0, -20, np.pi / 2)), 5, 25, params=rp, ds=0.1)
    s2 = CurvedRoadSegment(s1.x1, 5, 10, np.pi / 2, "right", params=rp, ds=0.1)
    s3 = CurvedRoadSegment(s2.x1, 5, 10, np.pi / 2, "left", params=rp, ds=0.1)
    s4 = StraightRoadSegment(s3.x1, 5, 20, params=rp, ds=0.1)
    segs = RoadSegmentCollection((s1, s2, s3, s4))
    vs = []
    for k, (x, y) in enumerate(((0.5, -19.0), (-0.8, -16.0), (1.0, -12.0))):
        v = PlanarPointBicycle((x, y, np.pi / 2, 4.0), id=str(k))
        v.setDestinations(*segs.get_destinations_from_segments())
        vs.append(v)
    ins = SocialForceIntersection(vs, road_elements=[segs])
    S = g["road_pp_S"]
    for k in range(1, S.shape[0]):
        ins.step_n(10)
        got = np.array([v.s for v in vs])
        np.testing.assert_allclose(got[:, :2], S[k][:, :2], rtol=0, atol=2e-4, err_msg=f"sample {k}")


def test_mutation_between_ticks_and_population_changes(golden):
    """User edits of vehicle.s / v_desired_default / destinations and add/remove are honoured by the next tick:
    checked against a fresh intersection built directly in the edited configuration."""
    def build(n):
        rng = np.random.default_rng(5)
        vs = []
        for k in range(n):
            x, y, psi = rng.uniform(0, 20), rng.uniform(0, 20), rng.uniform(-np.pi, np.pi)
            v = TwoDBicycle((x, y, psi, 4.0, 0.0), id=f"v{k}")
            v.setDestinations(x + np.array([30, 60]) * np.cos(psi), y + np.array([30, 60]) * np.sin(psi))
            vs.append(v)
        return vs

    vs = build(6)
    ins = SocialForceIntersection(vs[:5])
    ins.step()
    ins.step()
    # edit: teleport 0, slow down 1, new route for 2, remove 3, add the 6th vehicle
    vs[0].s[0] += 2.0
    vs[1].params.v_desired_default = 3.0
    vs[2].setDestinations((50.0,), (50.0,), reset=True)
    ins.remove_road_users_by_id(["v3"])
    ins.add_road_user(vs[5])
    for _ in range(20):
        ins.step()
    assert [v.id for v in ins.vehicles] == ["v0", "v1", "v2", "v4", "v5"]
    assert vs[1].s[3] < 3.6 and vs[0].s[3] > 3.6          # the new desired speed took effect
    assert vs[2].destqueue.shape == (1, 3) and vs[2].isLastDest()
    assert np.isfinite([v.s for v in ins.vehicles]).all() and vs[3]._owner is None
    assert vs[5].i == 20 and vs[0].i == 22
    # a standalone vehicle steps through its private engine (seam 1 of SURVEY.md §8(b))
    lone = vs[3]
    fx, fy = lone.calcDestinationForce()
    before = lone.s.copy()
    lone.step(fx, fy)
    assert not np.array_equal(before, lone.s) and np.isfinite(lone.s).all()


def test_vehicle_state_written_through_a_view_a_rebinding_or_an_old_reference(golden):
    """vehicle.s inside an intersection is a row of the bulk mirror.  The intersection compares the mirror with the device's
    state only once a vehicle.s has been handed out (Vehicle.s); from then on every way the reference object can be edited
    reaches the engine before the next tick: a write through the view (calibration.py:455-460), a rebinding
    (`vehicle.s = array`), and a write through a reference taken many ticks ago."""
    rng = np.random.default_rng(8)
    vs = []
    for k in range(40):
        x, y, psi = rng.uniform(0, 40), rng.uniform(0, 40), rng.uniform(-np.pi, np.pi)
        v = TwoDBicycle((x, y, psi, 4.0, 0.0), id=f"v{k}")
        v.setDestinations(x + np.array([30, 60, 90]) * np.cos(psi), y + np.array([30, 60, 90]) * np.sin(psi))
        vs.append(v)
    ins = SocialForceIntersection(vs)
    for _ in range(3):
        ins.step()
    assert not ins._s_watched                                  # nobody has looked: no compare, no shadow copy per tick
    old_ref = vs[7].s                                          # handed out now ...
    assert ins._s_watched
    for _ in range(3):
        ins.step()
    x_before = float(ins.vehicleX[7, 0])
    old_ref[0] += 5.0                                          # ... written three ticks later
    vs[8].s = np.array([1.0, 2.0, 0.5, 3.0, 0.0])              # rebinding
    vs[9].s[1] -= 4.0                                          # through a fresh view
    y9 = float(vs[9].s[1])
    ins.step()
    e = ins.engine
    dev = e.state()
    assert abs(dev[7, 0] - (x_before + 5.0)) < 0.1             # (+ one tick of motion)
    assert np.hypot(dev[8, 0] - 1.0, dev[8, 1] - 2.0) < 0.1 and vs[8].s is not None and vs[8]._s.base is not None
    assert abs(dev[9, 1] - y9) < 0.1
    np.testing.assert_array_equal(dev, np.array([v.s for v in ins.vehicles]))      # mirror == device after the pull
    np.testing.assert_array_equal(ins.vehicleX[:, 0], dev[:, 0])                   # refreshed when read
    np.testing.assert_array_equal(ins.vehicleTheta[:, 0], dev[:, 2])


def test_vehicle_hooks_against_golden(golden):
    """calcRepulsiveForce / calcDestinationForce / step on single vehicles (vehicle.py:250-328)."""
    g = golden("pair_fields")
    v = TwoDBicycle((0, 0, 0, 5, 0))
    fx, fy = v.calcRepulsiveForce(np.array([5.0, -3.0, 2.0, 0.5]), np.array([0.5, 4.0, -7.0, 0.0]),
                                  np.array([0.0, 1.0, -2.5, np.pi]))
    np.testing.assert_allclose(fx, g["twod_fx"][:4], rtol=2e-4, atol=1e-4)
    np.testing.assert_allclose(fy, g["twod_fy"][:4], rtol=2e-4, atol=1e-4)
    r = golden("dest_force_runs")
    S = r["demo_a_s"]
    a = TwoDBicycle(tuple(S[0]), id="a")
    a.params.v_desired_default = 4.5
    a.setDestinations((35, 64, 65), (0, 0, 0))
    for t in range(60):
        fx, fy = a.calcDestinationForce()
        np.testing.assert_allclose([fx, fy], r["demo_a_Fdest"][t], rtol=2e-7, atol=2e-7)
        a.step(fx, fy)
        np.testing.assert_allclose(a.s, S[t + 1], rtol=2e-7, atol=2e-7)
    assert a.i == 60 and a.destpointer == r["demo_a_ptr"][59]


def test_calibration_replay(golden):
    """calibration.py:438-460: recorded forces replayed through vehicle.step for many sequences at once
    (csf_replay_forces), against the per-call path and the golden closed-loop run."""
    from cyclistsocialforce_amd import parameters
    from cyclistsocialforce_amd.engine import Engine

    r = golden("dest_force_runs")
    S, F = r["demo_a_s"], r["demo_a_Fdest"]
    T = 300
    # three copies of the same recorded sequence, truncated to different lengths
    n = 3
    e = Engine(parameters.default_pod("twod"), n)
    e.add_agents(np.repeat(S[:1], n, axis=0), 4.5)
    dq = r["demo_a_dq"]
    e.set_dest_queue(np.arange(n), np.arange(n + 1) * dq.shape[0], np.tile(dq, (n, 1)), reset=True)
    Fx = np.repeat(F[:T, 0:1], n, axis=1)
    Fy = np.repeat(F[:T, 1:2], n, axis=1)
    lengths = np.array([T, 200, 50])
    out = e.replay_forces(Fx, Fy, lengths=lengths, stride=10)
    assert out.shape == (T // 10, n, 5)
    for k in range(T // 10):
        t = 10 * (k + 1)
        for a in range(n):
            np.testing.assert_allclose(out[k, a], S[min(t, lengths[a])], rtol=2e-7, atol=2e-7)
    np.testing.assert_allclose(e.state()[1], S[200], rtol=2e-7, atol=2e-7)
    # fix_speed: v is set to |F| before each step
    e2 = Engine(parameters.default_pod("twod"), 1)
    e2.add_agents(S[:1], 4.5)
    e2.set_dest_queue([0], [0, dq.shape[0]], dq, reset=True)
    o2 = e2.replay_forces(F[:5, 0:1], F[:5, 1:2], fix_speed=True)
    ref = Engine(parameters.default_pod("twod"), 1)
    ref.add_agents(S[:1], 4.5)
    ref.set_dest_queue([0], [0, dq.shape[0]], dq, reset=True)
    for t in range(5):
        s = ref.state()
        s[0, 3] = np.hypot(F[t, 0], F[t, 1])
        ref.push_state([0], s)
        ref.apply_forces(F[t, 0:1], F[t, 1:2])
        np.testing.assert_allclose(o2[t, 0], ref.state()[0], rtol=1e-12, atol=1e-12)


def test_stop_and_go():
    """Vehicle.stop(0) / Vehicle.go(0) (vehicle.py:459-535): the stop flag of the current destination is edited in
    place and the destination pointer survives the edit."""
    a = TwoDBicycle((0, 0, 0, 5, 0), id="a")
    a.setDestinations((20.0, 40.0, 60.0, 80.0), (0.0, 0.0, 0.0, 0.0))
    b = TwoDBicycle((0, 500, 0, 5, 0), id="b")
    b.setDestinations((200.0, 400.0, 401.0), (500.0, 500.0, 500.0))   # not on its last leg (the reference's
    ins = SocialForceIntersection((a, b))                              # last-leg planner circles far targets)
    for _ in range(300):                   # ~15 m: past the start row, heading for (20, 0)
        ins.step()
    ptr_before = a.destpointer
    assert ptr_before >= 1 and a.znav[0]
    target = a.destqueue[a.destpointer, :2].copy()
    a.stop()
    assert a.destqueue[a.destpointer, 2] == 1.0
    for _ in range(60):
        ins.step()
    assert a.destpointer == ptr_before and a.znav[1] and a.s[3] < 4.5      # braking (vehicle.py:436-450)
    a.go()                                  # abort the stop manoeuvre
    for _ in range(400):
        ins.step()
    assert a.znav[0] and a.s[3] > 4.5 and a.destpointer > ptr_before
    a.stop()                                # and stop for good at the next destination
    ptr_stop = a.destpointer
    target = a.destqueue[ptr_stop, :2].copy()
    for _ in range(2500):
        ins.step()
    assert a.destpointer == ptr_stop        # stopping vehicles do not advance their queue (vehicle.py:567-568)
    assert a.znav[2] and a.s[3] == 0.0 and np.hypot(*(a.s[:2] - target)) < 2.5
    assert b.s[0] > 100 and abs(b.s[1] - 500) < 1e-6 and target is not None


def test_untracked_foes_and_queue_methods(golden):
    """Members of the reference API that run single pieces of the tick: get_untracked_foes (intersection.py:690-745),
    Vehicle.updateDestination (vehicle.py:545-594), updateNavState (vehicle.py:354-457), stop types 1 / 2, go type 1."""
    g = golden("masks_totals")
    for tag in ("n16", "n16_p2r"):
        s0 = g[f"{tag}_s0"]
        bikes = [TwoDBicycle(tuple(r), id=str(k)) for k, r in enumerate(s0)]
        ins = SocialForceIntersection(bikes, priority_rule="p2r" if int(g[f"{tag}_p2r"]) else "unregulated")
        U = ins.get_untracked_foes()
        assert U.shape == (16, 16) and U.dtype == bool
        np.testing.assert_array_equal(U, g[f"{tag}_untracked0"])
    assert SocialForceIntersection([TwoDBicycle((0, 0, 0, 5, 0))]).get_untracked_foes() == np.array(True)
    # queue pointer: 1.5 m from the first destination -> advances; the next-next one is closer -> jumps again
    a = TwoDBicycle((0, 0, 0, 5, 0), id="a")
    a.setDestinations((1.5, 30.0, 31.0, 90.0), (0.0, 0.0, 0.0, 0.0), reset=True)
    a.updateDestination()
    assert a.destpointer == 1
    a.s[0] = 29.5                                         # within d_arrived_inter of (30, 0): advances once - (90, 0) is not
    a.updateDestination()                                 # closer than the destination just left was (vehicle.py:577-583)
    assert a.destpointer == 2
    # ... and twice in one call when the destination behind the next one is closer than the one just reached
    a2 = TwoDBicycle((0, 0, 0, 5, 0), id="a2")
    a2.setDestinations((1.5, 30.0, 31.0, 29.6, 90.0), (0.0, 0.0, 0.0, 0.0, 0.0), reset=True)
    a2.updateDestination()
    assert a2.destpointer == 1
    a2.s[0] = 29.5
    a2.updateDestination()
    assert a2.destpointer == 3
    a2.updateDestination()                                # (29.6, 0) is within reach as well; (90, 0) is the last row
    assert a2.destpointer == 4
    # navigation state machine through its three states with an explicit stop argument
    b = TwoDBicycle((0, 0, 0, 5, 0), id="b")
    b.setDestinations((40.0,), (0.0,), reset=True)
    vd, dd = b.updateNavState(False)
    assert vd == b.params.v_desired_default and abs(dd - 40.0) < 1e-12 and b.znav.tolist() == [True, False, False]
    b.s[0] = 34.0                                         # 6 m to go at 5 m/s: inside the braking distance
    vd, dd = b.updateNavState(True)
    assert b.znav.tolist() == [False, True, False] and 0 < vd < 5.0 and abs(dd - 6.0) < 1e-12
    b.s[0], b.s[3] = 39.0, 0.05
    vd, dd = b.updateNavState(True)
    assert b.znav.tolist() == [False, False, True] and vd == 0.0
    # stop type 1 fails as in the reference (no params.AMAX), type 2 steps the pointer back, go(1) re-evaluates the queue
    c = TwoDBicycle((0, 0, 0, 5, 0), id="c")
    c.setDestinations((1.0, 50.0, 100.0), (0.0, 0.0, 0.0), reset=True)
    c.updateDestination()
    assert c.destpointer == 1
    with pytest.raises(AttributeError, match="AMAX"):
        c.stop(1)
    c.stop(2, (10.0, 0.0))
    assert c.destpointer == 0
    c.go(1)
    assert c.destpointer == 1
    with pytest.raises(ValueError):
        c.stop(3)
    # the same through an intersection (the population engine)
    d1, d2 = TwoDBicycle((0, 0, 0, 5, 0), id="d1"), TwoDBicycle((0, 50, 0, 5, 0), id="d2")
    d1.setDestinations((1.0, 60.0, 120.0), (0.0, 0.0, 0.0), reset=True)
    d2.setDestinations((60.0, 120.0), (50.0, 50.0), reset=True)
    ins = SocialForceIntersection((d1, d2))
    ins.step()
    d1.updateDestination()
    assert d1.destpointer == 1 and d2.destpointer == 0
    ins.step()
    assert np.isfinite(d1.s).all()


def test_custom_force_hooks_on_single_vehicles():
    """vehicle.py:194-204, 250-299: a vehicle built with its own dest_force_func / rep_force_func calls them."""
    seen = []
    v = TwoDBicycle((0, 0, 0, 5, 0), id="h", dest_force_func=lambda veh: (seen.append(veh.id), (1.0, 2.0))[1],
                    rep_force_func=lambda veh, x, y, psi: (np.zeros_like(x) + 3.0, np.zeros_like(x)))
    v.setDestinations((30.0, 60.0), (0.0, 0.0))
    assert v.calcDestinationForce() == (1.0, 2.0) and seen == ["h"]
    fx, fy = v.calcRepulsiveForce(np.array([1.0, 2.0]), np.array([0.0, 0.0]), np.array([0.0, 0.0]))
    assert fx.tolist() == [3.0, 3.0] and fy.tolist() == [0.0, 0.0]


def test_sumo_cosimulation_loop():
    """the per-tick loop of SUMOScenario (scenario.py:376-466) on the engine with duck-typed sumolib /
    traci stand-ins - road users are handed over on the internal lanes, get a prototype across the junction, are
    simulated on the GPU, pushed back with moveToXY every tick, and leave again; arrivals and departures reach the
    device without a rebuild of the population."""
    from sumo_fakes import FakeNet, FakeTraci

    from cyclistsocialforce_amd.utils import angleSFMtoSUMO

    tr = FakeTraci()
    ins = SocialForceIntersection([], id="J", activate_sumo_cosimulation=True, net=FakeNet(), traci=tr, capacity=64)
    arms = {"W": (-1, 0), "E": (1, 0), "S": (0, -1), "N": (0, 1)}
    rng = np.random.default_rng(5)
    on_junction, born, seen_ids, arrivals, entry = {}, 0, [], 0, {}
    for tick in range(600):
        # SUMO's side: a road user enters every 25 ticks and is taken back by SUMO 200 ticks later
        if tick % 25 == 0:
            a, b = rng.choice(list(arms), 2, replace=False)
            ax, ay = arms[a]
            heading = np.arctan2(-ay, -ax)
            on_junction[f"veh{born}"] = dict(route=(a + "_in", b + "_out"), since=tick,
                                             s=[ax * 9.0 + ay * 1.6, ay * 9.0 - ax * 1.6, heading, 4.0, 0.0])
            born += 1
        for vid in [vid for vid, spec in on_junction.items() if tick - spec["since"] >= 200]:
            del on_junction[vid]
        tr.occupancy = {":J_0_0": tuple(on_junction)}
        # scenario.py:376-437: allocate_road_users
        entered, exited = ins.find_entered_exited_roadusers()
        ins.remove_road_users_by_id(list(exited))
        for vid in entered:
            spec = on_junction[vid]
            ins.add_road_user(TwoDBicycle(tuple(spec["s"]), id=str(vid), route=spec["route"]))
            entry[str(vid)] = tuple(spec["s"][:2])
            arrivals += 1
        n_before = len(tr.moves)
        ins.step()
        tr.simulationStep()
        moved = tr.moves[n_before:]
        assert [m[1] for m in moved] == ins.get_road_user_ids()          # one moveToXY per road user, in order
        assert sorted(ins.get_road_user_ids()) == sorted(on_junction)
        for m, v in zip(moved, ins.vehicles):
            assert m[2:4] == ("", -1) and m[7] == 6
            assert m[4] == v.s[0] and m[5] == v.s[1] and abs(m[6] - angleSFMtoSUMO(v.s[2])) < 1e-9
            assert np.isfinite(v.s).all()
        seen_ids += [v.id for v in ins.vehicles if v.id not in seen_ids]
    assert arrivals == 24 and len(seen_ids) == 24
    assert ins.n_bikes == 8 and len(ins.hist_n_vecs) == 600 and max(ins.hist_n_vecs) == 8
    # everyone who left had been driven on by the engine: SUMO's last position of it is metres from its entry
    last = {}
    for m in tr.moves:
        last[m[1]] = (m[4], m[5])
    gone = [vid for vid in last if vid not in ins.get_road_user_ids()]
    assert len(gone) == 16 and all(np.hypot(last[vid][0] - entry[vid][0], last[vid][1] - entry[vid][1]) > 2.0 for vid in gone)


def test_sumo_cosimulation_loop_against_the_reference(golden):
    """The same per-tick loop, scripted, against what the REFERENCE class did with the same duck-typed net / traci
    (tests/golden/sumo_seam.npz, make_golden.py: gen_sumo_seam): who is on the junction after every tick, every moveToXY
    call (road user, position, SUMO angle, keepRoute) and every state - the trajectories after the hand-over run on the
    HIP path."""
    import json

    from sumo_fakes import FakeNet, FakeTraci
    from test_host_api import sumo_script

    g = golden("sumo_seam")
    tr = FakeTraci()
    ins = SocialForceIntersection([], id="J", activate_sumo_cosimulation=True, net=FakeNet(), traci=tr, capacity=16)
    per_tick, specs = sumo_script(g)
    ids_ref = json.loads(str(g["ids"]))
    moves_ref, soff, states_ref = g["moves"], g["state_off"], g["states"]
    mk = 0
    worst = 0.0
    for tick, occ in enumerate(per_tick):
        tr.occupancy = {":J_0_0": occ}
        entered, exited = ins.find_entered_exited_roadusers()
        ins.remove_road_users_by_id(list(exited))
        for vid in entered:
            sp = specs[str(vid)]
            np.random.seed(1000 + int(str(vid)[3:]))
            ins.add_road_user(TwoDBicycle(tuple(sp["s"]), id=str(vid), route=sp["route"]))
        n0 = len(tr.moves)
        ins.step()
        tr.simulationStep()
        assert ins.get_road_user_ids() == ids_ref[tick], tick
        for m in tr.moves[n0:]:
            ref = moves_ref[mk]
            mk += 1
            assert (tick, int(m[1][3:])) == (int(ref[0]), int(ref[1])) and m[2:4] == ("", -1) and m[7] == int(ref[5])
            worst = max(worst, abs(m[4] - ref[2]), abs(m[5] - ref[3]))
            assert abs(m[4] - ref[2]) < 1e-5 and abs(m[5] - ref[3]) < 1e-5 and abs((m[6] - ref[4] + 180.0) % 360.0 - 180.0) < 1e-3, (tick, m, ref)
        got = np.array([np.r_[int(v.id[3:]), v.s] for v in ins.vehicles]).reshape(-1, 6)
        np.testing.assert_allclose(got, states_ref[soff[tick]:soff[tick + 1]], rtol=0, atol=1e-5, err_msg=f"tick {tick}")
    assert mk == moves_ref.shape[0] and ins.hist_n_vecs == list(g["hist_n"])
    print(f"SUMO loop vs the reference: {mk} moveToXY calls, worst position deviation {worst:.1e} m")


@pytest.mark.parametrize("cls", ["twod", "balancingrider"])
def test_animated_demo_on_agg_canvas(tmp_path, cls):
    """the reference's demo with animate=True (demo/demoCSFstandalone.py:120-156) - drawings created on
    the first tick, refreshed from the read-back of every tick, blitted by Scenario, histories plotted afterwards."""
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt

    fig, ax = plt.subplots(1, 1)
    ax.set_xlim(0, 30); ax.set_ylim(-10, 20)
    from cyclistsocialforce_amd.vehicle import BalancingRiderBicycle

    bikes = demo_bikes(TwoDBicycle if cls == "twod" else BalancingRiderBicycle)      # (the demo's default class: eight states)
    ins = SocialForceIntersection(bikes, animate=True, axes=ax)
    scn = Scenario(ins.step, t_r=0, verbose=False, animate=True, axes=ax)
    scn.run(1.0)
    assert all(v.drawing is not None and v.drawing.body.get_animated() for v in bikes)
    for v in bikes:                                              # the artists follow the device state
        assert np.allclose(v.drawing.trajectory.get_xdata()[-1], v.s[0]) and len(v.drawing.trajectory.get_xdata()) == 101
        assert np.allclose(v.drawing.body.get_xy().mean(axis=0), v.s[:2], atol=1.0)
    ins.set_animated(False)
    assert not any(v.drawing.body.get_animated() for v in bikes)
    axs = axf = None
    for v in bikes:
        axs = v.plot_states(t_end=1.0, axes=axs)
        axf = v.plot_forces(t_end=1.0, axes=axf, components_to_plot=["magnitude", "direction"])
    assert len(axs[0].lines) == 3 and len(axf[0].lines) == 3
    fig.savefig(tmp_path / "scene.png")
    late = TwoDBicycle((5.0, 5.0, 0.0, 4.0, 0.0), id="late")
    late.setDestinations((40.0, 80.0), (5.0, 5.0))
    ins.add_road_user(late)                                      # arrivals get an animated drawing (intersection.py:521-524)
    assert late.drawing is not None and late.drawing.body.get_animated()
    ins.step()
    plt.close("all")
    with pytest.raises(AssertionError):
        SocialForceIntersection(demo_bikes(TwoDBicycle), animate=True)


def test_scripts_written_against_the_reference_package_name(golden):
    """compat/cyclistsocialforce in front of the path: the reference demo's own import lines and call sequence
    (demoCSFstandalone.py:23-25, 94-146) on the engine, against the golden demo trajectory."""
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "compat"))
    try:
        from cyclistsocialforce.intersection import SocialForceIntersection
        from cyclistsocialforce.scenario import Scenario
        from cyclistsocialforce.vehicle import InvPendulumBicycle

        class DemoScenario(Scenario):
            def __init__(self):
                bike1 = InvPendulumBicycle((-23 + 17, 0, 0, 5, 0, 0, 0, 0), id="a", saveForces=True)
                bike1.params.v_desired_default = 4.5
                bike2 = InvPendulumBicycle((0 + 15, -20, np.pi / 2, 5, 0, 0, 0, 0), id="b", saveForces=True)
                bike2.params.v_desired_default = 5.0
                bike3 = InvPendulumBicycle((-2 + 15, -20, np.pi / 2, 5, 0, 0, 0, 0), id="c", saveForces=True)
                bike3.params.v_desired_default = 5.0
                bike1.setDestinations((35, 64, 65), (0, 0, 0))
                bike2.setDestinations((15, 15, 15), (20, 49, 50))
                bike3.setDestinations((13, 13, 13), (20, 49, 50))
                self.intersection = SocialForceIntersection((bike1, bike2, bike3), activate_sumo_cosimulation=False)
                Scenario.__init__(self, self._step_func, verbose=False)

            def _step_func(self):
                self.intersection.step()

        scn = DemoScenario()
        scn.run(7)
        g = golden("trajectories")
        got = np.array([v.s for v in scn.intersection.vehicles])
        np.testing.assert_allclose(got[:, :2], g["demo_invpend_S"][-1][:, :2], rtol=0, atol=2e-4)
    finally:
        sys.path.remove(os.path.join(root, "compat"))
        for k in [k for k in sys.modules if k == "cyclistsocialforce" or k.startswith("cyclistsocialforce.")]:
            del sys.modules[k]


def test_the_reference_default_demo_with_balancing_riders(golden):
    """`python demo/demoCSFstandalone.py` as the reference ships it runs -m balancingrider: three BalancingRiderBicycles, t = 7 s
    (:101-118, 144-146).  The same constructor calls, intersection and Scenario.run through the package with the reference's
    name (compat/), against the literal reference's own run of it."""
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "compat"))
    try:
        from cyclistsocialforce.intersection import SocialForceIntersection
        from cyclistsocialforce.scenario import Scenario
        from cyclistsocialforce.vehicle import BalancingRiderBicycle as bike_type

        class DemoScenario(Scenario):
            def __init__(self):
                bike1 = bike_type((-23 + 17, 0, 0, 5, 0, 0, 0, 0), id="a", saveForces=True)
                bike1.params.v_desired_default = 4.5
                bike2 = bike_type((0 + 15, -20, np.pi / 2, 5, 0, 0, 0, 0), id="b", saveForces=True)
                bike2.params.v_desired_default = 5.0
                bike3 = bike_type((-2 + 15, -20, np.pi / 2, 5, 0, 0, 0, 0), id="c", saveForces=True)
                bike3.params.v_desired_default = 5.0
                bike1.setDestinations((35, 64, 65), (0, 0, 0))
                bike2.setDestinations((15, 15, 15), (20, 49, 50))
                bike3.setDestinations((13, 13, 13), (20, 49, 50))
                self.intersection = SocialForceIntersection((bike1, bike2, bike3), activate_sumo_cosimulation=False)
                Scenario.__init__(self, self._step_func, verbose=False)

            def _step_func(self):
                self.intersection.step()

        scn = DemoScenario()
        scn.run(7)
        g = golden("balancingrider")
        S = g["stddemo_S"]
        bikes = scn.intersection.vehicles
        got = np.array([v.s for v in bikes])
        assert got.shape == (3, 8) and all(v.i == 700 for v in bikes)
        np.testing.assert_allclose(got[:, :2], S[-1][:, :2], rtol=0, atol=2e-4)
        np.testing.assert_allclose(got[:, 2:], S[-1][:, 2:], rtol=0, atol=2e-3)
        for k, v in enumerate(bikes):
            np.testing.assert_allclose(v.traj[:2, 10:701:10].T, S[1:, k, :2], rtol=0, atol=2e-4)      # vehicle.traj history
            np.testing.assert_allclose(v.force, g["stddemo_F"][k], rtol=0, atol=2e-3)
    finally:
        sys.path.remove(os.path.join(root, "compat"))
        for k in [k for k in sys.modules if k == "cyclistsocialforce" or k.startswith("cyclistsocialforce.")]:
            del sys.modules[k]


def test_uncontrolled_vehicle_in_an_intersection(golden):
    """UncontrolledVehicle through the drop-in classes (vehicle.py:920-988): cyclists, a car on a prescribed trajectory and a
    parked one in ONE SocialForceIntersection, constructed as the reference's users would - against the trajectories the
    reference produced (tests/golden/uncontrolled.npz)."""
    from cyclistsocialforce_amd.parameters import CarParameters
    from cyclistsocialforce_amd.vehicle import UncontrolledVehicle

    g = golden("uncontrolled")
    s0, off, dq, vdes = g["s0"], g["off"], g["dq"], g["vdes"]
    n = s0.shape[0]
    vs = []
    for k in range(n - 2):
        v = TwoDBicycle(tuple(s0[k, :5]), id=f"b{k}")
        v.params.v_desired_default = float(vdes[k])
        rows = dq[off[k] + 1:off[k + 1]]
        v.setDestinations(rows[:, 0], rows[:, 1])
        vs.append(v)
    script = g["script_rows"].T
    car = UncontrolledVehicle(tuple(script[:, 0]), trajectory=script, id="car")
    parked = UncontrolledVehicle((3.0, -2.0, 0.5, 0.0), id="parked", params=CarParameters(hfov=float(g["parked_hfov"]), f_0=float(g["parked_f0"])))
    assert car.calcDestinationForce() == (0, 0) and car.traj.shape == script.shape
    with pytest.raises(ValueError):
        UncontrolledVehicle((0, 0, 0, 0), trajectory=np.zeros((3, 5)))
    ins = SocialForceIntersection(vs + [car, parked])
    S = g["S"]
    extent = max(np.ptp(S[..., 0]), np.ptp(S[..., 1]), 1.0)
    for k in range(1, S.shape[0]):
        for _ in range(10):
            ins.step()
        got = np.zeros((n, 6))
        for r, v in enumerate(ins.vehicles):
            got[r, : v.s.size] = v.s
        np.testing.assert_allclose(got[:, :2], S[k][:, :2], rtol=0, atol=1e-4 * extent, err_msg=f"sample {k}")
        np.testing.assert_allclose(got[n - 2:, :4], S[k][n - 2:, :4], rtol=0, atol=1e-12)
    assert car.traj.shape == script.shape and car.force == (0.0, 0.0)


def test_uncontrolled_vehicle_trajectory_rewritten_while_running():
    """The reference reads car.traj[:, i] on every step (vehicle.py:964-979) and documents external control by writing
    `traj` during the run: a trajectory rewritten after the car has joined reaches the engine with the next step."""
    from cyclistsocialforce_amd.vehicle import UncontrolledVehicle

    T = 60
    script = np.stack([np.linspace(0.0, 6.0, T), np.full(T, 2.0), np.zeros(T), np.full(T, 1.0)])
    car = UncontrolledVehicle(tuple(script[:, 0]), trajectory=script.copy(), id="car")
    bike = TwoDBicycle((10.0, 0.0, np.pi / 2, 4.0, 0.0), id="b")
    bike.setDestinations([10.0, 10.0], [20.0, 40.0])
    ins = SocialForceIntersection([bike, car])
    for _ in range(5):
        ins.step()
    i = int(car.i)
    np.testing.assert_allclose(car.s[:4], car.traj[:4, i], rtol=0, atol=1e-12)
    car.traj[1, i + 1:] += 0.75                               # external control: the rest of the trajectory moves sideways
    for _ in range(7):
        ins.step()
    j = int(car.i)
    assert j == i + 7
    np.testing.assert_allclose(car.s[:4], car.traj[:4, j], rtol=0, atol=1e-12)
    assert abs(car.s[1] - 2.75) < 1e-12


def test_custom_force_hooks_inside_a_population_against_the_reference(golden):
    """vehicle.py:194-204, 250-299: `rep_force_func` / `dest_force_func` of single vehicles inside a SocialForceIntersection.  The
    literal reference ran seven PlanarPointBicycles of which one exerts 2.5 x its class's field, one follows its own destination force
    and one does both (tests/golden/make_golden.py: gen_hooks); the mirror forms such a population's forces on the host from the
    engine's pieces (csf_dest_force, csf_untracked, csf_pair_force, csf_apply_forces) and must land on the same trajectory."""
    from cyclistsocialforce_amd.intersection import SocialForceIntersection
    from cyclistsocialforce_amd.vehicle import PlanarPointBicycle

    g = golden("hooks")
    s0, vdes, off, dq, S, F = (g[k] for k in ("s0", "vdes", "off", "dq", "S", "F"))

    def strong_field(veh, x, y, psi):
        fx, fy = veh.class_field(x, y, psi)      # (the reference's hook calls TwoDBicycle.calcRepulsiveForce(veh, ...) unbound)
        return 2.5 * np.asarray(fx), 2.5 * np.asarray(fy)

    def constant_pull(veh):
        dx, dy = 30.0 - veh.s[0], 30.0 - veh.s[1]
        r = np.hypot(dx, dy)
        return 3.5 * dx / r + 0.4, 3.5 * dy / r - 0.2

    vs = []
    for a in range(s0.shape[0]):
        v = PlanarPointBicycle(tuple(s0[a]), id=str(a))
        v.params.v_desired_default = float(vdes[a])
        rows = dq[off[a] + 1:off[a + 1]]
        v.setDestinations(rows[:, 0], rows[:, 1], rows[:, 2])
        vs.append(v)
    for a in g["rep_hook"]:
        vs[int(a)].rep_force_func = strong_field
    ins = SocialForceIntersection(vs)
    for a in g["dest_hook"]:
        vs[int(a)].dest_force_func = constant_pull           # (assigned after the vehicle joined: the intersection is told)
    extent = max(np.ptp(S[..., 0]), np.ptp(S[..., 1]), 1.0)
    for k in range(1, S.shape[0]):
        for _ in range(10):
            ins.step()
        got = np.array([v.s for v in vs])
        np.testing.assert_allclose(got[:, :2], S[k][:, :2], rtol=0, atol=1e-4 * extent, err_msg=f"sample {k}")
        np.testing.assert_allclose(np.array([v.force for v in vs]), F[k - 1], rtol=0, atol=2e-3 * max(np.abs(F[k - 1]).max(), 1.0))
    assert [v.destpointer for v in vs] == list(g["ptr"]) and all(len(v.F) == 150 for v in vs)
    # the same population without hooks takes the engine's own tick and lands elsewhere (the hooks did act)
    assert np.abs(S[-1][:, :2] - S[0][:, :2]).max() > 1.0
