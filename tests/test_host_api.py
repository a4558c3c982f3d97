"""CPU-only tests of the host side: parameter objects, vehicle constructors, road geometry, helpers, the
scenario driver, the sharding arithmetic, and that libcsf_hip.so loads and exports every symbol that
include/csf.h declares (no compute calls — there is no GPU here)."""
import ctypes
import os
import re

import numpy as np
import pytest

from cyclistsocialforce_amd import _ffi, parallel, parameters, utils
from cyclistsocialforce_amd.engine import Engine, EngineError
from cyclistsocialforce_amd.intersection import (CurvedRoadSegment, RoadSegmentCollection,
                                                 SocialForceIntersection, StraightRoadSegment,
                                                 flatten_road_elements)
from cyclistsocialforce_amd.scenario import Scenario
from cyclistsocialforce_amd.vehicle import (Bicycle, InvertedPendulumBicycle, InvPendulumBicycle,
                                            PlanarPointBicycle, TwoDBicycle, Vehicle)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "csf.h")).read()
    declared = set(re.findall(r"\b(csf_[a-z_0-9]+)\s*\(", header))
    declared -= {"csf_engine", "csf_params"}
    assert declared == set(_ffi.SYMBOLS), declared ^ set(_ffi.SYMBOLS)
    lib = _ffi.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.csf_abi_version() == 9
    assert ctypes.sizeof(_ffi.Params) == 576


def test_engine_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(EngineError, match="no CPU fallback"):
        Engine(parameters.default_pod("twod"), 8)


def test_params_struct_matches_oracle_layout():
    """The test harness fills csf_params and the oracle's csfo_params from one table: layouts must agree."""
    from oracle import csf_oracle as orc

    assert ctypes.sizeof(orc.Params) == ctypes.sizeof(_ffi.Params)
    assert [f[0] for f in orc.Params._fields_] == [f[0] for f in _ffi.Params._fields_]
    for m in range(4):
        assert bytes(parameters.default_pod(m)) == bytes(orc.default_params(m)), m


def test_parameter_validation_follows_the_reference():
    p = parameters.InvPendulumBicycleParameters()
    assert p.l == 1.0 and p.l_1 == 0.5 and p.v_max_riding == [-1.0, 7.0] and p.hfov == np.pi * 2 / 3
    with pytest.raises(AttributeError):       # parameters.py:516-519
        p.t_s = 0.02
    with pytest.raises(TypeError):            # floats must be float
        p.v_desired_default = 5
    with pytest.raises(ValueError):
        p.v_desired_default = -1.0
    p.v_desired_default = 4.5                 # demoCSFstandalone.py:104
    with pytest.raises(ValueError):           # parameters.py:637-641: e_0 in ]e_1, 1]
        p.e_0 = 0.5
    p.f_0 = 3                                 # parameters.py:623: f_0 is cast
    assert p.f_0 == 3.0
    with pytest.raises(TypeError):
        parameters.BicycleParameters(a_max=(-1, 1))
    b = parameters.BicycleParameters()
    assert b.a_max == [-10.0, 10.0] and b.v_max_stop == 0.6 and b.l_2 == 0.5
    assert parameters.VehicleParameters().hfov == 2 * np.pi
    pp = parameters.PlanarPointBicycleParameters()
    assert pp.to_pod(_ffi.PLANARPOINT).k_psi == 2.0
    assert parameters.PlanarPointBicycleParameters(poles=[-3 + 0j]).to_pod(3).k_psi == 3.0
    r = parameters.RoadElementParameters(sigma=2.0, F_0=0.15)
    with pytest.raises(AttributeError):
        r.F_0 = 0.2
    pod = p.to_pod(_ffi.TWOD, 1)
    assert pod.traj_len == 3000 and pod.priority_rule == 1 and pod.a_max[0] == -3.0 and pod.m == 87.0


def test_vehicle_constructors_follow_the_reference():
    v = TwoDBicycle((-6, 0, 0, 5, 0, 0, 0, 0), id="a", saveForces=True)   # demo passes 8-tuples
    assert v.s.shape == (5,) and v.traj.shape == (5, 3000) and v.trajF.shape == (2, 3000)
    assert isinstance(v.params, parameters.InvPendulumBicycleParameters)
    assert TwoDBicycle.N_STATES == 5 and InvPendulumBicycle.N_STATES == 6 and PlanarPointBicycle.N_STATES == 4
    assert InvertedPendulumBicycle is InvPendulumBicycle
    assert InvPendulumBicycle((0, 0, 0, 5, 0, 0)).STATE_NAMES[-1] == "theta[rad]"
    with pytest.raises(ValueError):
        Bicycle((0, 0, 0, 5))                                             # vehicle.py:149-150
    with pytest.raises(AssertionError):
        Bicycle((0, 0, 0, 5, 0), id=3)
    with pytest.raises(AssertionError):
        Bicycle((0, 0, 0, 5, 0), route=["e1"])
    with pytest.raises(TypeError):
        Bicycle((0, 0, 0, 5, 0), params=parameters.VehicleParameters())   # vehicle.py:245-247
    with pytest.raises(NotImplementedError):
        Vehicle((0, 0, 0, 5))
    w = PlanarPointBicycle((1, 2, 7.0, 4))
    assert abs(w.s[2] - (7.0 - 2 * np.pi)) < 1e-15                        # limitAngle at construction
    np.testing.assert_array_equal(w.destqueue, [[1, 2, 0]])               # vehicle.py:183-185
    w.setDestinations((5, 9), (0, 1))
    assert w.destqueue.shape == (3, 3) and w.destpointer == 0             # appended (vehicle.py:646-647)
    w.setDestinations((7,), (7,), stop=(1,), reset=True)
    np.testing.assert_array_equal(w.destqueue, [[7, 7, 1]])
    assert w.isLastDest() and abs(w.getDestinationDistance() - np.hypot(6, 5)) < 1e-12


def test_intersection_host_bookkeeping():
    a, b = TwoDBicycle((0, 0, 0, 5, 0), id="a"), TwoDBicycle((1, 2, 0.5, 4, 0), id="b")
    ins = SocialForceIntersection((a, b))
    assert ins.n_bikes == 2 and ins.vehicleX.shape == (2, 1) and ins.vehicleTheta[1, 0] == 0.5
    assert ins.get_road_user_ids() == ["a", "b"] and ins.has_road_user("b")
    c = TwoDBicycle((3, 3, 0, 5, 0), id="c")
    ins.add_road_user(c)
    assert ins.n_bikes == 3
    ins.remove_road_users_by_id(["a"])
    assert ins.get_road_user_ids() == ["b", "c"] and ins.vehicleX[0, 0] == 1.0 and a._owner is None
    ins.remove_road_user(1)
    assert ins.get_road_user_ids() == ["b"]
    with pytest.raises(ValueError):                      # co-simulation needs the net (tests: test_sumo_seam_host_side)
        SocialForceIntersection((), activate_sumo_cosimulation=True)
    mixed = SocialForceIntersection((Bicycle((0, 0, 0, 5, 0)), PlanarPointBicycle((0, 0, 0, 5))))     # intersection.py:797-823:
    pods, cls = mixed._param_classes()                                    # any vehicle classes may share an intersection
    assert [p.model for p in pods] == [0, 3] and cls.tolist() == [0, 1]
    with pytest.raises(ValueError):
        SocialForceIntersection((), priority_rule="left")
    empty = SocialForceIntersection(())
    empty.step()                                                          # intersection.py:888, 896
    assert empty.hist_n_vecs == [0]


def test_road_geometry_matches_reference_vertices(golden):
    """StraightRoadSegment / CurvedRoadSegment (intersection.py:118-211) on scenarios/curve-scenario.py:63-81."""
    g = golden("road_edges")
    rp = parameters.RoadElementParameters(sigma=2.0, F_0=0.15)
    x0 = np.array((0, -20, np.pi / 2))
    s1 = StraightRoadSegment(x0, 5, 25, params=rp, ds=0.1)
    s2 = CurvedRoadSegment(s1.x1, 5, 10, np.pi / 2, "right", params=rp, ds=0.1)
    s3 = CurvedRoadSegment(s2.x1, 5, 10, np.pi / 2, "left", params=rp, ds=0.1)
    s4 = StraightRoadSegment(s3.x1, 5, 20, params=rp, ds=0.1)
    segs = RoadSegmentCollection((s1, s2, s3, s4))
    off, verts, F0, sg = flatten_road_elements([segs])
    np.testing.assert_array_equal(off, g["off"])
    np.testing.assert_allclose(verts, g["verts"], rtol=0, atol=1e-12)
    np.testing.assert_array_equal(F0, g["F0"])
    np.testing.assert_array_equal(sg, g["sigma"])
    np.testing.assert_allclose(np.array([s.x1 for s in segs.segs]), g["x1"], atol=1e-12)
    assert s1.params.F_0 == 0.05                                          # segment params reset (intersection.py:74)
    xs, ys = segs.get_destinations_from_segments()
    assert len(xs) == 4 and segs[1] is s2


def test_utils_against_golden(golden):
    g = golden("utils")
    np.testing.assert_allclose([utils.limitAngle(float(t)) for t in g["a"]], g["limitAngle"], atol=1e-15)
    np.testing.assert_allclose(utils.limitAngle(g["a"].copy()), g["limitAngle"], atol=1e-15)
    np.testing.assert_allclose(utils.angleDifference(g["a1"], g["a2"]), g["angleDifference"], atol=1e-15)
    lx, ly = utils.limitMagnitude(g["x"].copy(), g["y"].copy(), g["r"].copy())
    np.testing.assert_allclose(lx, g["lx"], rtol=1e-15)
    np.testing.assert_allclose(ly, g["ly"], rtol=1e-15)
    rho, phi = utils.cart2polar(np.array([1.0, 0.0, -1.0]), np.array([0.0, -2.0, 1e-30]))
    np.testing.assert_allclose(rho, [1, 2, 1]); np.testing.assert_allclose(phi, [0, -np.pi / 2, np.pi])
    assert utils.thresh(5.0, (-1.0, 2.0)) == 2.0


def test_scenario_driver(capsys, monkeypatch):
    """Constructor / run() / reset() semantics of scenario.py:59-122, 169-195, 226-228."""
    calls = []
    scn = Scenario(lambda: calls.append(1), t_s=0.01, t_r=0, verbose=False)
    scn.run(0.25)
    assert len(calls) == 25 and scn.i == 25 and scn.i_end == 25 and abs(scn.t - 0.25) < 1e-12
    assert len(scn.hist_run_time) == 25 and scn.ticks_per_second() > 0
    scn.run(0.30)                                    # run() continues to a later end time
    assert len(calls) == 30
    scn.run(0.10)                                    # an earlier one is already reached
    assert len(calls) == 30
    scn.reset()
    assert scn.i == 0 and scn.t == 0
    paced = Scenario(lambda: None, t_s=0.01, t_r=0.02, verbose=False)     # real-time pacing: >= t_r per tick
    import time
    t0 = time.time()
    paced.run(0.05)
    assert time.time() - t0 >= 5 * 0.02 * 0.95
    monkeypatch.setattr("builtins.input", lambda *_: "")                   # verbose waits for <Enter> and reports
    Scenario(lambda: None, t_s=0.01, t_r=0, verbose=True).run(0.03)
    out = capsys.readouterr().out
    assert "3 of 3" in out and "finished" in out
    with pytest.raises(NotImplementedError):
        Scenario(lambda: None, write_animation=True)
    with pytest.raises(TypeError):
        Scenario(None)


def test_parameter_sets_of_an_intersection():
    """Every vehicle owns its params object (vehicle.py:64-204).  The mirror groups the population into distinct
    csf_params (anything but v_desired_default, which is per road user anyway) for the engine's table
    (csf_set_param_classes); what the engine cannot hold is refused: another clock, more than 256 sets, custom hooks."""
    from cyclistsocialforce_amd import parameters as P
    from cyclistsocialforce_amd.intersection import SocialForceIntersection
    from cyclistsocialforce_amd.vehicle import TwoDBicycle

    a = TwoDBicycle((0, 0, 0, 5, 0), id="a")
    b = TwoDBicycle((0, 5, 0, 5, 0), id="b")
    b.params.v_desired_default = 4.0                 # the per-vehicle parameter: no set of its own
    c = TwoDBicycle((0, 9, 0, 5, 0), id="c")
    c.params.f_0 = 3.0
    d = TwoDBicycle((0, 12, 0, 5, 0), id="d", params=P.InvPendulumBicycleParameters(f_0=3.0))     # equal to c's by value
    s = TwoDBicycle((0, 15, 0, 5, 0), id="s", params=a.params)                                     # a's object
    h = TwoDBicycle((0, 18, 0, 5, 0), id="h", params=P.InvPendulumBicycleParameters(hfov=1.0, sigma_0=0.6))
    ins = SocialForceIntersection((a, b, c, d, s, h))
    pods, cls = ins._param_classes()
    assert cls.tolist() == [0, 0, 1, 1, 0, 2] and len(pods) == 3
    assert pods[1].f_0 == 3.0 and pods[2].hfov == 1.0 and pods[2].sigma_0 == 0.6 and pods[0].f_0 == 7.0
    b.params.e_1 = 0.6                               # a later assignment moves b into a set of its own
    pods, cls = ins._param_classes()
    assert cls.tolist() == [0, 1, 2, 2, 0, 3] and pods[1].e_1 == 0.6
    slow = TwoDBicycle((0, 21, 0, 5, 0), id="slow", params=P.InvPendulumBicycleParameters(t_s=0.02))
    with pytest.raises(NotImplementedError, match="share the clock"):
        ins._param_classes(ins.vehicles + [slow])
    many = [TwoDBicycle((k, 30, 0, 5, 0), id=f"m{k}", params=P.InvPendulumBicycleParameters(f_0=1.0 + 0.01 * k)) for k in range(257)]
    with pytest.raises(NotImplementedError, match="256"):
        ins._param_classes(many)
    # a vehicle with a custom force hook may join (vehicle.py:194-204): the intersection then forms its forces on the host
    hooked = TwoDBicycle((0, 15, 0, 5, 0), id="h2", dest_force_func=lambda v: (0.0, 0.0))
    assert not ins._hooked
    ins.add_road_user(hooked)
    assert ins._hooked and hooked._owner is ins
    late = SocialForceIntersection((TwoDBicycle((0, 0, 0, 5, 0), id="l"),))
    late.vehicles[0].rep_force_func = lambda v, x, y, psi: (0.0 * x, 0.0 * y)     # (assigned after joining: the intersection is told)
    assert late._hooked


def test_shard_bounds_cover_the_population():
    for n in (1, 3, 63, 64, 65, 1000, 16384, 262144, 1048576):
        for world in (1, 2, 4, 8):
            blocks = [parallel.shard_bounds(n, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            for (lo, hi), (lo2, _) in zip(blocks, blocks[1:]):
                assert hi == lo2 and lo <= hi
            if world > 1:
                size = parallel.shard_size(n, world)
                assert size % 64 == 0 and size * world >= n
                assert all(hi - lo <= size for lo, hi in blocks)
    assert parallel.shard_bounds(16384, 8, 3) == (6144, 8192)


def test_sumo_seam_host_side():
    """intersection.py:341-453, 458-539, 679-688 on duck-typed sumolib / traci stand-ins (tests/sumo_fakes.py): lane
    end points, internal lanes, the spline prototype of an arriving road user, the TraCI push-back convention."""
    from sumo_fakes import FakeNet, FakeTraci

    from cyclistsocialforce_amd.intersection import SocialForceIntersection
    from cyclistsocialforce_amd.utils import angleSFMtoSUMO, angleSUMOtoSFM
    from cyclistsocialforce_amd.vehicle import TwoDBicycle

    tr = FakeTraci()
    ins = SocialForceIntersection([], id="J", activate_sumo_cosimulation=True, net=FakeNet(), traci=tr)
    assert set(ins.inEdges) == {"W_in", "E_in", "S_in", "N_in"} and set(ins.outEdges) == {"W_out", "E_out", "S_out", "N_out"}
    (xi, yi), = ins.inEdges["W_in"]                       # the last two samples of the approach lane, towards the node
    assert xi.shape == (2,) and xi[0] < xi[1] <= -5 + 1e-9 and abs(xi[1] + 5.0) < 1e-9
    (xo, yo), = ins.outEdges["N_out"]                     # the first two samples of the exit lane
    assert abs(yo[0] - 5.0) < 1e-9 and yo[1] > yo[0]
    assert ins.internal_lane_ids == [":J_0_0", ":J_0_1"]
    with pytest.raises(ValueError, match="internal"):
        SocialForceIntersection([], id="J", activate_sumo_cosimulation=True, net=FakeNet(internal=False), traci=tr)
    with pytest.raises(ValueError):
        SocialForceIntersection([], id="J", activate_sumo_cosimulation=True, traci=tr)
    # an arrival from the west that leaves to the north: 5-point prototype, only the points ahead of it are kept
    np.random.seed(0)
    u = TwoDBicycle((-9.0, -1.6, 0.0, 5.0, 0.0), id="veh0", route=("W_in", "N_out"))
    ins.add_road_user(u)
    assert u.destpointer == 0 and 3 <= u.destqueue.shape[0] <= 5
    assert np.all(np.diff(np.hypot(u.destqueue[:, 0] - u.destqueue[-1, 0], u.destqueue[:, 1] - u.destqueue[-1, 1])) < 0)
    np.testing.assert_allclose(u.destqueue[-1, :2], [xo[1], yo[1]], atol=1e-9)
    with pytest.raises(AssertionError, match="unknown edge"):
        ins.add_road_user(TwoDBicycle((0, 0, 0, 5, 0), id="bad", route=("X_in", "N_out")))
    # arrivals / departures are read off the internal lanes
    tr.occupancy = {":J_0_0": ("veh0", "veh7"), ":J_0_1": ("veh9",)}
    entered, exited = ins.find_entered_exited_roadusers()
    assert sorted(entered) == ["veh7", "veh9"] and list(exited) == []
    tr.occupancy = {":J_0_0": ()}
    entered, exited = ins.find_entered_exited_roadusers()
    assert list(entered) == [] and list(exited) == ["veh0"]
    # SUMO's angle convention: degrees, clockwise from north (utils.py:114-121)
    assert angleSFMtoSUMO(0.0) == 90.0 and angleSFMtoSUMO(np.pi / 2) == 0.0 and abs(angleSFMtoSUMO(-np.pi / 2) - 180.0) < 1e-12
    for deg in (0.0, 45.0, 90.0, 200.0, 359.0):
        assert abs(angleSFMtoSUMO(angleSUMOtoSFM(deg)) - deg) < 1e-9
    ins.update_road_user_positions()                       # pushes (x, y, angle) of every road user: intersection.py:679-688
    assert tr.moves[-1][1:] == ("veh0", "", -1, -9.0, -1.6, 90.0, 6)


def sumo_script(g):
    """SUMO's side of the scripted loop of tests/golden/sumo_seam.npz (make_golden.py: sumo_script - the same recipe)."""
    import json

    cfg = json.loads(str(g["script"]))
    arms = {"W": (-1, 0), "E": (1, 0), "S": (0, -1), "N": (0, 1)}
    rng = np.random.default_rng(cfg["seed"])
    on, born, per_tick, specs = {}, 0, [], {}
    for tick in range(cfg["ticks"]):
        if tick % cfg["every"] == 0:
            a, b = rng.choice(list(arms), 2, replace=False)
            ax, ay = arms[a]
            vid = f"veh{born}"
            specs[vid] = dict(route=(a + "_in", b + "_out"), since=tick,
                              s=[ax * 9.0 + ay * 1.6, ay * 9.0 - ax * 1.6, float(np.arctan2(-ay, -ax)), 4.0, 0.0])
            on[vid] = specs[vid]
            born += 1
        for vid in [v for v, sp in on.items() if tick - sp["since"] >= cfg["stay"]]:
            del on[vid]
        per_tick.append(tuple(on))
    return per_tick, specs


def test_sumo_seam_against_the_reference(golden):
    """The host side of the SUMO seam against what the REFERENCE class computed from the same duck-typed net / traci
    (tests/golden/sumo_seam.npz, captured by make_golden.py: gen_sumo_seam): footprint, the end points of every approach
    and exit lane, the internal lanes, the entered / exited road users of every tick of a scripted occupancy, and the
    spline prototype every arrival is given across the junction (intersection.py:341-453, 458-539, 576-634)."""
    import json

    from sumo_fakes import FakeNet, FakeTraci

    from cyclistsocialforce_amd.intersection import SocialForceIntersection
    from cyclistsocialforce_amd.vehicle import TwoDBicycle

    g = golden("sumo_seam")
    tr = FakeTraci()
    ins = SocialForceIntersection([], id="J", activate_sumo_cosimulation=True, net=FakeNet(), traci=tr)
    np.testing.assert_array_equal(ins.shape_vertices, g["shape"])
    assert sorted(ins.inEdges) == list(g["edges_in"]) and sorted(ins.outEdges) == list(g["edges_out"])
    for k, e in enumerate(g["edges_in"]):
        for lane, (x, y) in enumerate(ins.inEdges[str(e)]):
            np.testing.assert_allclose(np.r_[x, y], g["lane_in"][k, lane], rtol=0, atol=1e-12)
    for k, e in enumerate(g["edges_out"]):
        for lane, (x, y) in enumerate(ins.outEdges[str(e)]):
            np.testing.assert_allclose(np.r_[x, y], g["lane_out"][k, lane], rtol=0, atol=1e-12)
    assert ins.internal_lane_ids == list(g["internal_lane_ids"])
    per_tick, specs = sumo_script(g)
    entered_ref, exited_ref = json.loads(str(g["entered"])), json.loads(str(g["exited"]))
    qids, qoff, qrows = list(g["queue_ids"]), g["queue_off"], g["queue_rows"]
    for tick, occ in enumerate(per_tick):                      # bookkeeping only: no tick is run (no GPU here)
        tr.occupancy = {":J_0_0": occ}
        entered, exited = ins.find_entered_exited_roadusers()
        assert [str(v) for v in entered] == entered_ref[tick] and [str(v) for v in exited] == exited_ref[tick], tick
        ins.remove_road_users_by_id(list(exited))
        for vid in entered:
            sp = specs[str(vid)]
            u = TwoDBicycle(tuple(sp["s"]), id=str(vid), route=sp["route"])
            np.random.seed(1000 + int(str(vid)[3:]))
            ins.add_road_user(u)
            k = qids.index(str(vid))
            np.testing.assert_allclose(u.destqueue, qrows[qoff[k]:qoff[k + 1]], rtol=0, atol=1e-12, err_msg=str(vid))
        assert len(ins.vehicles) == len(occ)
    assert len(qids) == 6


def test_visual_hook_drawings():
    """Vehicle.add_drawing / update_drawing / plot_states / plot_forces on matplotlib's Agg canvas
    (vehicle.py:695-917); no GPU involved - the drawings read the host mirror."""
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt

    from cyclistsocialforce_amd.vehicle import TwoDBicycle

    fig, ax = plt.subplots(1, 1)
    v = TwoDBicycle((1.0, 2.0, 0.0, 5.0, 0.0), id="a", saveForces=True)
    v.setDestinations((30.0, 60.0), (2.0, 2.0))
    v.add_drawing(ax)
    xy0 = v.drawing.body.get_xy().copy()
    assert np.allclose(xy0[0], [1.0 + 0.6 * 1.8, 2.0])                    # the nose points along psi = 0
    v.s[:3] = (4.0, 3.0, np.pi / 2)
    v.update_drawing(Fres=(0.0, 2.0))
    assert np.allclose(v.drawing.body.get_xy()[0], [4.0, 3.0 + 0.6 * 1.8])
    assert np.allclose(v.drawing.force.get_ydata(), [3.0, 3.0 + 0.3 * 2.0])
    assert list(v.drawing.destinations.get_xdata()) == [1.0, 30.0, 60.0]
    v.drawing.set_animated(True)
    assert all(a.get_animated() for a in v.drawing.artists())
    v.traj[:, 1] = v.s; v.trajF[:, 1] = (0.0, 2.0); v.i = 1
    axs = v.plot_states(t_end=1.0)
    assert len(axs) == 5 and len(axs[0].lines) == 1
    axf = v.plot_forces(components_to_plot=["magnitude", "direction"])
    assert len(axf) == 2 and np.allclose(axf[0].lines[0].get_ydata(), [2.0])
    with pytest.raises(ValueError):
        TwoDBicycle((0, 0, 0, 5, 0)).plot_forces()
    plt.close("all")


def test_reference_package_name_resolves_to_the_mirror():
    """compat/cyclistsocialforce: the import lines of the reference's scripts (demoCSFstandalone.py:23-25) load the mirror
    classes, BalancingRiderBicycle (vehicle.py:1953-1990) among them since round 5."""
    import importlib
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "compat"))
    try:
        for mod in ("vehicle", "intersection", "scenario", "parameters", "utils", "vizualisation"):
            alias = importlib.import_module(f"cyclistsocialforce.{mod}")
            impl = importlib.import_module(f"cyclistsocialforce_amd.{mod}")
            names = [k for k in vars(impl) if not k.startswith("_")]
            assert names and all(getattr(alias, k) is getattr(impl, k) for k in names), mod
        from cyclistsocialforce.vehicle import BalancingRiderBicycle, Bicycle, InvPendulumBicycle, PlanarPointBicycle  # noqa: F401
        b = BalancingRiderBicycle((0, 0, 0, 5, 0, 0, 0, 0, 99), id="a")      # (longer start states are cut: vehicle.py:149-152)
        assert b.N_STATES == 8 and b.s.shape == (8,) and b.params.l == b.params.bp_params["w"] and b.traj.shape == (8, 3000)
    finally:
        sys.path.remove(os.path.join(root, "compat"))
        for k in [k for k in sys.modules if k == "cyclistsocialforce" or k.startswith("cyclistsocialforce.")]:
            del sys.modules[k]


def test_balancingrider_parameters_follow_the_reference(golden):
    """parameters.py:1214-1411 in the host mirror: the Whipple-Carvallo matrices of the formulas of Meijaard et al. (2007)
    against the paper's benchmark (typed in from the paper: tests/golden/make_golden_balancingrider.py), the state-space matrices
    and the poles over speed against what the reference's objects returned there, and the POD the engine takes against the
    oracle's (which reproduces the reference's trajectories to 1e-14, tests/test_oracle_golden.py)."""
    from cyclistsocialforce_amd import parameters
    from oracle import csf_oracle as orc

    g = golden("balancingrider")
    bench = dict(zip([str(k) for k in g["wc_benchmark_names"]], [float(v) for v in g["wc_benchmark_params"]]))
    M, C1, K0, K2 = parameters.whipple_carvallo_matrices(bench)
    for got, name in ((M, "M"), (C1, "C1"), (K0, "K0"), (K2, "K2")):
        np.testing.assert_allclose(got, g[f"wc_benchmark_{name}"], rtol=0, atol=2e-13, err_msg=name)
    p = parameters.BalancingRiderBicycleParameters()
    assert dict(zip([str(k) for k in g["default_bike_names"]], g["default_bike_params"])) == p.bp_params
    A, B = p.get_state_space_matrices(4.0)
    np.testing.assert_allclose(A, g["ss_A_v4"][:4, :4], rtol=0, atol=1e-12)
    np.testing.assert_allclose(B, g["ss_B_v4"][:4], rtol=0, atol=1e-12)
    for v, want in zip(g["poles_BR1_v"], g["poles_BR1"]):
        p.update_control_params(v)
        np.testing.assert_allclose(p.poles, want, rtol=0, atol=1e-12)
    pod, ref = p.to_pod(6), orc.default_params("balancingrider")
    for name in ("br_minv_k0g", "br_minv_k2", "br_minv_c1", "br_minv_steer", "br_yaw", "br_pole_fun", "v_max_riding", "a_max"):
        np.testing.assert_allclose(list(getattr(pod, name)), list(getattr(ref, name)), rtol=1e-13, atol=1e-14, err_msg=name)
    assert pod.l == ref.l == 1.113 and pod.k_p_v == ref.k_p_v and pod.br_mode == 0 and pod.model == 6
    # fixed poles / fixed gains (parameters.py:1309-1316; dynamics.py:382-391, 604-605)
    fixed = parameters.BalancingRiderBicycleParameters(poles=(-8.0, -1 + 2j, -1 - 2j, -2 + 6j, -2 - 6j)).to_pod(6)
    assert list(fixed.br_pole_fun) == [-8.0, 0.0, -1.0, 0.0, 2.0, 0.0, -2.0, 0.0, 6.0, 0.0] and fixed.br_mode == 0
    gained = parameters.BalancingRiderBicycleParameters(gains=(-10.0, 2.0, -7.0, -0.1, -7.0)).to_pod(6)
    assert gained.br_mode == 2 and list(gained.br_gains) == [-10.0, 2.0, -7.0, -0.1, -7.0]
    assert parameters.BalancingRiderBicycleParameters(stochastic_control_behavior=True).polesampler is not None   # (round 6: built)


def test_pole_sampler_draws_what_the_reference_draws():
    """stochastic_control_behavior (parameters.py:1380-1396): polemodel.PoleSampler against 60 consecutive
    `PoleModel.sample_poles(1, X_given=v)` of the literal reference per model file, both on NumPy's global generator behind
    np.random.seed(1234) (tests/golden/make_golden_balancingrider.py stochastic)."""
    from cyclistsocialforce_amd import polemodel

    g = np.load(os.path.join(ROOT, "tests", "golden", "balancingrider_stochastic.npz"))
    for tag, fname in (("BR0", "BR0_ImRe5GivenV_pole-model-params.yaml"), ("BR1", "BR1_ImRe5GivenV_pole-model-params.yaml")):
        smp = polemodel.PoleSampler(fname)
        np.random.seed(1234)
        got = np.array([smp.sample(v) for v in g[f"draws_{tag}_v"]])
        ref = g[f"draws_{tag}_poles"]
        np.testing.assert_allclose(got, ref, rtol=1e-11, atol=1e-12)
        assert (got.real <= 0).all() and np.allclose(got[:, 1], np.conj(got[:, 2])) and np.allclose(got[:, 3], np.conj(got[:, 4]))
        # a stream of one's own: a RandomState gives the same numbers as the global generator it mirrors, a Generator others
        a = np.array([smp.sample(3.0, rng=np.random.RandomState(5)) for _ in range(2)])
        np.random.seed(5)
        b = smp.sample(3.0)
        assert np.array_equal(a[0], a[1]) and np.array_equal(a[0], b)
        assert not np.array_equal(smp.sample(3.0, rng=np.random.default_rng(5)), b)


def test_stochastic_rider_parameters():
    """parameters.py:1380-1396: the first draw at construction, a new one only once the speed has moved 0.8333 m/s since the last;
    the POD carries the drawn poles as constants"""
    from cyclistsocialforce_amd import parameters as P
    from cyclistsocialforce_amd.vehicle import BalancingRiderBicycle

    np.random.seed(3)
    prm = P.BalancingRiderBicycleParameters(stochastic_control_behavior=True)
    assert prm.poles is None
    with pytest.raises(ValueError, match="no poles drawn"):
        prm.to_pod(6)
    b = BalancingRiderBicycle((0, 0, 0, 4.0, 0, 0, 0, 0), params=prm)
    first = np.array(prm.poles)
    assert first.shape == (5,) and prm.v_last_update == 4.0 and b.params is prm
    prm.update_control_params(4.5)
    assert np.array_equal(prm.poles, first) and prm.v_last_update == 4.0            # within the threshold: the poles stay
    prm.update_control_params(4.9)
    assert not np.array_equal(prm.poles, first) and prm.v_last_update == 4.9
    pod = prm.to_pod(6)
    fun = np.array(pod.br_pole_fun).reshape(5, 2)
    pl = np.asarray(prm.poles)
    assert pod.br_mode == 0 and (fun[:, 1] == 0).all() and np.allclose(fun[:, 0], [pl[0].real, pl[1].real, abs(pl[1].imag), pl[3].real, abs(pl[3].imag)])
