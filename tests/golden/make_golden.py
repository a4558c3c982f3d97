#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING the reference.

This script is test infrastructure.  It imports the literal reference package
from /root/reference/src (by path; nothing is copied), feeds it seeded inputs and
stores inputs + outputs as small fp64 ``.npz`` files.  It only runs in the build
container: the GPU box has no /root/reference and only ever reads the ``.npz``.

What the import needs:

* ``sys.modules`` stand-ins for packages that are absent here and only touched at
  import time (pypaperutils, mypyutils, bicycleparameters, controlbehavior) —
  none of them is on the hot path;
* python-control is absent: a ``control`` shim provides ``ss``, ``forced_response`` (block matrix
  exponential of a first-order hold, which is the published formula python-control uses
  for continuous LTI systems), ``place`` (scipy.signal.place_poles) and ``ctrb``.  The
  InvPendulum and PlanarBicycle vectors are therefore "parity unpinned against
  python-control"; gen_invpend_yawstep checks the shim's time stepping against two
  SciPy-only integrations of the same matrices;
* ``TwoDBicycle.__init__`` is broken at reference HEAD (vehicle.py:1359 passes
  keyword-only arguments positionally).  ``_TwoD`` / ``_InvPend`` below repair only
  that call; every *method* that runs is the reference's own;
* a no-op drawing on every vehicle (intersection.py:881-885 would otherwise
  create matplotlib artists on the first tick).

Usage:  python tests/golden/make_golden.py            (writes tests/golden/*.npz)
"""
import os
import sys
import types

import numpy as np

REF_SRC = "/root/reference/src"
OUT = os.path.dirname(os.path.abspath(__file__))


# ----------------------------------------------------------------------------
# import scaffolding
# ----------------------------------------------------------------------------
def _install_stubs():
    import matplotlib

    matplotlib.use("Agg")

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Colors:
        def __init__(self, *a, **k):
            pass

        def get(self, *a, **k):
            return (0.0, 0.0, 0.0)

        def __getattr__(self, k):
            return lambda *a, **kw: (0.0, 0.0, 0.0)

    mod("pypaperutils")
    mod("pypaperutils.design", TUDcolors=_Colors)
    mod("mypyutils")
    mod("mypyutils.io", read_yaml=lambda *a, **k: {})
    mod("mypyutils.misc")
    mod("bicycleparameters")
    mod("bicycleparameters.parameter_dicts", meijaard2007_browser_jason={})
    mod("bicycleparameters.parameter_sets", Meijaard2007ParameterSet=object)
    mod("bicycleparameters.models", Meijaard2007Model=object)
    mod("cyclistsocialforce.controlbehavior", PoleModel=object)

    # --- python-control shim -------------------------------------------------
    from scipy.linalg import expm

    class _SS:
        def __init__(self, A, B, C, D):
            self.A = np.atleast_2d(np.asarray(A, dtype=float))
            B = np.asarray(B, dtype=float)
            self.B = B.reshape(self.A.shape[0], -1)
            self.C = np.atleast_2d(np.asarray(C, dtype=float))
            self.D = np.atleast_2d(np.asarray(D, dtype=float))

    def forced_response(sys_, T=None, U=0.0, X0=0.0, return_x=False, squeeze=None):
        A = np.atleast_2d(np.asarray(sys_.A, dtype=float))
        n = A.shape[0]
        B = np.asarray(sys_.B, dtype=float).reshape(n, -1)
        C = np.atleast_2d(np.asarray(sys_.C, dtype=float))
        D = np.atleast_2d(np.asarray(sys_.D, dtype=float))
        m = B.shape[1]
        T = np.asarray(T, dtype=float)
        U = np.asarray(U, dtype=float).reshape(m, -1)
        x = np.zeros((n, T.size))
        x[:, 0] = np.asarray(X0, dtype=float).reshape(n)
        dt = T[1] - T[0]
        M = np.zeros((n + 2 * m, n + 2 * m))
        M[:n, :n] = A * dt
        M[:n, n:n + m] = B * dt
        M[n:n + m, n + m:] = np.eye(m)
        E = expm(M)
        Ad = E[:n, :n]
        Bd1 = E[:n, n + m:]
        Bd0 = E[:n, n:n + m] - Bd1
        for i in range(1, T.size):
            x[:, i] = Ad @ x[:, i - 1] + Bd0 @ U[:, i - 1] + Bd1 @ U[:, i]
        y = C @ x + D @ U
        return T, y, x

    class _Response:
        """what python-control's forced_response returns: unpacks as (t, y[, x]) and has .time / .outputs / .states"""

        def __init__(self, t, y, x, with_states):
            self.time, self.outputs, self.states, self._with_states = t, y, x, with_states

        def __iter__(self):
            return iter((self.time, self.outputs, self.states) if self._with_states else (self.time, self.outputs))

    def forced_response_obj(sys_, T=None, U=0.0, X0=0.0, return_x=False, squeeze=None):
        n = np.atleast_2d(np.asarray(sys_.A, dtype=float)).shape[0]
        if np.ndim(X0) == 0:
            X0 = np.full(n, float(X0))
        t, y, x = forced_response(sys_, T=T, U=U, X0=X0)
        return _Response(t, y, x, return_x)

    def place(A, B, poles):          # SISO pole placement has one solution: scipy.signal.place_poles finds it
        from scipy.signal import place_poles

        return place_poles(np.asarray(A, dtype=float), np.asarray(B, dtype=float), np.asarray(poles)).gain_matrix

    def ctrb(A, B):
        A = np.asarray(A, dtype=float)
        B = np.asarray(B, dtype=float).reshape(A.shape[0], -1)
        return np.hstack([np.linalg.matrix_power(A, k) @ B for k in range(A.shape[0])])

    mod("control", ss=_SS, StateSpace=_SS, forced_response=forced_response_obj, place=place, ctrb=ctrb)


_install_stubs()
sys.path.insert(0, REF_SRC)

from cyclistsocialforce import vehicle as rv  # noqa: E402
from cyclistsocialforce import intersection as ri  # noqa: E402
from cyclistsocialforce import parameters as rp  # noqa: E402
from cyclistsocialforce import utils as ru  # noqa: E402
from cyclistsocialforce.dynamics import PIDcontroller  # noqa: E402


class _NoDrawing:
    def update(self, *a, **k):
        pass

    def set_animated(self, *a, **k):
        pass


class _TwoD(rv.TwoDBicycle):
    """Constructor repair only (vehicle.py:1353-1363)."""

    def __init__(self, s0, id="unknown", route=(), saveForces=False, params=None):
        if params is None:
            params = rp.InvPendulumBicycleParameters()
        rv.Bicycle.__init__(self, s0, id=id, route=route, saveForces=saveForces, params=params)
        self.speed_controller = PIDcontroller(self.params.k_p_v, 0, 0, self.params.t_s, isangle=False)


class _InvPend(rv.InvPendulumBicycle):
    """Constructor repair only (vehicle.py:1720-1736)."""

    def __init__(self, s0, **kwargs):
        kwargs = self.verify_params_class(kwargs)
        _TwoD.__init__(self, s0, **kwargs)
        self.init_dynamics_statespace()
        self.x = np.array([[self.s[4]], [0], [self.s[5]], [0], [self.s[2]]])
        self.zrid = np.zeros((2), dtype=bool)
        if s0[3] < self.params.v_max_walk:
            self.zrid[1] = True
        else:
            self.zrid[0] = True


MODELS = {
    "bicycle": rv.Bicycle,
    "twod": _TwoD,
    "invpend": _InvPend,
    "planarpoint": rv.PlanarPointBicycle,
    "planarbike": rv.PlanarBicycle,
}
NSTATES = {"bicycle": 5, "twod": 5, "invpend": 6, "planarpoint": 4, "planarbike": 5}


def make_vehicle(model, s0, vdes=None, **kw):
    v = MODELS[model](tuple(s0), **kw)
    v.drawing = _NoDrawing()
    if vdes is not None:
        v.params.v_desired_default = float(vdes)
    return v


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print(f"wrote {path}  ({os.path.getsize(path)} B)")


# ----------------------------------------------------------------------------
# (1) pair fields  — vehicle.py:1560-1648 (A2) and 1054-1147 (A2')
# ----------------------------------------------------------------------------
def gen_pair_fields():
    rng = np.random.default_rng(101)
    # survey's known-answer case first, then random + structured cases
    src = [(0.0, 0.0, 0.0)]
    rx = [np.array([5.0, -3.0, 2.0, 0.5])]
    ry = [np.array([0.5, 4.0, -7.0, 0.0])]
    rpsi = [np.array([0.0, 1.0, -2.5, np.pi])]
    for _ in range(24):
        m = 48
        src.append((rng.uniform(-20, 20), rng.uniform(-20, 20), rng.uniform(-np.pi, np.pi)))
        rho = np.exp(rng.uniform(np.log(0.3), np.log(60.0), m))
        az = rng.uniform(-np.pi, np.pi, m)
        rx.append(src[-1][0] + rho * np.cos(az))
        ry.append(src[-1][1] + rho * np.sin(az))
        rpsi.append(rng.uniform(-np.pi, np.pi, m))
    # exactly ahead / behind / abeam of the source, parallel and perpendicular headings
    psi0 = 0.7
    src.append((1.0, -2.0, psi0))
    ang = np.array([0.0, np.pi, np.pi / 2, -np.pi / 2, 1e-9, -1e-9, np.pi - 1e-9, -np.pi + 1e-9])
    rho = np.array([0.5, 1.0, 2.0, 5.0, 10.0, 20.0, 40.0, 80.0])
    rx.append(1.0 + rho * np.cos(ang + psi0))
    ry.append(-2.0 + rho * np.sin(ang + psi0))
    rpsi.append(np.array([psi0, psi0 + np.pi / 2, psi0 - np.pi / 2, psi0 + np.pi, 0.0, 1.0, 2.0, 3.0]))

    twod = make_vehicle("twod", (0, 0, 0, 5, 0))
    bike = make_vehicle("bicycle", (0, 0, 0, 5, 0))
    S, X, Y, P, V = [], [], [], [], []
    F2x, F2y, F1x, F1y = [], [], [], []
    for k, (s, x, y, p) in enumerate(zip(src, rx, ry, rpsi)):
        v = [5.0, 0.0, 1e-3, 2.0, 9.5, 10.0][k % 6]
        twod.s[:3] = s
        bike.s[:3] = s
        bike.s[3] = v
        fx, fy = rv.TwoDBicycle.calcRepulsiveForce(twod, x.copy(), y.copy(), p.copy())
        gx, gy = rv.Bicycle.calcRepulsiveForce(bike, x.copy(), y.copy())
        for i in range(x.size):
            S.append(s)
            V.append(v)
        X.append(x); Y.append(y); P.append(p)
        F2x.append(fx); F2y.append(fy); F1x.append(gx); F1y.append(gy)
    cat = np.concatenate
    save(
        "pair_fields",
        src=np.array(S), src_v=np.array(V), x=cat(X), y=cat(Y), psi=cat(P),
        twod_fx=cat(F2x), twod_fy=cat(F2y), bicycle_fx=cat(F1x), bicycle_fy=cat(F1y),
    )


# ----------------------------------------------------------------------------
# (2)+(3) masks and force totals — intersection.py:690-745, 747-864
# ----------------------------------------------------------------------------
def random_population(rng, model, n, box, with_dests=True):
    ns = NSTATES[model]
    vs = []
    for k in range(n):
        s0 = np.zeros(ns)
        s0[0] = rng.uniform(0, box)
        s0[1] = rng.uniform(0, box)
        s0[2] = rng.uniform(-np.pi, np.pi)
        s0[3] = rng.uniform(3, 6)
        v = make_vehicle(model, s0, vdes=rng.uniform(4, 5.5), id=str(k))
        if with_dests:
            d = np.array([15.0, 29.0, 30.0])
            v.setDestinations(s0[0] + d * np.cos(s0[2]), s0[1] + d * np.sin(s0[2]))
        vs.append(v)
    return vs


def pop_arrays(vs):
    s0 = np.array([v.s.copy() for v in vs])
    vdes = np.array([v.params.v_desired_default for v in vs])
    off = np.cumsum([0] + [v.destqueue.shape[0] for v in vs])
    dq = np.vstack([v.destqueue for v in vs])
    return s0, vdes, off, dq


def gen_masks_and_totals():
    rng = np.random.default_rng(202)
    out = {}
    for tag, (n, box, rule) in {
        "n2": (2, 6.0, "unregulated"),
        "n3": (3, 8.0, "unregulated"),
        "n16": (16, 30.0, "unregulated"),
        "n32": (32, 40.0, "unregulated"),
        "n16_p2r": (16, 30.0, "p2r"),
    }.items():
        vs = random_population(rng, "twod", n, box)
        ins = ri.SocialForceIntersection(vs, priority_rule=rule)
        s0, vdes, off, dq = pop_arrays(vs)
        U = ins.get_untracked_foes()
        # warm one tick so that i != 0 and the spline branch of the destination force is used
        ins.step()
        s1 = np.array([v.s.copy() for v in vs])
        U1 = ins.get_untracked_foes()
        Fx, Fy = ins.calc_forces()
        out.update({
            f"{tag}_s0": s0, f"{tag}_vdes": vdes, f"{tag}_off": off, f"{tag}_dq": dq,
            f"{tag}_untracked0": U, f"{tag}_s1": s1, f"{tag}_untracked1": U1,
            f"{tag}_Fx1": Fx, f"{tag}_Fy1": Fy, f"{tag}_p2r": np.array(rule == "p2r"),
        })
    save("masks_totals", **out)


# ----------------------------------------------------------------------------
# (4) control + move single steps — vehicle.py:1218-1272
# ----------------------------------------------------------------------------
def gen_control_move():
    rng = np.random.default_rng(303)
    S, F, D, O, M = [], [], [], [], []
    # survey KAT first
    cases = [("twod", (1, 2, .3, 4, .05), ((50, 99, 100), (20, 40, 41)), (3.0, 2.0))]
    for k in range(200):
        model = ("twod", "bicycle")[k % 2]
        s = (rng.uniform(-5, 5), rng.uniform(-5, 5), rng.uniform(-np.pi, np.pi),
             rng.uniform(-1, 8), rng.uniform(-1.4, 1.4))
        far = rng.uniform(0.5, 60)
        th = rng.uniform(-np.pi, np.pi)
        if k % 5 == 0:  # single, near destination: exercises the final-approach taper
            dx = (s[0] + far * 0.05 * np.cos(th),)
            dy = (s[1] + far * 0.05 * np.sin(th),)
        else:
            dx = tuple(s[0] + far * np.cos(th) * np.array([1, 2, 2.1]))
            dy = tuple(s[1] + far * np.sin(th) * np.array([1, 2, 2.1]))
        f = (rng.normal(0, 4), rng.normal(0, 4))
        if k % 17 == 0:
            f = (-abs(f[0]), 0.0)  # theta = pi exactly
        cases.append((model, s, (dx, dy), f))
    for model, s, (dx, dy), f in cases:
        v = make_vehicle(model, s)
        v.setDestinations(dx, dy, reset=True)
        v.dest = np.array(v.dest, dtype=float)
        v.step(f[0], f[1])
        S.append(s); F.append(f); O.append(v.s.copy()); M.append(model == "bicycle")
        D.append((dx[0], dy[0], float(len(dx) == 1)))
    save("control_move", s=np.array(S), F=np.array(F), dest=np.array(D),
         is_bicycle=np.array(M), s_next=np.array(O))


# ----------------------------------------------------------------------------
# (5)+(6) destination force, queue advance, nav machine — vehicle.py:354-457, 545-594, 1416-1558
# driven through single-agent closed-loop runs so that every internal state is reached naturally
# ----------------------------------------------------------------------------
def gen_dest_force():
    rng = np.random.default_rng(404)
    runs = {}
    specs = [
        # (tag, model, s0, dests x, dests y, stop flags, ticks)
        ("demo_a", "twod", (-6, 0, 0, 5, 0), (35, 64, 65), (0, 0, 0), None, 700),
        ("turn", "twod", (0, 0, 0.2, 4, 0), (10, 20, 25, 25, 25, 30), (0, 5, 15, 25, 40, 60), None, 1500),
        ("stop_last", "twod", (0, 0, 0, 5, 0), (15, 30, 40), (0, 3, 5), (0, 0, 1), 1600),
        ("stop_mid", "twod", (0, 0, 1.0, 3, 0), (5, 12, 20, 30), (8, 16, 20, 22), (0, 1, 0, 0), 900),
        ("pp_turn", "planarpoint", (0, 0, 0.2, 4), (10, 20, 25, 25), (0, 5, 15, 40), None, 900),
    ]
    for tag, model, s0, dx, dy, stop, ticks in specs:
        v = make_vehicle(model, s0, vdes=4.5 if tag == "demo_a" else None)
        v.setDestinations(dx, dy, stop=stop)
        dq = v.destqueue.copy()
        S = np.zeros((ticks + 1, len(v.s)))
        Fd = np.zeros((ticks, 2))
        ptr = np.zeros(ticks, dtype=np.int64)
        zn = np.zeros((ticks, 3), dtype=bool)
        S[0] = v.s
        for t in range(ticks):
            fx, fy = v.calcDestinationForce()
            Fd[t] = (fx, fy)
            ptr[t] = v.destpointer
            zn[t] = v.znav
            v.step(fx, fy)
            S[t + 1] = v.s
        runs.update({f"{tag}_s": S, f"{tag}_Fdest": Fd, f"{tag}_ptr": ptr, f"{tag}_znav": zn,
                     f"{tag}_dq": dq, f"{tag}_vdes": np.array(v.params.v_desired_default),
                     f"{tag}_model": np.array(model)})
    save("dest_force_runs", **runs)


# ----------------------------------------------------------------------------
# (7) PlanarPoint single steps — dynamics.py:996-1079
# ----------------------------------------------------------------------------
def gen_planarpoint_steps():
    rng = np.random.default_rng(505)
    S, F, O, X = [], [], [], []
    for k in range(120):
        s = (rng.uniform(-5, 5), rng.uniform(-5, 5), rng.uniform(-np.pi, np.pi), rng.uniform(0, 9))
        v = make_vehicle("planarpoint", s)
        f = (rng.normal(0, 4), rng.normal(0, 4))
        v.step(f[0], f[1])
        f2 = (rng.normal(0, 4), rng.normal(0, 4))
        s1 = v.s.copy()
        x1 = np.array(v.dynamics.x, dtype=float).copy()
        v.step(f2[0], f2[1])
        S.append(np.r_[s, s1]); F.append(np.r_[f, f2]); O.append(v.s.copy()); X.append(np.r_[x1, v.dynamics.x])
    save("planarpoint_steps", s01=np.array(S), F01=np.array(F), s2=np.array(O), x12=np.array(X))


# ----------------------------------------------------------------------------
# (8) road edges — intersection.py:118-242 ; scenarios/curve-scenario.py:63-81 geometry
# ----------------------------------------------------------------------------
def curve_road():
    roadparams = rp.RoadElementParameters(sigma=2.0, F_0=0.15)
    x0 = np.array((0, -20, np.pi / 2))
    import io
    import contextlib

    with contextlib.redirect_stdout(io.StringIO()):
        seg1 = ri.StraightRoadSegment(x0, 5, 25, params=roadparams, ds=0.1)
        seg2 = ri.CurvedRoadSegment(seg1.x1, 5, 10, np.pi / 2, "right", params=roadparams, ds=0.1)
        seg3 = ri.CurvedRoadSegment(seg2.x1, 5, 10, np.pi / 2, "left", params=roadparams, ds=0.1)
        seg4 = ri.StraightRoadSegment(seg3.x1, 5, 20, params=roadparams, ds=0.1)
    return ri.RoadSegmentCollection((seg1, seg2, seg3, seg4))


def road_arrays(segs):
    verts, off, F0, sg = [], [0], [], []
    for seg in segs.segs:
        for e in seg.edges:
            verts.append(np.asarray(e.vertices, dtype=float))
            off.append(off[-1] + verts[-1].shape[0])
            F0.append(e.params.F_0)
            sg.append(e.params.sigma)
    return np.vstack(verts), np.array(off), np.array(F0), np.array(sg)


def gen_road():
    segs = curve_road()
    verts, off, F0, sg = road_arrays(segs)
    rng = np.random.default_rng(606)
    x = rng.uniform(-4, 24, (64, 1))
    y = rng.uniform(-22, 42, (64, 1))
    Fx, Fy = segs.calcRepulsiveForce(x, y)
    save("road_edges", verts=verts, off=off, F0=F0, sigma=sg, x=x.ravel(), y=y.ravel(),
         Fx=Fx.ravel(), Fy=Fy.ravel(),
         x1=np.array([s.x1 for s in segs.segs], dtype=float))


# ----------------------------------------------------------------------------
# (9) population trajectories through SocialForceIntersection.step — intersection.py:866-896
# ----------------------------------------------------------------------------
def run_population(vs, ticks, rule="unregulated", road=None, every=1):
    ins = ri.SocialForceIntersection(vs, priority_rule=rule, road_elements=[road] if road else [])
    n = len(vs)
    S = np.zeros((ticks // every + 1, n, len(vs[0].s)))
    Ftot = np.zeros((ticks // every, n, 2))
    S[0] = [v.s for v in vs]
    for t in range(ticks):
        ins.step()
        if (t + 1) % every == 0:
            S[(t + 1) // every] = [v.s for v in vs]
            Ftot[(t + 1) // every - 1] = [v.force for v in vs]
    return S, Ftot


def demo_bikes(model):
    a = make_vehicle(model, (-23 + 17, 0, 0, 5, 0, 0, 0, 0), vdes=4.5, id="a")
    b = make_vehicle(model, (0 + 15, -20, np.pi / 2, 5, 0, 0, 0, 0), vdes=5.0, id="b")
    c = make_vehicle(model, (-2 + 15, -20, np.pi / 2, 5, 0, 0, 0, 0), vdes=5.0, id="c")
    a.setDestinations((35, 64, 65), (0, 0, 0))
    b.setDestinations((15, 15, 15), (20, 49, 50))
    c.setDestinations((13, 13, 13), (20, 49, 50))
    return [a, b, c]


def gen_trajectories():
    out = {}
    for model in ("twod", "bicycle", "planarpoint", "invpend"):
        vs = demo_bikes(model)
        s0, vdes, off, dq = pop_arrays(vs)
        S, Ft = run_population(vs, 700, every=10)
        out.update({f"demo_{model}_s0": s0, f"demo_{model}_vdes": vdes, f"demo_{model}_off": off,
                    f"demo_{model}_dq": dq, f"demo_{model}_S": S, f"demo_{model}_F": Ft})
    rng = np.random.default_rng(707)
    for model, n, box, ticks in (("twod", 24, 30.0, 300), ("invpend", 12, 25.0, 200),
                                 ("bicycle", 16, 30.0, 200), ("planarpoint", 12, 25.0, 150)):
        vs = random_population(rng, model, n, box)
        s0, vdes, off, dq = pop_arrays(vs)
        S, Ft = run_population(vs, ticks, every=10)
        out.update({f"dense_{model}_s0": s0, f"dense_{model}_vdes": vdes, f"dense_{model}_off": off,
                    f"dense_{model}_dq": dq, f"dense_{model}_S": S, f"dense_{model}_F": Ft})
    # p2r rule
    vs = random_population(rng, "twod", 12, 25.0)
    s0, vdes, off, dq = pop_arrays(vs)
    S, Ft = run_population(vs, 150, rule="p2r", every=10)
    out.update({"p2r_twod_s0": s0, "p2r_twod_vdes": vdes, "p2r_twod_off": off, "p2r_twod_dq": dq,
                "p2r_twod_S": S, "p2r_twod_F": Ft})
    # planarpoint on the curve-scenario road (config-5 shape: agents + static obstacle forces)
    segs = curve_road()
    verts, roff, F0, sg = road_arrays(segs)
    vs = []
    for k, (x, y) in enumerate(((0.5, -19.0), (-0.8, -16.0), (1.0, -12.0))):
        v = make_vehicle("planarpoint", (x, y, np.pi / 2, 4.0), id=str(k))
        dxs, dys = segs.get_destinations_from_segments()
        v.setDestinations(dxs, dys)
        vs.append(v)
    s0, vdes, off, dq = pop_arrays(vs)
    S, Ft = run_population(vs, 300, road=segs, every=10)
    out.update({"road_pp_s0": s0, "road_pp_vdes": vdes, "road_pp_off": off, "road_pp_dq": dq,
                "road_pp_S": S, "road_pp_F": Ft, "road_pp_verts": verts, "road_pp_roff": roff,
                "road_pp_F0": F0, "road_pp_sigma": sg})
    # ring-buffer lap: one TwoD agent alone for 3100 ticks crosses i == 0 (vehicle.py:1407-1408, 1455)
    v = make_vehicle("twod", (0, 0, 0.3, 5, 0), id="lap")
    d = np.array([60.0, 120.0, 180.0, 240.0])
    v.setDestinations(d * np.cos(0.1), d * np.sin(0.1))
    s0, vdes, off, dq = pop_arrays([v])
    S, Ft = run_population([v], 3100, every=50)
    out.update({"lap_twod_s0": s0, "lap_twod_vdes": vdes, "lap_twod_off": off, "lap_twod_dq": dq,
                "lap_twod_S": S, "lap_twod_F": Ft})
    save("trajectories", **out)


# ----------------------------------------------------------------------------
# (9b) custom per-vehicle force hooks inside a population — vehicle.py:194-204, 250-299
# ----------------------------------------------------------------------------
def hook_strong_field(veh, x, y, psi):
    """a repulsive-force hook: 2.5 x the TwoD field the class would exert (vehicle.py:1560-1648, called unbound as the
    reference's own classes do: vehicle.py:2024)"""
    fx, fy = rv.TwoDBicycle.calcRepulsiveForce(veh, x, y, psi)
    return 2.5 * fx, 2.5 * fy


def hook_constant_pull(veh):
    """a destination-force hook that looks at the vehicle: a pull of 3.5 towards (30, 30) + a lateral bias"""
    dx, dy = 30.0 - veh.s[0], 30.0 - veh.s[1]
    r = np.hypot(dx, dy)
    return 3.5 * dx / r + 0.4, 3.5 * dy / r - 0.2


def gen_hooks():
    """Seven PlanarPointBicycles (the class whose forces go through Vehicle's hook dispatch, vehicle.py:2024-2025) in 18 m: vehicle 1
    exerts a stronger field, vehicle 3 follows its own destination force, vehicle 5 has both hooks; 150 ticks."""
    rng = np.random.default_rng(4242)
    vs = random_population(rng, "planarpoint", 7, 18.0)
    vs[1].rep_force_func = hook_strong_field
    vs[3].dest_force_func = hook_constant_pull
    vs[5].rep_force_func = hook_strong_field
    vs[5].dest_force_func = hook_constant_pull
    s0, vdes, off, dq = pop_arrays(vs)
    S, Ft = run_population(vs, 150, every=10)
    ptr = np.array([v.destpointer for v in vs])
    save("hooks", s0=s0, vdes=vdes, off=off, dq=dq, S=S, F=Ft, ptr=ptr, rep_hook=np.array([1, 5]), dest_hook=np.array([3, 5]))


# ----------------------------------------------------------------------------
# (10) helpers — utils.py:56-86, 124-227
# ----------------------------------------------------------------------------
def gen_utils():
    rng = np.random.default_rng(808)
    a = np.r_[rng.uniform(-10, 10, 200), [0, np.pi, -np.pi, 2 * np.pi, -2 * np.pi, 3 * np.pi, 1e-300]]
    la = np.array([ru.limitAngle(float(t)) for t in a])
    a1 = np.r_[rng.uniform(-np.pi, np.pi, 300), [0, np.pi, -np.pi, np.pi, 1.0, -3.0]]
    a2 = np.r_[rng.uniform(-np.pi, np.pi, 300), [0, -np.pi, np.pi, 0.0, 1.0 - np.pi, 3.0]]
    ad = np.array([ru.angleDifference(float(p), float(q)) for p, q in zip(a1, a2)])
    x = rng.normal(0, 3, 100); y = rng.normal(0, 3, 100); r = np.abs(rng.normal(0, 3, 100))
    lx, ly = ru.limitMagnitude(x.copy(), y.copy(), r.copy())
    save("utils", a=a, limitAngle=la, a1=a1, a2=a2, angleDifference=ad, x=x, y=y, r=r, lx=lx, ly=ly)


# ----------------------------------------------------------------------------
# (11) InvPendulum yaw step response WITHOUT the `control` stand-in — the scenario of the reference's own
#      test (src/cyclistsocialforce/test.py:15-165): 30 deg yaw step after 20 % of 10 s, speed held at
#      v_desired_default.  The closed-loop matrices come from the REFERENCE's assembly
#      (vehicle.py:1738-1786 + the gain tables of parameters.py:1857-1892, read off `bike.dynamics` after
#      `update_dynamics()`); the time stepping is done twice by SciPy alone, with no matrix exponential of
#      ours in the way:
#        zoh   scipy.signal.cont2discrete(method="zoh") recursion (the exact answer for an input that is
#              constant over each tick, which is what step_yaw feeds forced_response: vehicle.py:1835-1842)
#        ode   scipy.integrate.solve_ivp(Radau, rtol 1e-12) tick by tick on x' = A x + B u
#      and once through the reference's step_yaw on the `control` stand-in (`shim`).  All three are stored; the
#      oracle and the HIP kernel are tested against `zoh`.
#      test.py's own expected curve is built by pole placement (poles 30 x (-0.2, -0.1 +- 0.1j, -0.15, -0.1),
#      Ku = 1 / -0.4631...): those are the gains of the table's commented-out predecessor
#      (parameters.py:1858-1861), NOT of HEAD's table, whose closed-loop poles move with speed.  The placement
#      gains and HEAD's table gains at v = 5 are stored side by side to document that test.py is stale at HEAD.
# ----------------------------------------------------------------------------
def gen_invpend_yawstep():
    from scipy.integrate import solve_ivp
    from scipy.signal import cont2discrete, place_poles

    bike = make_vehicle("invpend", (0, 0, 0, 0, 0, 0))
    v = bike.params.v_desired_default
    bike.s[3] = v                                             # test.py:31
    h = bike.params.t_s
    t = np.arange(0, 10, h)
    psi_d = np.zeros_like(t)
    psi_d[int(0.2 * len(t)):] = 2 * np.pi * 30 / 360          # test.py:35-36
    Fx, Fy = v * np.cos(psi_d), v * np.sin(psi_d)             # test.py:38-39

    bike.update_dynamics()                                    # the reference's closed loop at this speed
    A = np.array(bike.dynamics.A, dtype=float)
    B = np.array(bike.dynamics.B, dtype=float).reshape(5, 1)

    # (shim) the reference's own step_yaw, tick by tick, exactly as test.py:42-49 drives it
    shim = np.zeros((len(t), 3))                              # psi, delta, theta after each tick
    for k in range(len(t)):
        shim[k] = bike.step_yaw(Fx[k], Fy[k])
    # (zoh) SciPy's discretisation
    Ad, Bd, *_ = cont2discrete((A, B, np.eye(5), np.zeros((5, 1))), h, method="zoh")
    x = np.zeros(5)
    zoh = np.zeros((len(t), 3))
    for k in range(len(t)):
        x = Ad @ x + Bd[:, 0] * np.arctan2(Fy[k], Fx[k])
        zoh[k] = (x[4], x[0], x[2])
    # (ode) an implicit Runge-Kutta integration of the same linear system, tick by tick
    x = np.zeros(5)
    ode = np.zeros((len(t), 3))
    for k in range(len(t)):
        u = np.arctan2(Fy[k], Fx[k])
        sol = solve_ivp(lambda _t, y: A @ y + B[:, 0] * u, (0.0, h), x, method="Radau", rtol=1e-12, atol=1e-15,
                        jac=lambda _t, y: A)
        x = sol.y[:, -1]
        ode[k] = (x[4], x[0], x[2])
    print(f"invpend yaw step: |shim - zoh| max {np.abs(shim - zoh).max():.2e}, |ode - zoh| max {np.abs(ode - zoh).max():.2e}, "
          f"final yaw {zoh[-1, 0]:.6f} (target {psi_d[-1]:.6f})")
    assert np.abs(ode - zoh).max() < 1e-9 and np.abs(shim - zoh).max() < 1e-9

    # test.py:51-99: the placement the reference's test compares with, and HEAD's table at the same speed
    Ao, Bo, _, _ = bike.get_openloop_statespace_matrices()
    poles_test = np.array((-0.2 + 0j, -0.1 + 0.1j, -0.1 - 0.1j, -0.15 + 0j, -0.1 + 0j)) * 30
    K_place = place_poles(Ao, Bo[:, None], poles_test).gain_matrix[0]
    K_table, Ku_table = bike.params.fullstate_feedback_gains(v)
    speeds = np.array([2.0, 3.0, 5.0, 7.0])
    poles_head = []
    for vv in speeds:                                         # closed-loop poles of HEAD's table: they move with speed
        bike.s[3] = vv
        bike.update_dynamics()
        poles_head.append(np.sort_complex(np.linalg.eigvals(np.array(bike.dynamics.A, dtype=float))))
    save("invpend_yawstep", t_s=h, v=v, psi_d=psi_d, Fx=Fx, Fy=Fy, A=A, B=B[:, 0], zoh=zoh, ode=ode, shim=shim,
         K_place_testpy=K_place, Ku_testpy=1 / -0.46313878281084603, K_table_head=K_table[0], Ku_table_head=Ku_table,
         speeds=speeds, poles_head=np.array(poles_head))


# ----------------------------------------------------------------------------
# (12) PlanarBicycle — vehicle.py:2031-2076, dynamics.py:178-258, 1167-1226: planar two-wheeler with a pole-placed
#      steer / yaw loop (gains and input gain re-derived from the speed EVERY step, the input gain from a simulated
#      10-s step response) and first-order speed dynamics; TwoD force field and spline destination force.
#      Needs `control.place`, `control.ctrb` and the attribute form of `forced_response` from the stand-in.
# ----------------------------------------------------------------------------
def gen_planarbike():
    rng = np.random.default_rng(1212)
    S, F, O, X, G = [], [], [], [], []
    for k in range(120):
        s = (rng.uniform(-5, 5), rng.uniform(-5, 5), rng.uniform(-np.pi, np.pi), rng.uniform(0.5, 9), rng.uniform(-0.3, 0.3))
        v = make_vehicle("planarbike", s)
        f = (rng.normal(0, 4), rng.normal(0, 4))
        v.step(f[0], f[1])
        s1 = v.s.copy()
        x1 = np.array(v.dynamics.x, dtype=float).copy()
        gains = np.r_[np.ravel(v.dynamics.gains[0]), np.ravel(v.dynamics.gains[1])]     # of the first step's speed
        f2 = (rng.normal(0, 4), rng.normal(0, 4))
        v.step(f2[0], f2[1])
        S.append(np.r_[s, s1]); F.append(np.r_[f, f2]); O.append(v.s.copy()); X.append(np.r_[x1, v.dynamics.x]); G.append(gains)
    out = {"steps_s01": np.array(S), "steps_F01": np.array(F), "steps_s2": np.array(O), "steps_x12": np.array(X),
           "steps_gains": np.array(G)}
    vs = demo_bikes("planarbike")
    s0, vdes, off, dq = pop_arrays(vs)
    St, Ft = run_population(vs, 700, every=10)
    out.update({"demo_s0": s0, "demo_vdes": vdes, "demo_off": off, "demo_dq": dq, "demo_S": St, "demo_F": Ft})
    vs = random_population(np.random.default_rng(1313), "planarbike", 12, 25.0)
    s0, vdes, off, dq = pop_arrays(vs)
    St, Ft = run_population(vs, 150, every=10)
    out.update({"dense_s0": s0, "dense_vdes": vdes, "dense_off": off, "dense_dq": dq, "dense_S": St, "dense_F": Ft})
    save("planarbike", **out)


# ----------------------------------------------------------------------------
# (13) populations whose vehicles own DIFFERENT parameter sets — vehicle.py:64-204 (params per vehicle),
#      vehicle.py:1592-1612 (the field of vehicle i with ITS parameters), intersection.py:733-735 (its hfov)
# ----------------------------------------------------------------------------
HETERO_RECIPES = {
    "twod": [
        {},
        dict(hfov=1.1 * np.pi, f_0=10.0, sigma_0=0.6, sigma_1=5.5),
        dict(hfov=1.0, e_0=0.9, e_1=0.4, sigma_2=0.25, sigma_3=4.0, d_arrived_inter=3.0, k_p_v=13.0),
        dict(hfov=2 * np.pi, f_0=0.0, v_max_riding=[-1.0, 5.0], a_max=[-2.0, 0.8]),
    ],
    "bicycle": [
        {},
        dict(hfov=1.1 * np.pi, p_0=40.0, p_decay=4.0),
        dict(hfov=1.0, p_decay=6.0, d_arrived_inter=3.0, k_p_v=13.0, l=1.2),
        dict(v_max_riding=[-1.0, 5.0], k_p_delta=8.0, a_max=[-6.0, 6.0], delta_max=1.0),
    ],
    "invpend": [
        {},
        dict(hfov=1.1 * np.pi, f_0=10.0, h=1.1, m=80.0),
        dict(hfov=1.0, e_0=0.9, k_p_v=12.0, v_max_walk=1.2, delta_max=1.2, d_arrived_inter=2.5),
        dict(f_0=0.0, v_max_riding=[-1.0, 6.0], a_max=[-2.5, 0.8], l_1=0.55, l_2=0.5),
    ],
}
HETERO_PARAMS = {"twod": rp.InvPendulumBicycleParameters, "bicycle": rp.BicycleParameters,
                 "invpend": rp.InvPendulumBicycleParameters}


def gen_hetero():
    import json
    rng = np.random.default_rng(1414)
    out = {}
    for model, n, box, ticks in (("twod", 16, 28.0, 250), ("bicycle", 12, 28.0, 200), ("invpend", 10, 24.0, 150)):
        recipes = HETERO_RECIPES[model]
        ns = NSTATES[model]
        vs, cls = [], []
        for k in range(n):
            s0 = np.zeros(ns)
            s0[0] = rng.uniform(0, box)
            s0[1] = rng.uniform(0, box)
            s0[2] = rng.uniform(-np.pi, np.pi)
            s0[3] = rng.uniform(3, 4.8)
            c = k % len(recipes)
            v = make_vehicle(model, s0, vdes=rng.uniform(4, 4.9), id=str(k), params=HETERO_PARAMS[model](**recipes[c]))
            d = np.array([15.0, 29.0, 30.0])
            v.setDestinations(s0[0] + d * np.cos(s0[2]), s0[1] + d * np.sin(s0[2]))
            vs.append(v)
            cls.append(c)
        s0, vdes, off, dq = pop_arrays(vs)
        S, Ft = run_population(vs, ticks, every=10)
        out.update({f"{model}_s0": s0, f"{model}_vdes": vdes, f"{model}_off": off, f"{model}_dq": dq,
                    f"{model}_cls": np.array(cls), f"{model}_S": S, f"{model}_F": Ft,
                    f"{model}_recipes": np.array(json.dumps(recipes))})
    save("hetero", **out)



# ----------------------------------------------------------------------------
# (14) several vehicle CLASSES in one intersection — intersection.py:797-823 calls each vehicle's own
#      calcDestinationForce / calcRepulsiveForce / step, so any mix of rider models may share it
# ----------------------------------------------------------------------------
def gen_mixed():
    rng = np.random.default_rng(1515)
    order = ["twod", "bicycle", "invpend", "planarpoint", "planarbike"]
    n, box, ticks = 15, 28.0, 250
    vs, models = [], []
    for k in range(n):
        m = order[k % len(order)]
        s0 = np.zeros(NSTATES[m])
        s0[0] = rng.uniform(0, box)
        s0[1] = rng.uniform(0, box)
        s0[2] = rng.uniform(-np.pi, np.pi)
        s0[3] = rng.uniform(3, 4.8)
        kw = {}
        if k >= 10:                                   # the third vehicle of every class with parameters of its own
            kw["params"] = {"twod": rp.InvPendulumBicycleParameters(hfov=1.0, f_0=10.0),
                            "bicycle": rp.BicycleParameters(hfov=1.1 * np.pi, p_0=40.0, p_decay=4.0),
                            "invpend": rp.InvPendulumBicycleParameters(hfov=2.5, e_0=0.9, k_p_v=12.0),
                            "planarpoint": rp.PlanarPointBicycleParameters(hfov=1.5, f_0=5.0, poles=[-3.0 + 0j]),
                            "planarbike": rp.PlanarBicycleParameters(hfov=2.8, sigma_0=0.6)}[m]
        v = make_vehicle(m, s0, vdes=rng.uniform(4, 4.9), id=str(k), **kw)
        d = np.array([15.0, 29.0, 30.0])
        v.setDestinations(s0[0] + d * np.cos(s0[2]), s0[1] + d * np.sin(s0[2]))
        vs.append(v)
        models.append(m)
    s6 = np.zeros((n, 6))
    for k, v in enumerate(vs):
        s6[k, : len(v.s)] = v.s
    vdes = np.array([v.params.v_desired_default for v in vs])
    off = np.cumsum([0] + [v.destqueue.shape[0] for v in vs])
    dq = np.vstack([v.destqueue for v in vs])
    ins = ri.SocialForceIntersection(vs)
    S = np.zeros((ticks // 10 + 1, n, 6))
    F = np.zeros((ticks // 10, n, 2))
    S[0] = s6
    for t in range(ticks):
        ins.step()
        if (t + 1) % 10 == 0:
            for k, v in enumerate(vs):
                S[(t + 1) // 10, k, : len(v.s)] = v.s
                F[(t + 1) // 10 - 1, k] = v.force
    save("mixed", s0=s6, vdes=vdes, off=off, dq=dq, models=np.array(models), own=np.arange(n) >= 10, S=S, F=F)



# ----------------------------------------------------------------------------
# (15) the SUMO co-simulation seam — intersection.py:341-453 (footprint, lane end points, internal lanes,
#      entered / exited road users), 458-539 (route prototype of an arrival), 576-634 (departures), 679-688 (moveToXY push-back)
#      sumolib / traci are absent here: the reference class takes `net` as an argument and talks to a module-level
#      `traci`, so the duck-typed stand-ins of tests/sumo_fakes.py (the ones the mirror is driven with) are handed to it -
#      data for the reference's own code, none of which is replaced.
# ----------------------------------------------------------------------------
SUMO_SCRIPT = dict(ticks=130, every=25, stay=70, seed=5)


def sumo_script():
    """SUMO's side of the loop (scenario.py:376-437), scripted: who is on the internal lanes at every tick, and the
    state / route an arrival is handed over with.  Shared with the tests through the fixture."""
    arms = {"W": (-1, 0), "E": (1, 0), "S": (0, -1), "N": (0, 1)}
    rng = np.random.default_rng(SUMO_SCRIPT["seed"])
    on, born, per_tick, specs = {}, 0, [], {}
    for tick in range(SUMO_SCRIPT["ticks"]):
        if tick % SUMO_SCRIPT["every"] == 0:
            a, b = rng.choice(list(arms), 2, replace=False)
            ax, ay = arms[a]
            vid = f"veh{born}"
            specs[vid] = dict(route=(a + "_in", b + "_out"), since=tick,
                              s=[ax * 9.0 + ay * 1.6, ay * 9.0 - ax * 1.6, float(np.arctan2(-ay, -ax)), 4.0, 0.0])
            on[vid] = specs[vid]
            born += 1
        for vid in [v for v, sp in on.items() if tick - sp["since"] >= SUMO_SCRIPT["stay"]]:
            del on[vid]
        per_tick.append(tuple(on))
    return per_tick, specs


def gen_sumo_seam():
    import json

    sys.path.insert(0, os.path.join(os.path.dirname(OUT)))
    from sumo_fakes import FakeNet, FakeTraci

    tr = FakeTraci()
    ri.traci = tr                                               # the module-level name intersection.py:429-453, 679-688 use
    ins = ri.SocialForceIntersection([], id="J", activate_sumo_cosimulation=True, net=FakeNet())
    edges_in, edges_out = sorted(ins.inEdges), sorted(ins.outEdges)
    lane_in = np.array([[np.r_[x, y] for (x, y) in ins.inEdges[e]] for e in edges_in])       # [edge, lane, (x0 x1 y0 y1)]
    lane_out = np.array([[np.r_[x, y] for (x, y) in ins.outEdges[e]] for e in edges_out])
    per_tick, specs = sumo_script()
    entered_log, exited_log, ids_log, queues, moves, states = [], [], [], {}, [], []
    for tick, occ in enumerate(per_tick):
        tr.occupancy = {":J_0_0": occ}
        entered, exited = ins.find_entered_exited_roadusers()
        entered_log.append([str(v) for v in entered])
        exited_log.append([str(v) for v in exited])
        ins.remove_road_users_by_id(list(exited))
        for vid in entered:
            sp = specs[str(vid)]
            v = make_vehicle("twod", sp["s"], id=str(vid), route=sp["route"])
            np.random.seed(1000 + int(str(vid)[3:]))            # (the exit lane is drawn at random: one lane per arm here)
            ins.add_road_user(v)
            queues[str(vid)] = v.destqueue.copy()
        n0 = len(tr.moves)
        ins.step()
        tr.simulationStep()
        ids_log.append(ins.get_road_user_ids())
        for m in tr.moves[n0:]:
            moves.append([tick, int(m[1][3:]), m[4], m[5], m[6], m[7]])
        states.append(np.array([np.r_[int(v.id[3:]), v.s] for v in ins.vehicles]).reshape(-1, 6))
    qids = sorted(queues, key=lambda q: int(q[3:]))
    qoff = np.cumsum([0] + [queues[q].shape[0] for q in qids])
    save("sumo_seam", shape=np.asarray(ins.shape.vertices), edges_in=np.array(edges_in), edges_out=np.array(edges_out),
         lane_in=lane_in, lane_out=lane_out, internal_lane_ids=np.array(ins.internal_lane_ids),
         script=np.array(json.dumps(SUMO_SCRIPT)), entered=np.array(json.dumps(entered_log)), exited=np.array(json.dumps(exited_log)),
         ids=np.array(json.dumps(ids_log)), queue_ids=np.array(qids), queue_off=qoff, queue_rows=np.vstack([queues[q] for q in qids]),
         moves=np.array(moves), state_off=np.cumsum([0] + [s.shape[0] for s in states]), states=np.vstack(states),
         hist_n=np.array(ins.hist_n_vecs))


# ----------------------------------------------------------------------------
# (16) UncontrolledVehicle (vehicle.py:920-988) among cyclists: a scripted obstacle that exerts the TwoDBicycle field with
#      its CarParameters, follows the columns of its prescribed `traj` and feels no force; one with a trajectory that
#      crosses the cyclists' way (and ends before the run does), one without (the reference then reads the zeros of the
#      ring Vehicle.__init__ allocated: it sits at the origin from its first tick on)
# ----------------------------------------------------------------------------
def gen_uncontrolled():
    rng = np.random.default_rng(2024)
    ticks, n_bikes = 250, 8
    vs, models = [], []
    for k in range(n_bikes):
        s0 = np.array([rng.uniform(-14, -2), rng.uniform(-5, 5), rng.uniform(-0.3, 0.3), rng.uniform(3.5, 4.8), 0.0])
        v = make_vehicle("twod", s0, vdes=rng.uniform(4, 4.9), id=f"b{k}")
        d = np.array([12.0, 24.0, 25.0])
        v.setDestinations(s0[0] + d * np.cos(s0[2]), s0[1] + d * np.sin(s0[2]))
        vs.append(v)
        models.append("twod")
    T = 180                                                     # shorter than the run: the car stops where its script ends
    t = np.arange(T) * 0.01
    script = np.vstack([6.0 - 0.0 * t, -7.0 + 5.0 * t, np.full(T, np.pi / 2), np.full(T, 5.0)])
    car = rv.UncontrolledVehicle(tuple(script[:, 0]), trajectory=script, id="car")
    car.drawing = _NoDrawing()
    parked = rv.UncontrolledVehicle((3.0, -2.0, 0.5, 0.0), id="parked", params=rp.CarParameters(hfov=2.0, f_0=9.0))
    parked.drawing = _NoDrawing()
    vs += [car, parked]
    models += ["uncontrolled", "uncontrolled"]
    n = len(vs)
    s6 = np.zeros((n, 6))
    for k, v in enumerate(vs):
        s6[k, : len(v.s)] = v.s
    vdes = np.array([getattr(v.params, 'v_desired_default', 0.0) for v in vs])
    off = np.cumsum([0] + [v.destqueue.shape[0] for v in vs])
    dq = np.vstack([v.destqueue for v in vs])
    soff = np.cumsum([0] * (n_bikes + 1) + [T, 0])
    ins = ri.SocialForceIntersection(vs)
    S = np.zeros((ticks // 10 + 1, n, 6))
    F = np.zeros((ticks // 10, n, 2))
    S[0] = s6
    for tk in range(ticks):
        ins.step()
        if (tk + 1) % 10 == 0:
            for k, v in enumerate(vs):
                S[(tk + 1) // 10, k, : len(v.s)] = v.s
                F[(tk + 1) // 10 - 1, k] = v.force
    save("uncontrolled", s0=s6, vdes=vdes, off=off, dq=dq, models=np.array(models), script_off=soff, script_rows=script.T.copy(),
         parked_hfov=2.0, parked_f0=9.0, S=S, F=F)


if __name__ == "__main__":
    which = sys.argv[1:] or ["pair", "masks", "control", "dest", "pp", "road", "traj", "utils", "yawstep", "planarbike", "hetero", "mixed", "sumo", "uncontrolled", "hooks"]
    gens = {"pair": gen_pair_fields, "masks": gen_masks_and_totals, "control": gen_control_move,
            "dest": gen_dest_force, "pp": gen_planarpoint_steps, "road": gen_road,
            "traj": gen_trajectories, "utils": gen_utils, "yawstep": gen_invpend_yawstep,
            "planarbike": gen_planarbike, "hetero": gen_hetero, "mixed": gen_mixed, "sumo": gen_sumo_seam, "uncontrolled": gen_uncontrolled, "hooks": gen_hooks}
    for w in which:
        gens[w]()
