"""ctypes front-end of the CPU oracle (oracle/csf_oracle.c).

TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# CSF_ORACLE_LIB: another build of the same source (oracle/Makefile: `make asan` - AddressSanitizer + UBSan on the CPU)
LIB = os.environ.get("CSF_ORACLE_LIB") or os.path.join(HERE, "libcsf_oracle.so")

BICYCLE, TWOD, INVPEND, PLANARPOINT, PLANARBIKE, UNCONTROLLED, BALANCINGRIDER = 0, 1, 2, 3, 4, 5, 6
MODEL_IDS = {"bicycle": BICYCLE, "twod": TWOD, "invpend": INVPEND, "planarpoint": PLANARPOINT, "planarbike": PLANARBIKE,
             "uncontrolled": UNCONTROLLED, "balancingrider": BALANCINGRIDER}
N_STATES = {BICYCLE: 5, TWOD: 5, INVPEND: 6, PLANARPOINT: 4, PLANARBIKE: 5, UNCONTROLLED: 4, BALANCINGRIDER: 8}

ST_SPLINE, ST_NAN, ST_NAVSTATE = 1, 2, 4


class Params(C.Structure):
    _fields_ = [
        ("t_s", C.c_double), ("d_arrived_inter", C.c_double), ("d_arrived_stop", C.c_double),
        ("v_max_stop", C.c_double), ("v_max_harddecel", C.c_double), ("hfov", C.c_double),
        ("f_0", C.c_double), ("e_0", C.c_double), ("e_1", C.c_double),
        ("sigma_0", C.c_double), ("sigma_1", C.c_double), ("sigma_2", C.c_double), ("sigma_3", C.c_double),
        ("v_max_riding", C.c_double * 2), ("p_decay", C.c_double), ("p_0", C.c_double),
        ("l", C.c_double), ("l_2", C.c_double), ("delta_max", C.c_double),
        ("a_max", C.c_double * 2), ("a_desired_default", C.c_double * 2),
        ("k_p_v", C.c_double), ("k_p_delta", C.c_double), ("g", C.c_double),
        ("h", C.c_double), ("m", C.c_double), ("i_bike_longlong", C.c_double),
        ("i_steer_vertvert", C.c_double), ("c_steer", C.c_double),
        ("v_max_walk", C.c_double), ("delta_max_walk", C.c_double),
        ("k_psi", C.c_double), ("pb_poles", C.c_double * 4),
        ("br_minv_k0g", C.c_double * 4), ("br_minv_k2", C.c_double * 4), ("br_minv_c1", C.c_double * 4),
        ("br_minv_steer", C.c_double * 2), ("br_yaw", C.c_double * 2), ("br_pole_fun", C.c_double * 10), ("br_gains", C.c_double * 5),
        ("model", C.c_int32), ("priority_rule", C.c_int32), ("traj_len", C.c_int32), ("br_mode", C.c_int32),
    ]


# Defaults of the reference parameter classes.
#   VehicleParameters parameters.py:430-451; BicycleParameters :780-800;
#   InvPendulumBicycleParameters :1429-1472; PlanarPointBicycleParameters :1180-1201.
_VEHICLE = dict(t_s=0.01, d_arrived_inter=2.0, d_arrived_stop=2.0, v_max_stop=0.1, v_max_harddecel=2.5,
                hfov=2 * np.pi, f_0=7.0, e_0=0.995, e_1=0.7, sigma_0=0.5, sigma_1=5.0, sigma_2=0.3,
                sigma_3=4.9)
_BICYCLE = dict(_VEHICLE, v_max_riding=(-1.0, 10.0), p_decay=5.0, p_0=30.0, hfov=np.pi * 2 / 3,
                v_max_stop=0.6, l=1.0, l_2=0.5, delta_max=1.4, a_max=(-10.0, 10.0),
                a_desired_default=(-5.0, 5.0), k_p_v=10.0, k_p_delta=10.0, g=9.81,
                h=0.0, m=0.0, i_bike_longlong=0.0, i_steer_vertvert=1.0, c_steer=0.0,
                v_max_walk=0.0, delta_max_walk=0.0, k_psi=0.0, pb_poles=(0.0, 0.0, 0.0, 0.0))
_INVPEND = dict(_BICYCLE, v_max_riding=(-1.0, 7.0), a_max=(-3.0, 1.0), a_desired_default=(-1.0, 0.5),
                l=1.0, l_2=0.5, h=1.0, m=87.0, i_bike_longlong=3.28, i_steer_vertvert=0.07,
                c_steer=50.0, v_max_walk=1.5, delta_max_walk=0.174)
_PLANARPOINT = dict(_BICYCLE, k_psi=2.0)
_PLANARBIKE = dict(_BICYCLE, pb_poles=(-1.0141284591434665, 1.226826644413086, -1.0141284591434665, -1.226826644413086))
_CAR = dict(_VEHICLE, i_steer_vertvert=1.0)       # CarParameters (parameters.py:752-764): VehicleParameters + a footprint


def whipple_carvallo(p):
    """(M, C1, K0, K2) of the linearised Whipple-Carvallo bicycle from a parameter dictionary: Meijaard, Papadopoulos, Ruina &
    Schwab (2007), Appendix A - what bicycleparameters' Meijaard2007Model.form_reduced_canonical_matrices implements
    (parameters.py:1284-1300).  Pinned to the paper's benchmark matrices and eigenvalues in tests/test_oracle_golden.py."""
    w, c, lam = p["w"], p["c"], p["lam"]
    mT = p["mR"] + p["mB"] + p["mH"] + p["mF"]
    xT = (p["xB"] * p["mB"] + p["xH"] * p["mH"] + w * p["mF"]) / mT
    zT = (-p["rR"] * p["mR"] + p["zB"] * p["mB"] + p["zH"] * p["mH"] - p["rF"] * p["mF"]) / mT
    ITxx = p["IRxx"] + p["IBxx"] + p["IHxx"] + p["IFxx"] + p["mR"] * p["rR"]**2 + p["mB"] * p["zB"]**2 + p["mH"] * p["zH"]**2 + p["mF"] * p["rF"]**2
    ITxz = p["IBxz"] + p["IHxz"] - p["mB"] * p["xB"] * p["zB"] - p["mH"] * p["xH"] * p["zH"] + p["mF"] * w * p["rF"]
    ITzz = p["IRxx"] + p["IBzz"] + p["IHzz"] + p["IFxx"] + p["mB"] * p["xB"]**2 + p["mH"] * p["xH"]**2 + p["mF"] * w**2
    mA = p["mH"] + p["mF"]
    xA = (p["xH"] * p["mH"] + w * p["mF"]) / mA
    zA = (p["zH"] * p["mH"] - p["rF"] * p["mF"]) / mA
    IAxx = p["IHxx"] + p["IFxx"] + p["mH"] * (p["zH"] - zA)**2 + p["mF"] * (p["rF"] + zA)**2
    IAxz = p["IHxz"] - p["mH"] * (p["xH"] - xA) * (p["zH"] - zA) + p["mF"] * (w - xA) * (p["rF"] + zA)
    IAzz = p["IHzz"] + p["IFxx"] + p["mH"] * (p["xH"] - xA)**2 + p["mF"] * (w - xA)**2
    sl, cl = np.sin(lam), np.cos(lam)
    uA = (xA - w - c) * cl - zA * sl
    IAll = mA * uA**2 + IAxx * sl**2 + 2 * IAxz * sl * cl + IAzz * cl**2
    IAlx = -mA * uA * zA + IAxx * sl + IAxz * cl
    IAlz = mA * uA * xA + IAxz * sl + IAzz * cl
    mu = c / w * cl
    SR, SF = p["IRyy"] / p["rR"], p["IFyy"] / p["rF"]
    ST = SR + SF
    SA = mA * uA + mu * mT * xT
    M = np.array([[ITxx, IAlx + mu * ITxz], [IAlx + mu * ITxz, IAll + 2 * mu * IAlz + mu**2 * ITzz]])
    K0 = np.array([[mT * zT, -SA], [-SA, -SA * sl]])
    K2 = np.array([[0.0, (ST - mT * zT) / w * cl], [0.0, (SA + SF * sl) / w * cl]])
    C1 = np.array([[0.0, mu * ST + SF * cl + ITxz / w * cl - mu * mT * zT], [-(mu * ST + SF * cl), IAlz / w * cl + mu * (SA + ITzz / w * cl)]])
    return M, C1, K0, K2


def balancingrider_fields(bike, pole_fun=None, gains=None):
    """the br_* members of Params from a bicycle parameter dictionary and a control model: pole_fun [5, 2] ((intercept, slope) of
    p0_real, p1_real, p1_imag, p2_real, p2_imag; constant poles: slopes 0) or fixed gains [5]"""
    M, C1, K0, K2 = whipple_carvallo(bike)
    Minv = np.linalg.inv(M)
    d = dict(br_minv_k0g=tuple((Minv @ (bike["g"] * K0)).ravel()), br_minv_k2=tuple((Minv @ K2).ravel()), br_minv_c1=tuple((Minv @ C1).ravel()),
             br_minv_steer=tuple(Minv[:, 1]), br_yaw=(np.cos(bike["lam"]) / bike["w"], np.cos(bike["lam"]) * bike["c"] / bike["w"]),
             br_pole_fun=tuple(np.zeros(10)), br_gains=tuple(np.zeros(5)), br_mode=0)
    if gains is not None:
        d.update(br_gains=tuple(np.asarray(gains, dtype=float).ravel()), br_mode=2)
    else:
        d.update(br_pole_fun=tuple(np.asarray(pole_fun, dtype=float).ravel()))
    return d


def _balancingrider_defaults():
    g = np.load(os.path.join(os.path.dirname(HERE), "tests", "golden", "balancingrider.npz"))
    bike = dict(zip([str(k) for k in g["default_bike_names"]], [float(v) for v in g["default_bike_params"]]))
    # BalancingRiderBicycleParameters() (parameters.py:1214-1320): BicycleParameters with the wheelbase of the bicycle, and the
    # mean poles of component 0 of the model file BR1 (the reference's own PoleModel: tests/golden/make_golden_balancingrider.py)
    return dict(_BICYCLE, l=bike["w"], g=bike["g"], **balancingrider_fields(bike, pole_fun=g["polefun_BR1"][0]))


DEFAULTS = {BICYCLE: _BICYCLE, TWOD: _INVPEND, INVPEND: _INVPEND, PLANARPOINT: _PLANARPOINT, PLANARBIKE: _PLANARBIKE,
            UNCONTROLLED: _CAR}


def default_params(model, priority_rule=0, **overrides):
    if isinstance(model, str):
        model = MODEL_IDS[model]
    if model == BALANCINGRIDER and model not in DEFAULTS:
        DEFAULTS[model] = _balancingrider_defaults()
    d = dict(DEFAULTS[model], **overrides)
    p = Params()
    for k, v in d.items():
        if isinstance(v, (tuple, list)):
            setattr(p, k, (C.c_double * len(v))(*v))
        else:
            setattr(p, k, v)
    p.model = model
    p.priority_rule = priority_rule
    p.traj_len = int(30 / p.t_s)
    return p


def build(force=False):
    src = os.path.join(HERE, "csf_oracle.c")
    if os.environ.get("CSF_ORACLE_LIB"):
        return LIB
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", HERE, "-B", "libcsf_oracle.so"])
    return LIB


_lib = None


def cpu_share():
    """CPUs this process may really use: the cgroup quota if there is one (a GPU box gives a 1-GPU job 16 of its 128
    hardware threads), else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:                       # cgroup v2: "<quota> <period>" or "max <period>"
            quota, period = fh.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
                quota, period = int(fq.read()), int(fp.read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return n


def lib():
    global _lib
    if _lib is None:
        build()
        # OpenMP would start one thread per hardware thread of the host; beyond the CPU share they only spin against
        # each other (a 1 024-agent tick took 0.25 s instead of 6 ms on a GPU box).  OMP_NUM_THREADS wins if it is set.
        want = int(os.environ.get("OMP_NUM_THREADS", "0") or 0) or cpu_share()
        L = C.CDLL(LIB)
        dp = C.POINTER(C.c_double)
        L.csfo_limit_angle.restype = C.c_double
        L.csfo_limit_angle.argtypes = [C.c_double]
        L.csfo_angle_difference.restype = C.c_double
        L.csfo_angle_difference.argtypes = [C.c_double, C.c_double]
        L.csfo_limit_magnitude.argtypes = [dp, dp, C.c_double]
        L.csfo_pair_twod.argtypes = [C.POINTER(Params)] + [C.c_double] * 6 + [dp, dp]
        L.csfo_pair_bicycle.argtypes = [C.POINTER(Params)] + [C.c_double] * 6 + [dp, dp]
        L.csfo_untracked.restype = C.c_int
        L.csfo_untracked.argtypes = [C.c_double, C.c_int, C.c_int, C.c_int] + [C.c_double] * 5
        L.csfo_untracked_matrix.argtypes = [C.c_void_p, C.c_int, C.c_int64] + [C.c_void_p] * 4
        L.csfo_road_force.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_double, C.c_double, dp, dp]
        L.csfo_column_sums.argtypes = [C.POINTER(Params), C.c_int64] + [C.c_void_p] * 4 + [C.c_int64] + [C.c_void_p] * 3
        L.csfo_column_sums_classes.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int64] + [C.c_void_p] * 4 + [C.c_int64] + [C.c_void_p] * 3
        L.csfo_road_forces.argtypes = [C.c_int64] + [C.c_void_p] * 4 + [C.c_int64] + [C.c_void_p] * 4
        L.csfo_spline20.restype = C.c_int
        L.csfo_spline20.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.csfo_control_move.argtypes = [C.POINTER(Params), C.c_void_p, C.c_void_p, C.c_int, C.c_double,
                                        C.c_double, C.c_void_p]
        L.csfo_expm.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        L.csfo_planarbike_gains.argtypes = [C.POINTER(Params), C.c_double, C.c_void_p, C.POINTER(C.c_double)]
        L.csfo_balancingrider_gains.argtypes = [C.POINTER(Params), C.c_double, C.c_void_p]
        L.csfo_create.restype = C.c_void_p
        L.csfo_create.argtypes = [C.POINTER(Params), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.csfo_create_ns.restype = C.c_void_p
        L.csfo_create_ns.argtypes = [C.POINTER(Params), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.csfo_destroy.argtypes = [C.c_void_p]
        L.csfo_set_classes.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.csfo_set_road.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.csfo_set_script.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.csfo_step.argtypes = [C.c_void_p, C.c_int]
        for f in ("csfo_calc_forces_range", "csfo_integrate_range", "csfo_update_snapshot_range"):
            getattr(L, f).argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.csfo_get_state.argtypes = [C.c_void_p, C.c_void_p]
        L.csfo_get_forces.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.csfo_get_force_parts.argtypes = [C.c_void_p] + [C.c_void_p] * 4
        L.csfo_get_nav.argtypes = [C.c_void_p] + [C.c_void_p] * 4
        L.csfo_get_snapshot.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.csfo_set_snapshot.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.csfo_push_state.argtypes = [C.c_void_p] * 5
        L.csfo_dest_force.argtypes = [C.c_void_p, C.c_int, dp, dp]
        L.csfo_apply_forces.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.csfo_get_lti.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.csfo_set_lti.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.csfo_num_threads.restype = C.c_int
        L.csfo_set_num_threads.argtypes = [C.c_int]
        L.csfo_set_num_threads(want)
        L.csfo_sizeof_params.restype = C.c_size_t
        assert L.csfo_sizeof_params() == C.sizeof(Params)
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def limit_angle(t):
    return lib().csfo_limit_angle(float(t))


def angle_difference(a1, a2):
    return lib().csfo_angle_difference(float(a1), float(a2))


def limit_magnitude(x, y, r):
    x = np.array(x, dtype=np.float64)
    y = np.array(y, dtype=np.float64)
    for k in range(x.size):
        cx, cy = C.c_double(x[k]), C.c_double(y[k])
        lib().csfo_limit_magnitude(C.byref(cx), C.byref(cy), float(r[k]))
        x[k], y[k] = cx.value, cy.value
    return x, y


def pair_twod(params, src, x, y, psi):
    fx = np.zeros(len(x)); fy = np.zeros(len(x))
    a, b = C.c_double(), C.c_double()
    for k in range(len(x)):
        lib().csfo_pair_twod(C.byref(params), src[0], src[1], src[2], x[k], y[k], psi[k], C.byref(a), C.byref(b))
        fx[k], fy[k] = a.value, b.value
    return fx, fy


def pair_bicycle(params, src, v, x, y):
    fx = np.zeros(len(x)); fy = np.zeros(len(x))
    a, b = C.c_double(), C.c_double()
    for k in range(len(x)):
        lib().csfo_pair_bicycle(C.byref(params), src[0], src[1], src[2], v, x[k], y[k], C.byref(a), C.byref(b))
        fx[k], fy[k] = a.value, b.value
    return fx, fy


def untracked_matrix(hfov, rule, x, y, psi):
    """get_untracked_foes (intersection.py:690-745): U[i, j] = receiver j ignores source i; hfov a scalar or one per source"""
    n = len(x)
    x, y, psi = (np.ascontiguousarray(a, dtype=np.float64) for a in (x, y, psi))
    h = np.ascontiguousarray(np.broadcast_to(np.asarray(hfov, dtype=np.float64), (n,)))
    U = np.zeros((n, n), dtype=np.uint8)
    lib().csfo_untracked_matrix(_p(h), int(rule), n, _p(x), _p(y), _p(psi), U.ctypes.data_as(C.c_void_p))
    return U.astype(bool)


def road_force(verts, off, F0, sigma, x, y):
    verts = np.ascontiguousarray(verts, dtype=np.float64)
    vF0 = np.zeros(len(verts)); vsg = np.zeros(len(verts))
    for e in range(len(off) - 1):
        vF0[off[e]:off[e + 1]] = F0[e]
        vsg[off[e]:off[e + 1]] = sigma[e]
    vx = np.ascontiguousarray(verts[:, 0]); vy = np.ascontiguousarray(verts[:, 1])
    fx = np.zeros(len(x)); fy = np.zeros(len(x))
    a, b = C.c_double(), C.c_double()
    for k in range(len(x)):
        lib().csfo_road_force(len(verts), _p(vx), _p(vy), _p(vF0), _p(vsg), x[k], y[k], C.byref(a), C.byref(b))
        fx[k], fy[k] = a.value, b.value
    return fx, fy


def column_sums(params, x, y, psi, v, recv, cls=None, rule=0):
    """unclamped repulsive column sums (intersection.py:814-843) of the receivers `recv` over all sources; with `cls`,
    `params` is a list of parameter sets and cls[i] the set of source i"""
    x, y, psi, v = (np.ascontiguousarray(a, dtype=np.float64) for a in (x, y, psi, v))
    recv = np.ascontiguousarray(recv, dtype=np.int64)
    rx = np.zeros(recv.size); ry = np.zeros(recv.size)
    if cls is not None:
        tab = (Params * len(params))(*params)
        cls = np.ascontiguousarray(cls, dtype=np.uint8)
        lib().csfo_column_sums_classes(tab, _p(cls), int(rule), x.size, _p(x), _p(y), _p(psi), _p(v), recv.size, _p(recv), _p(rx), _p(ry))
        return rx, ry
    lib().csfo_column_sums(C.byref(params), x.size, _p(x), _p(y), _p(psi), _p(v), recv.size, _p(recv), _p(rx), _p(ry))
    return rx, ry


def road_forces(verts, off, F0, sigma, x, y):
    """road_force for many receivers at once (OpenMP)"""
    verts = np.ascontiguousarray(verts, dtype=np.float64)
    vF0 = np.repeat(np.asarray(F0, dtype=np.float64), np.diff(off))
    vsg = np.repeat(np.asarray(sigma, dtype=np.float64), np.diff(off))
    vx = np.ascontiguousarray(verts[:, 0]); vy = np.ascontiguousarray(verts[:, 1])
    x = np.ascontiguousarray(x, dtype=np.float64); y = np.ascontiguousarray(y, dtype=np.float64)
    fx = np.zeros(x.size); fy = np.zeros(x.size)
    lib().csfo_road_forces(len(verts), _p(vx), _p(vy), _p(vF0), _p(vsg), x.size, _p(x), _p(y), _p(fx), _p(fy))
    return fx, fy


def spline20(px, py):
    px = np.ascontiguousarray(px, dtype=np.float64); py = np.ascontiguousarray(py, dtype=np.float64)
    out = np.zeros((20, 6))
    rc = lib().csfo_spline20(len(px), _p(px), _p(py), _p(out))
    return rc, out


def control_move(params, s, dest, is_last, Fx, Fy):
    s = np.ascontiguousarray(s, dtype=np.float64); dest = np.ascontiguousarray(dest, dtype=np.float64)
    out = np.zeros(5)
    lib().csfo_control_move(C.byref(params), _p(s), _p(dest), int(is_last), float(Fx), float(Fy), _p(out))
    return out


def planarbike_gains(params, v):
    """(K_x [2], K_u) of PlanarTwoWheelerDynamics.update at speed v (dynamics.py:203-223, 1167-1226)"""
    kx = np.zeros(2)
    ku = C.c_double()
    lib().csfo_planarbike_gains(C.byref(params), float(v), _p(kx), C.byref(ku))
    return kx, ku.value


def balancingrider_gains(params, v):
    """K [5] of BalancingRiderDynamics._get_gains at speed v (dynamics.py:600-615)"""
    k = np.zeros(5)
    lib().csfo_balancingrider_gains(C.byref(params), float(v), _p(k))
    return k


def expm(A):
    A = np.ascontiguousarray(A, dtype=np.float64)
    E = np.zeros_like(A)
    lib().csfo_expm(A.shape[0], _p(A), _p(E))
    return E


class Population:
    """Population of one vehicle class stepped by the CPU oracle (SocialForceIntersection.step)."""

    def __init__(self, params, s0, vdes, qoff, dq, ns=None):
        """ns: state columns (default: those of params.model; 6 for a population of several vehicle classes, set_classes)"""
        self.params = params
        self.ns = N_STATES[params.model] if ns is None else int(ns)
        s0 = np.ascontiguousarray(np.asarray(s0, dtype=np.float64)[:, :self.ns])
        self.n = s0.shape[0]
        vdes = np.ascontiguousarray(np.broadcast_to(np.asarray(vdes, dtype=np.float64), (self.n,)))
        qoff = np.ascontiguousarray(qoff, dtype=np.int64)
        dq = np.ascontiguousarray(dq, dtype=np.float64).reshape(-1, 3)
        assert qoff.shape == (self.n + 1,) and qoff[-1] == dq.shape[0]
        self.h = C.c_void_p(lib().csfo_create_ns(C.byref(params), self.n, self.ns, _p(s0), _p(vdes), _p(qoff), _p(dq)))

    def __del__(self):
        try:
            if self.h:
                lib().csfo_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def set_classes(self, classes, cls):
        """per-vehicle parameter sets: `classes` a list of Params, `cls[a]` the set of agent a"""
        tab = (Params * len(classes))(*classes)
        cls = np.ascontiguousarray(cls, dtype=np.uint8)
        assert cls.shape == (self.n,) and (len(classes) == 0 or cls.max() < len(classes))
        lib().csfo_set_classes(self.h, len(classes), tab, _p(cls))

    def set_script(self, off, rows):
        """prescribed trajectories of UncontrolledVehicle agents (vehicle.py:958-960): CSR over all agents, rows (x, y, psi, v)"""
        off = np.ascontiguousarray(off, dtype=np.int64)
        rows = np.ascontiguousarray(rows, dtype=np.float64).reshape(-1, 4)
        assert off.shape == (self.n + 1,) and off[-1] == rows.shape[0]
        lib().csfo_set_script(self.h, _p(off), _p(rows))

    def set_road(self, off, verts, F0, sigma):
        off = np.ascontiguousarray(off, dtype=np.int64)
        verts = np.ascontiguousarray(verts, dtype=np.float64)
        F0 = np.ascontiguousarray(F0, dtype=np.float64); sigma = np.ascontiguousarray(sigma, dtype=np.float64)
        lib().csfo_set_road(self.h, len(off) - 1, _p(off), _p(verts), _p(F0), _p(sigma))

    def step(self, nticks=1):
        lib().csfo_step(self.h, int(nticks))

    def calc_forces_range(self, lo, hi):
        lib().csfo_calc_forces_range(self.h, lo, hi)

    def integrate_range(self, lo, hi):
        lib().csfo_integrate_range(self.h, lo, hi)

    def update_snapshot_range(self, lo, hi):
        lib().csfo_update_snapshot_range(self.h, lo, hi)

    def state(self):
        out = np.zeros((self.n, self.ns))
        lib().csfo_get_state(self.h, _p(out))
        return out

    def forces(self):
        fx = np.zeros(self.n); fy = np.zeros(self.n)
        lib().csfo_get_forces(self.h, _p(fx), _p(fy))
        return fx, fy

    def force_parts(self):
        a = [np.zeros(self.n) for _ in range(4)]
        lib().csfo_get_force_parts(self.h, *[_p(x) for x in a])
        return a

    def nav(self):
        ptr = np.zeros(self.n, dtype=np.int32); zn = np.zeros((self.n, 3), dtype=np.uint8)
        i = np.zeros(self.n, dtype=np.int32); st = np.zeros(self.n, dtype=np.uint32)
        lib().csfo_get_nav(self.h, _p(ptr), _p(zn), _p(i), _p(st))
        return ptr, zn.astype(bool), i, st

    def snapshot(self, lo, hi):
        out = np.zeros((hi - lo, 4))
        lib().csfo_get_snapshot(self.h, lo, hi, _p(out))
        return out

    def set_snapshot(self, lo, hi, snap):
        snap = np.ascontiguousarray(snap, dtype=np.float64)
        lib().csfo_set_snapshot(self.h, lo, hi, _p(snap))

    def push_state(self, s, ptr=None, znav=None, col=None):
        """re-anchor the oracle on a state produced elsewhere (test aid, see csfo_push_state); col = vehicle.i"""
        s = np.ascontiguousarray(s, dtype=np.float64).reshape(self.n, self.ns)
        ptr = None if ptr is None else np.ascontiguousarray(ptr, dtype=np.int32)
        znav = None if znav is None else np.ascontiguousarray(znav, dtype=np.uint8).reshape(self.n, 3)
        col = None if col is None else np.ascontiguousarray(np.broadcast_to(np.asarray(col, dtype=np.int32), (self.n,)))
        lib().csfo_push_state(self.h, _p(s), *[None if a is None else _p(a) for a in (ptr, znav, col)])

    def lti(self):
        """(vehicle.x [n, 5] with psi unwrapped, vehicle.zrid [n, 2]) of InvPendulum riders (vehicle.py:1728-1736)"""
        x = np.zeros((self.n, 5)); z = np.zeros((self.n, 2), dtype=np.uint8)
        lib().csfo_get_lti(self.h, _p(x), _p(z))
        return x, z.astype(bool)

    def set_lti(self, x=None, zrid=None):
        x = None if x is None else np.ascontiguousarray(x, dtype=np.float64).reshape(self.n, 5)
        zrid = None if zrid is None else np.ascontiguousarray(zrid, dtype=np.uint8).reshape(self.n, 2)
        lib().csfo_set_lti(self.h, None if x is None else _p(x), None if zrid is None else _p(zrid))

    def dest_force(self, a):
        fx, fy = C.c_double(), C.c_double()
        lib().csfo_dest_force(self.h, a, C.byref(fx), C.byref(fy))
        return fx.value, fy.value

    def apply_forces(self, Fx, Fy):
        Fx = np.ascontiguousarray(Fx, dtype=np.float64); Fy = np.ascontiguousarray(Fy, dtype=np.float64)
        lib().csfo_apply_forces(self.h, _p(Fx), _p(Fy))


def num_threads():
    return lib().csfo_num_threads()
