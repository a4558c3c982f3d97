"""Validated parameter objects with the names, defaults and error behaviour of the reference's
`cyclistsocialforce.parameters` (VehicleParameters :421, BicycleParameters :766,
PlanarPointBicycleParameters :1175, InvPendulumBicycleParameters :1414, RoadElementParameters :367),
plus `to_pod()` which flattens one object into the `csf_params` struct the kernels read.

Type rules follow the reference's property setters: floats must be `float` (ints raise TypeError),
ranges raise ValueError, and the attributes the reference marks "immutable" raise AttributeError on a
second assignment (e.g. parameters.py:516-528).  Drawing/visual parameter classes are out of scope.
"""
import numpy as np

from . import _ffi


_MUTATIONS = [0]


def mutation_count():
    """Number of assignments to any parameter attribute so far (the intersection rescans its population for changed
    per-agent parameters only when this moved)."""
    return _MUTATIONS[0]


class _Field:
    """Descriptor restating one reference property: (type check, range check, optional immutability)."""

    def __init__(self, kind="float", check=None, immutable=True, doc=""):
        self.kind, self.check, self.immutable, self.doc = kind, check, immutable, doc

    def __set_name__(self, owner, name):
        self.name = name
        self.slot = "_" + name

    def __get__(self, obj, objtype=None):
        if obj is None:
            return self
        return getattr(obj, self.slot)

    def __set__(self, obj, value):
        if self.immutable and hasattr(obj, self.slot):
            raise AttributeError(f"{self.name} is immutable.")
        if self.kind == "float":
            if not isinstance(value, float):
                raise TypeError(f"{self.name} must be a float.")
        elif self.kind == "anyfloat":  # reference casts: f_0 = float(f_0)  (parameters.py:623)
            value = float(value)
        elif self.kind == "pair":
            if not isinstance(value, (list, tuple)):
                raise TypeError(f"{self.name} must be list or tuple.")
            if not isinstance(value[0], float) or not isinstance(value[1], float):
                raise TypeError(f"{self.name}[0] and {self.name}[1] must be float.")
        if self.check is not None:
            msg = self.check(obj, value)
            if msg:
                raise ValueError(f"{self.name} {msg}, instead it was {value}")
        setattr(obj, self.slot, value)


def _ge0(obj, v):
    return None if v >= 0 else "must be >=0"


def _pair_order(obj, v):
    return None if v[0] <= v[1] else "must be ordered [min, max]"


class RoadElementParameters:
    """parameters.py:367-418 (F_0 and sigma of road-edge forces; the colours are drawing-only)."""

    F_0 = _Field(check=_ge0)
    sigma = _Field(check=_ge0)

    def __init__(self, roadsurface_color=(0.8, 0.8, 0.8), roadedge_color="white", roadedge_linewidth=1,
                 F_0=0.05, sigma=3.0):
        self.roadsurface_color = roadsurface_color
        self.roadedge_color = roadedge_color
        self.roadedge_linewidth = roadedge_linewidth
        self.F_0 = F_0
        self.sigma = sigma


class VehicleParameters:
    """parameters.py:421-750 — tactical parameters and the repulsive force field."""

    LIMIT_PREC = 1e-4

    def __setattr__(self, name, value):   # every assignment (validated fields and plain attributes) is counted
        object.__setattr__(self, name, value)
        _MUTATIONS[0] += 1

    t_s = _Field(check=_ge0)
    d_arrived_inter = _Field(check=_ge0, immutable=False)
    d_arrived_stop = _Field(check=_ge0)
    v_max_stop = _Field(check=_ge0)
    v_max_harddecel = _Field(check=_ge0)
    hfov = _Field(check=lambda o, v: None if 0 < v <= 2 * np.pi else "must be in ]0, 2pi]")
    f_0 = _Field(kind="anyfloat", check=_ge0, immutable=False)
    e_0 = _Field(check=lambda o, v: None if o.e_1 < v <= 1 else f"must be in ]e_1={o.e_1:.2f}, 1]", immutable=False)
    e_1 = _Field(check=lambda o, v: None if 0 <= v < o.e_0 else f"must be in [0, e_0={o.e_0:.2f}[", immutable=False)
    sigma_0 = _Field(check=_ge0, immutable=False)
    sigma_1 = _Field(check=_ge0, immutable=False)
    sigma_2 = _Field(check=lambda o, v: None if 0 < v < o.sigma_0 else "must be in ]0, sigma_0[", immutable=False)
    sigma_3 = _Field(check=lambda o, v: None if 0 < v < o.sigma_1 else "must be in ]0, sigma_1[", immutable=False)

    def __init__(self, t_s: float = 0.01, d_arrived_inter: float = 2.0, d_arrived_stop: float = 2.0,
                 v_max_stop: float = 0.1, v_max_harddecel: float = 2.5, hfov: float = 2 * np.pi,
                 calib_mode=False, verbose=True, rep_force={}, dest_force={}, dynamics={},
                 f_0: float = 7.0, e_0: float = 0.995, e_1: float = 0.7, sigma_0: float = 0.5,
                 sigma_1: float = 5.0, sigma_2: float = 0.3, sigma_3: float = 4.9) -> None:
        self.calib_mode = calib_mode
        self.verbose = verbose
        self.t_s = t_s
        self.d_arrived_inter = d_arrived_inter
        self.d_arrived_stop = d_arrived_stop
        self.v_max_stop = v_max_stop
        self.v_max_harddecel = v_max_harddecel
        self.hfov = hfov
        self.rep_force = rep_force
        self.dest_force = dest_force
        self._e_1 = 0  # parameters.py:501: lets the first e_0 assignment validate
        self.f_0 = f_0
        self.e_0 = e_0
        self.e_1 = e_1
        self.sigma_0 = sigma_0
        self.sigma_1 = sigma_1
        self.sigma_2 = sigma_2
        self.sigma_3 = sigma_3

    # -- flattening ---------------------------------------------------------------------------
    def to_pod(self, model, priority_rule=0):
        """csf_params for one vehicle class."""
        p = _ffi.Params()
        val = lambda name: float(getattr(self, name, 0.0) or 0.0)  # noqa: E731
        for name in ("t_s", "d_arrived_inter", "d_arrived_stop", "v_max_stop", "v_max_harddecel", "hfov",
                     "f_0", "e_0", "e_1", "sigma_0", "sigma_1", "sigma_2", "sigma_3"):
            setattr(p, name, val(name))
        for name in ("p_decay", "p_0", "l", "l_2", "delta_max", "k_p_v", "k_p_delta", "g", "h", "m",
                     "i_bike_longlong", "i_steer_vertvert", "c_steer", "v_max_walk", "delta_max_walk"):
            setattr(p, name, val(name))
        for name in ("v_max_riding", "a_max", "a_desired_default"):
            v = getattr(self, name, (0.0, 0.0))
            setattr(p, name, (_ffi.C.c_double * 2)(float(v[0]), float(v[1])))
        if p.i_steer_vertvert == 0.0:
            p.i_steer_vertvert = 1.0
        gains = getattr(self, "gains", None)
        poles = getattr(self, "poles", None)
        if poles is not None:  # dynamics.py:933-940: desired poles overwrite desired gains
            p.k_psi = float(-np.real(np.asarray(poles).flatten()[0]))
        elif gains is not None:
            p.k_psi = float(np.asarray(gains, dtype=float).flatten()[0])
        if int(model) == _ffi.PLANARBIKE and poles is not None:      # dynamics.py:190: the poles placed every step
            pl = np.asarray(poles, dtype=complex).flatten()
            if pl.size != 2 or abs((pl[0] + pl[1]).imag) > 1e-12 or abs((pl[0] * pl[1]).imag) > 1e-12:
                raise ValueError("PlanarBicycle needs two poles, real or a conjugate pair")
            p.pb_poles = (_ffi.C.c_double * 4)(pl[0].real, pl[0].imag, pl[1].real, pl[1].imag)
            p.k_psi = 0.0
        p.model = int(model)
        p.priority_rule = int(priority_rule)
        p.traj_len = int(30 / self.t_s)  # vehicle.py:159
        return p


class BicycleParameters(VehicleParameters):
    """parameters.py:766-1174."""

    v_max_riding = _Field(kind="pair", check=_pair_order)
    v_desired_default = _Field(check=_ge0, immutable=False)
    p_decay = _Field(check=_ge0)
    p_0 = _Field(check=_ge0)
    delta_max = _Field(check=lambda o, v: None if 0 <= v <= np.pi else "must be in [0, pi]")
    a_max = _Field(kind="pair", check=_pair_order)
    a_desired_default = _Field(kind="pair", check=_pair_order)
    k_p_v = _Field(check=_ge0)
    k_p_delta = _Field(check=_ge0)

    def __init__(self, v_max_riding: tuple = [-1.0, 10.0], v_desired_default: float = 5.0,
                 p_decay: float = 5.0, p_0: float = 30.0, hfov: float = np.pi * 2 / 3,
                 v_max_stop: float = 0.6, l: float = 1.0, l_1: float = None, l_2: float = None,
                 delta_max: float = 1.4, a_max: tuple = [-10.0, 10.0],
                 a_desired_default: tuple = [-5.0, 5.0], k_p_v: float = 10.0, k_p_delta: float = 10.0,
                 t_s: float = 0.01, d_arrived_inter: float = 2.0, d_arrived_stop: float = 2.0,
                 v_max_harddecel: float = 2.5, g=9.81, **kwargs) -> None:
        VehicleParameters.__init__(self, t_s=t_s, d_arrived_inter=d_arrived_inter,
                                   d_arrived_stop=d_arrived_stop, v_max_stop=v_max_stop,
                                   v_max_harddecel=v_max_harddecel, hfov=hfov, **kwargs)
        self.v_max_riding = v_max_riding
        self.v_desired_default = v_desired_default
        self.p_decay = p_decay
        self.p_0 = p_0
        # wheelbase bookkeeping — parameters.py:890-921: any one of l, l_1, l_2 may be None
        for name, v in (("l", l), ("l_1", l_1), ("l_2", l_2)):
            if v is not None:
                if not isinstance(v, float):
                    raise TypeError(f"{name} must be a float.")
                if not v >= 0:
                    raise ValueError(f"{name} must be >=0, instead it was {v:.2f}")
        if l_1 is None and l_2 is None:
            assert l is not None, "If l_1 and l_2 are None, l may not be None!"
            l_1 = l / 2
            l_2 = l / 2
        if l is None:
            assert l_1 is not None and l_2 is not None, "Only one of l, l_1, l_2 may be None!"
            l = l_1 + l_2
        elif l_1 is None:
            l_1 = l - l_2
        elif l_2 is None:
            l_2 = l - l_1
        else:
            assert l == l_1 + l_2, "Equality l = l_1 + l_2 must hold!"
        self._l, self._l_1, self._l_2 = l, l_1, l_2
        self.delta_max = delta_max
        self.a_max = a_max
        self.a_desired_default = a_desired_default
        self.k_p_v = k_p_v
        self.k_p_delta = k_p_delta
        self.g = g

    def _immutable(name):  # noqa: N805
        def get(self):
            return getattr(self, "_" + name)

        def set_(self, v):
            raise AttributeError(f"{name} is immutable.")

        return property(get, set_)

    l = _immutable("l")
    l_1 = _immutable("l_1")
    l_2 = _immutable("l_2")
    del _immutable


class PlanarPointBicycleParameters(BicycleParameters):
    """parameters.py:1175-1201: one real pole (default -2) or one gain of the yaw-tracking loop."""

    FIXED_POLES = 0 + 0j
    N_POLES = 4

    def __init__(self, poles=[-2 + 0j], gains=[2], **kwargs):
        BicycleParameters.__init__(self, **kwargs)
        self.gains = gains
        self.poles = poles

    @property
    def poles(self):
        return self._poles

    @poles.setter
    def poles(self, poles):
        if poles is None:
            poles = [-2 + 0j]
        if not isinstance(poles, (list, tuple, np.ndarray)):
            poles = np.array(poles)
        self._poles = [poles[0]]


class InvPendulumBicycleParameters(BicycleParameters):
    """parameters.py:1414-1892 — also the parameter class of TwoDBicycle (vehicle.py:1353-1357)."""

    h = _Field(check=_ge0)
    m = _Field(check=_ge0)
    i_bike_longlong = _Field(check=_ge0)
    i_steer_vertvert = _Field(check=_ge0)
    c_steer = _Field(check=_ge0)
    v_max_walk = _Field(check=_ge0)
    delta_max_walk = _Field(check=lambda o, v: None if 0 < v <= np.pi else "must be in ]0,pi]")

    def __init__(self, v_max_riding: tuple = [-1.0, 7.0], v_desired_default: float = 5.0,
                 hfov: float = np.pi * 2 / 3, a_max: tuple = [-3.0, 1.0],
                 a_desired_default: tuple = [-1.0, 0.5], l: float = None, l_1: float = 0.5,
                 l_2: float = 0.5, delta_max: float = 1.4, h: float = 1.0, m: float = 87.0,
                 i_bike_longlong: float = 3.28, i_steer_vertvert: float = 0.07, c_steer: float = 50.0,
                 k_p_v: float = 10.0, k_d0_r2: float = -600.0, k_d1_r2: float = 0.2, k_p_r1: float = 0.25,
                 k_i0_r1: float = 0.2, t_s: float = 0.01, d_arrived_inter: float = 2.0,
                 d_arrived_stop: float = 2.0, v_max_harddecel: float = 2.5, v_max_stop: float = 0.6,
                 v_max_walk: float = 1.5, delta_max_walk: float = 0.174, f_0: float = 7.0,
                 e_0: float = 0.995, e_1: float = 0.7, sigma_0: float = 0.5, sigma_1: float = 5.0,
                 sigma_2: float = 0.3, sigma_3: float = 4.9, g: float = 9.81) -> None:
        BicycleParameters.__init__(
            self, v_max_riding=v_max_riding, v_desired_default=v_desired_default, hfov=hfov, a_max=a_max,
            a_desired_default=a_desired_default, l=l, l_1=l_1, l_2=l_2, delta_max=delta_max, k_p_v=k_p_v,
            t_s=t_s, d_arrived_inter=d_arrived_inter, d_arrived_stop=d_arrived_stop, v_max_stop=v_max_stop,
            v_max_harddecel=v_max_harddecel, f_0=f_0, e_0=e_0, e_1=e_1, sigma_0=sigma_0, sigma_1=sigma_1,
            sigma_2=sigma_2, sigma_3=sigma_3)
        self.h = h
        self.m = m
        self.i_bike_longlong = i_bike_longlong
        self.i_steer_vertvert = i_steer_vertvert
        self.c_steer = c_steer
        self.k_d0_r2 = k_d0_r2
        self.k_d1_r2 = k_d1_r2
        self.k_p_r1 = k_p_r1
        self.k_i0_r1 = k_i0_r1
        self.v_max_walk = v_max_walk
        self.delta_max_walk = delta_max_walk
        self.g = g
        self.tau_1_squared = (self.i_bike_longlong + self.m * self.h ** 2) / (self.m * self.g * self.h)

    def timevarying_combined_params(self, v: float):
        """parameters.py:1832-1855."""
        K_tau_2 = (v * self.l_2) / (self.g * self.l)
        K = (v ** 2) / (self.g * self.l)
        tau_3 = self.l / v
        return K, K_tau_2, tau_3


class PlanarBicycleParameters(BicycleParameters):
    """parameters.py:1203-1211: BicycleParameters + the desired poles of the steer / yaw loop."""

    def __init__(self, poles=(-1.0141284591434665 + 1.226826644413086j, -1.0141284591434665 - 1.226826644413086j),
                 **kwargs):
        BicycleParameters.__init__(self, **kwargs)
        self.poles = poles


# The bicycle of BalancingRiderBicycleParameters' default (parameters.py:16, 1217): the Balance Assist v1 bicycle with an average
# rider, as data/bicycleparams/balanceassist_bikeparams.py lists it (derived there from Moore's BicycleParameters data; BSD-2-Clause,
# Copyright (c) 2011-2025 BicycleParameters Authors: the licence text is in THIRD_PARTY_NOTICES.md at the root of this repository).
balanceassistv1_with_averagerider = dict(
    IBxx=16.136560964517308, IBxz=-2.5375819134691833, IByy=18.98228436804581, IBzz=4.308368614306412, IFxx=0.0995, IFyy=0.1902,
    IHxx=0.2984, IHxz=-0.038, IHyy=0.257, IHzz=0.0566, IRxx=0.1023, IRyy=0.1887, c=0.042, g=9.81, lam=0.255,
    mB=91.50000000000003, mF=2.235, mH=4.3, mR=4.085, rF=0.35231, rR=0.34895, v=1.0, w=1.113, xB=0.373106714751133, xH=0.921,
    yB=0.0, zB=-0.9697039390081493, zH=-0.86)


def whipple_carvallo_matrices(p):
    """(M, C1, K0, K2) of the linearised Whipple-Carvallo bicycle, M q'' + v C1 q' + (g K0 + v^2 K2) q = f with q = (roll, steer),
    from a parameter dictionary: Meijaard, Papadopoulos, Ruina & Schwab, Proc. R. Soc. A 463 (2007), Appendix A.  The reference
    takes them from bicycleparameters' Meijaard2007Model (parameters.py:1284-1300, dynamics.py:570); the formulas are checked
    against the paper's benchmark matrices and eigenvalues (tests/test_host_api.py)."""
    w, c, lam = p["w"], p["c"], p["lam"]
    mT = p["mR"] + p["mB"] + p["mH"] + p["mF"]
    xT = (p["xB"] * p["mB"] + p["xH"] * p["mH"] + w * p["mF"]) / mT
    zT = (-p["rR"] * p["mR"] + p["zB"] * p["mB"] + p["zH"] * p["mH"] - p["rF"] * p["mF"]) / mT
    ITxx = (p["IRxx"] + p["IBxx"] + p["IHxx"] + p["IFxx"] + p["mR"] * p["rR"] ** 2 + p["mB"] * p["zB"] ** 2 + p["mH"] * p["zH"] ** 2
            + p["mF"] * p["rF"] ** 2)
    ITxz = p["IBxz"] + p["IHxz"] - p["mB"] * p["xB"] * p["zB"] - p["mH"] * p["xH"] * p["zH"] + p["mF"] * w * p["rF"]
    ITzz = p["IRxx"] + p["IBzz"] + p["IHzz"] + p["IFxx"] + p["mB"] * p["xB"] ** 2 + p["mH"] * p["xH"] ** 2 + p["mF"] * w ** 2
    mA = p["mH"] + p["mF"]                                   # the front assembly: handlebar + fork, front wheel
    xA = (p["xH"] * p["mH"] + w * p["mF"]) / mA
    zA = (p["zH"] * p["mH"] - p["rF"] * p["mF"]) / mA
    IAxx = p["IHxx"] + p["IFxx"] + p["mH"] * (p["zH"] - zA) ** 2 + p["mF"] * (p["rF"] + zA) ** 2
    IAxz = p["IHxz"] - p["mH"] * (p["xH"] - xA) * (p["zH"] - zA) + p["mF"] * (w - xA) * (p["rF"] + zA)
    IAzz = p["IHzz"] + p["IFxx"] + p["mH"] * (p["xH"] - xA) ** 2 + p["mF"] * (w - xA) ** 2
    sl, cl = np.sin(lam), np.cos(lam)
    uA = (xA - w - c) * cl - zA * sl                         # the front assembly's centre of mass ahead of the steer axis
    IAll = mA * uA ** 2 + IAxx * sl ** 2 + 2 * IAxz * sl * cl + IAzz * cl ** 2
    IAlx = -mA * uA * zA + IAxx * sl + IAxz * cl
    IAlz = mA * uA * xA + IAxz * sl + IAzz * cl
    mu = c / w * cl
    SR, SF = p["IRyy"] / p["rR"], p["IFyy"] / p["rF"]        # gyrostatic coefficients of the wheels
    ST = SR + SF
    SA = mA * uA + mu * mT * xT
    M = np.array([[ITxx, IAlx + mu * ITxz], [IAlx + mu * ITxz, IAll + 2 * mu * IAlz + mu ** 2 * ITzz]])
    K0 = np.array([[mT * zT, -SA], [-SA, -SA * sl]])
    K2 = np.array([[0.0, (ST - mT * zT) / w * cl], [0.0, (SA + SF * sl) / w * cl]])
    C1 = np.array([[0.0, mu * ST + SF * cl + ITxz / w * cl - mu * mT * zT], [-(mu * ST + SF * cl), IAlz / w * cl + mu * (SA + ITzz / w * cl)]])
    return M, C1, K0, K2


class BalancingRiderBicycleParameters(BicycleParameters):
    """parameters.py:1214-1411: BicycleParameters + the Whipple-Carvallo bicycle (a parameter dictionary in the notation of
    Meijaard et al. 2007) + the control model that says where the rider wants the closed loop's poles - fixed `poles`, fixed
    `gains`, or a pole model file whose component mean over speed gives them (polemodel.py)."""

    def __init__(self, bicycleParameterDict=balanceassistv1_with_averagerider, poles=None, gains=None,
                 controlparam_filename="BR1_ImRe5GivenV_pole-model-params.yaml", stochastic_control_behavior=False,
                 controlparam_resampling_speedthresh=0.8333, controlparam_polemodel_component=0, p_dist_roll=0.00, p_dist_steer=0.00,
                 T_dist_roll=9000, T_dist_steer=1000, **kwargs):
        from . import polemodel

        bike = dict(bicycleParameterDict)
        kwargs = dict(kwargs, l=bike["w"], l_1=bike["w"] / 2)          # parameters.py:1291-1293: the dictionary's wheelbase wins
        BicycleParameters.__init__(self, **kwargs)
        self.bp_params = bike
        self.m = bike["mB"] + bike["mF"] + bike["mH"] + bike["mR"]     # :1300-1301
        self.g = bike["g"]
        self.stochastic_control_behavior = stochastic_control_behavior
        self.controlparam_filename = controlparam_filename
        self.controlparam_resampling_speedthresh = controlparam_resampling_speedthresh
        self.controlparam_polemodel_component = controlparam_polemodel_component
        self.polefuns = None
        if poles is None and gains is None:                            # :1309-1316
            self.controlparam_fix = False
            self.polesampler = None
            if stochastic_control_behavior:
                # parameters.py:1391-1396: the poles are SAMPLED from the pole model's mixture, conditioned on the speed, whenever
                # the speed has moved by controlparam_resampling_speedthresh since the last draw.  The draws are the host's
                # (polemodel.PoleSampler, on NumPy's global generator like the reference); the device holds the drawn poles as
                # constants until the next draw (to_pod).
                self.polesampler = polemodel.PoleSampler(controlparam_filename)
                self.poles = None
                self.gains = None
                self.v_last_update = -10000
            elif controlparam_filename in polemodel.MEAN_FUNCTIONS:
                self.polefuns = polemodel.MEAN_FUNCTIONS[controlparam_filename]
            else:                                                      # a model file of the caller's: its path
                import os

                if not os.path.exists(controlparam_filename):
                    raise FileNotFoundError(f"Couldn't find Balancing Rider Control Behavior model {controlparam_filename}. "
                                            f"Available models are: {sorted(polemodel.MEAN_FUNCTIONS)} (or the path of a model file)")
                self.polefuns = polemodel.component_mean_functions(controlparam_filename)
            if not stochastic_control_behavior and controlparam_polemodel_component >= self.polefuns.shape[0]:
                raise ValueError(f"Balancing Rider Control Behavior model {controlparam_filename} has only {self.polefuns.shape[0]} "
                                 f"components but controlparam_polemodel_component is set to {controlparam_polemodel_component}!")
            self.v_last_update = -10000
            self.poles = None
            self.gains = None
        else:
            self.controlparam_fix = True
            self.poles = poles
            self.gains = gains
        if p_dist_roll > 0 or p_dist_steer:                            # dynamics.py:312-314
            raise Warning("Support for steer and roll torque disturbance removed!")
        self.p_dist_roll, self.p_dist_steer, self.T_dist_roll, self.T_dist_steer = p_dist_roll, p_dist_steer, T_dist_roll, T_dist_steer

    def get_state_space_matrices(self, v):
        """parameters.py:1324-1340: (A [4, 4], B [4, 2]) of the Whipple-Carvallo bicycle at speed v"""
        M, C1, K0, K2 = whipple_carvallo_matrices(self.bp_params)
        Minv = np.linalg.inv(M)
        A = np.zeros((4, 4))
        A[0:2, 2:4] = np.eye(2)
        A[2:4, 0:2] = -Minv @ (self.bp_params["g"] * K0 + v ** 2 * K2)
        A[2:4, 2:4] = -Minv @ (v * C1)
        B = np.zeros((4, 2))
        B[2:4, :] = Minv
        return A, B

    def update_control_params(self, v):
        """parameters.py:1374-1411: the desired poles at speed v (component mean of the pole model; fixed ones stay)"""
        if not self.controlparam_fix:
            from . import polemodel

            if self.stochastic_control_behavior:                       # :1391-1396
                if np.abs(v - self.v_last_update) > self.controlparam_resampling_speedthresh:
                    self.poles = self.polesampler.sample(float(v))
                    self.v_last_update = v
                return
            self.poles = polemodel.poles_at(self.polefuns[self.controlparam_polemodel_component], v)
            self.v_last_update = v

    def to_pod(self, model, priority_rule=0):
        p = BicycleParameters.to_pod(self, model, priority_rule)
        p.k_psi = 0.0
        bike = self.bp_params
        M, C1, K0, K2 = whipple_carvallo_matrices(bike)
        Minv = np.linalg.inv(M)
        d4 = _ffi.C.c_double * 4
        p.br_minv_k0g = d4(*(Minv @ (bike["g"] * K0)).ravel())
        p.br_minv_k2 = d4(*(Minv @ K2).ravel())
        p.br_minv_c1 = d4(*(Minv @ C1).ravel())
        p.br_minv_steer = (_ffi.C.c_double * 2)(*Minv[:, 1])
        p.br_yaw = (_ffi.C.c_double * 2)(np.cos(bike["lam"]) / bike["w"], np.cos(bike["lam"]) * bike["c"] / bike["w"])   # dynamics.py:301-303
        fun = np.zeros((5, 2))
        p.br_mode = 0
        if self.controlparam_fix and self.poles is not None:           # dynamics.py:382-391: desired poles overwrite desired gains
            pl = np.asarray(self.poles, dtype=complex).flatten()
            if pl.size != 5 or abs(pl[0].imag) > 1e-12 or abs((pl[1] - np.conj(pl[2])))  > 1e-12 or abs((pl[3] - np.conj(pl[4]))) > 1e-12:
                raise ValueError("BalancingRider poles: one real pole and two conjugate pairs (p0, p1, conj(p1), p2, conj(p2))")
            fun[:, 0] = [pl[0].real, pl[1].real, abs(pl[1].imag), pl[3].real, abs(pl[3].imag)]
        elif self.controlparam_fix:
            g = np.asarray(self.gains, dtype=float).flatten()
            if g.size != 5:
                raise ValueError("BalancingRider gains: five (k_phi, k_delta, k_phidot, k_deltadot, k_psi)")
            p.br_gains = (_ffi.C.c_double * 5)(*g)
            p.br_mode = 2
        elif self.stochastic_control_behavior:                         # the poles of the last draw, until the next one
            if self.poles is None:
                raise ValueError("stochastic_control_behavior: no poles drawn yet (BalancingRiderBicycle.__init__ draws the first, dynamics.py:306)")
            pl = np.asarray(self.poles, dtype=complex).flatten()
            fun[:, 0] = [pl[0].real, pl[1].real, abs(pl[1].imag), pl[3].real, abs(pl[3].imag)]
        else:
            fun = np.asarray(self.polefuns[self.controlparam_polemodel_component], dtype=float)
        p.br_pole_fun = (_ffi.C.c_double * 10)(*fun.ravel())
        return p


class CarParameters(VehicleParameters):
    """parameters.py:752-764: VehicleParameters + the footprint of the car (UncontrolledVehicle, vehicle.py:920-988)."""

    def __init__(self, length=4, width=2.0, **kwargs):
        VehicleParameters.__init__(self, **kwargs)
        self.length = length
        self.width = width


PARAMS_OF_MODEL = {
    _ffi.BALANCINGRIDER: BalancingRiderBicycleParameters,
    _ffi.UNCONTROLLED: CarParameters,
    _ffi.PLANARBIKE: PlanarBicycleParameters,
    _ffi.BICYCLE: BicycleParameters,
    _ffi.TWOD: InvPendulumBicycleParameters,
    _ffi.INVPEND: InvPendulumBicycleParameters,
    _ffi.PLANARPOINT: PlanarPointBicycleParameters,
}


def default_pod(model, priority_rule=0, **overrides):
    """csf_params of a vehicle class with the reference's defaults (keyword overrides allowed)."""
    if isinstance(model, str):
        model = {"bicycle": 0, "twod": 1, "invpend": 2, "planarpoint": 3, "planarbike": 4, "uncontrolled": 5, "balancingrider": 6}[model]
    return PARAMS_OF_MODEL[model](**overrides).to_pod(model, priority_rule)
