"""Array-level host API of the stepping engine: a thin, typed wrapper over the C ABI (include/csf.h).

`Engine` is what `intersection.SocialForceIntersection` drives; benchmarks and parity tests use it
directly with NumPy arrays.  Every number it returns was computed by the HIP kernels.
"""
import ctypes as C

import numpy as np

from . import _ffi
from ._ffi import BICYCLE, INVPEND, N_STATES, PLANARBIKE, PLANARPOINT, TWOD, UNCONTROLLED, EngineError, Params  # noqa: F401

MODEL_IDS = {"bicycle": BICYCLE, "twod": TWOD, "invpend": INVPEND, "planarpoint": PLANARPOINT, "planarbike": PLANARBIKE,
             "uncontrolled": UNCONTROLLED, "balancingrider": _ffi.BALANCINGRIDER}


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Engine:
    """One population of one vehicle class on one MI355X (csf_engine)."""

    def __init__(self, params, capacity, device=0):
        self._lib = _ffi.load()
        self._h = None
        if not isinstance(params, Params):
            raise TypeError("params must be a cyclistsocialforce_amd._ffi.Params")
        self.params = params
        self.model = int(params.model)
        self.ns = N_STATES[self.model]
        h = self._lib.csf_create_v(C.byref(params), C.sizeof(params), _ffi.ABI_VERSION, int(capacity), int(device))
        if not h:
            raise EngineError(self._lib.csf_last_error(None).decode())
        self._h = C.c_void_p(h)
        self.capacity = int(capacity)

    # -- plumbing ---------------------------------------------------------------------------
    def _ck(self, rc):
        if rc != 0:
            raise EngineError(f"[{rc}] " + self._lib.csf_last_error(self._h).decode())

    def close(self):
        if self._h:
            self._lib.csf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def n(self):
        return int(self._lib.csf_num_agents(self._h))

    # -- population -------------------------------------------------------------------------
    def add_agents(self, s0, v_desired):
        s0 = np.asarray(s0, dtype=np.float64)
        if s0.ndim != 2 or s0.shape[1] < self.ns:
            raise ValueError(f"s0 must be [n, >={self.ns}]")
        s0 = _f64(s0[:, : self.ns])
        vd = _f64(np.broadcast_to(np.asarray(v_desired, dtype=np.float64), (s0.shape[0],)))
        self._ck(self._lib.csf_add_agents(self._h, s0.shape[0], _ptr(s0), _ptr(vd)))

    def remove_agents(self, idx):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        self._ck(self._lib.csf_remove_agents(self._h, idx.size, _ptr(idx)))

    def replace_agents(self, leave, s0, v_desired, offsets, xyz_stop):
        """One traffic step (include/csf.h: csf_replace_agents): road users `leave` (indices) go, the rows of s0 join behind the rest
        with destination queues (offsets [n + 1], xyz_stop [sum, 3]).  Arrays that are already contiguous and of the right type
        are passed as they are."""
        leave = np.ascontiguousarray(leave, dtype=np.int32)
        s0 = np.asarray(s0, dtype=np.float64).reshape(-1, max(np.shape(s0)[-1] if np.ndim(s0) == 2 else self.ns, 1))
        if s0.size and s0.shape[1] < self.ns:
            raise ValueError(f"s0 must be [n, >={self.ns}]")
        s0 = _f64(s0[:, : self.ns])
        n = s0.shape[0]
        vd = _f64(np.broadcast_to(np.asarray(v_desired, dtype=np.float64), (n,)))
        off = np.ascontiguousarray(offsets, dtype=np.int64)
        rows = _f64(xyz_stop)
        if off.size != n + 1 or rows.size != 3 * (int(off[-1]) if n else 0):
            raise ValueError("offsets [n + 1] and xyz_stop [offsets[-1], 3] must describe one queue per arrival")
        self._ck(self._lib.csf_replace_agents(self._h, leave.size, _ptr(leave), n, _ptr(s0), _ptr(vd), _ptr(off), _ptr(rows)))

    def set_dest_queue(self, agents, offsets, xyz_stop, reset=False):
        agents = np.ascontiguousarray(agents, dtype=np.int32)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        xyz = _f64(xyz_stop).reshape(-1, 3)
        if offsets.shape != (agents.size + 1,) or offsets[-1] != xyz.shape[0]:
            raise ValueError("offsets must be [n+1] and end at the number of rows")
        self._ck(self._lib.csf_set_dest_queue(self._h, agents.size, _ptr(agents), _ptr(offsets), _ptr(xyz), int(reset)))  # reset: 0 append, 1 replace, 2 replace + keep pointer

    def set_script(self, agents, offsets, rows):
        """prescribed trajectories of UncontrolledVehicle road users (include/csf.h: csf_set_script): rows (x, y, psi, v)"""
        agents = np.ascontiguousarray(agents, dtype=np.int32)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        rows = _f64(rows).reshape(-1, 4)
        if offsets.shape != (agents.size + 1,) or offsets[-1] != rows.shape[0]:
            raise ValueError("offsets must be [n+1] and end at the number of rows")
        self._ck(self._lib.csf_set_script(self._h, agents.size, _ptr(agents), _ptr(offsets), _ptr(rows)))

    def set_incremental(self, on=True):
        """population changes after the first tick: straight into the device arrays (True, default) or through the host
        mirror (include/csf.h: csf_set_incremental)"""
        self._ck(self._lib.csf_set_incremental(self._h, int(bool(on))))

    def set_road(self, offsets, verts, F0, sigma):
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        verts = _f64(verts).reshape(-1, 2)
        F0 = _f64(F0)
        sigma = _f64(sigma)
        self._ck(self._lib.csf_set_road_vertices(self._h, offsets.size - 1, _ptr(offsets), _ptr(verts), _ptr(F0), _ptr(sigma)))

    def set_params(self, params):
        self._ck(self._lib.csf_set_params(self._h, C.byref(params)))
        self.params = params

    def set_param_classes(self, classes, cls=None, idx=None):
        """Parameter sets for a population whose vehicles own different params objects (vehicle.py:64-204) or are of
        different classes (intersection.py:797-823): `classes` a sequence of csf_params (set 0 replaces the engine's own),
        `cls[k]` the set of agent `idx[k]` (default: everyone).  With sets of several vehicle classes the state arrays
        take the widest layout (x, y, psi, v, delta, theta); install them before add_agents."""
        classes = list(classes)
        tab = (type(self.params) * len(classes))(*classes)
        rows_first = cls is not None and len(classes) < getattr(self, "_n_classes", 1)    # no row may point beyond the table
        if rows_first:
            self.set_agent_class(np.arange(self.n) if idx is None else idx, cls)
        self._ck(self._lib.csf_set_param_classes(self._h, len(classes), tab))
        self.params = classes[0]
        self._n_classes = len(classes)
        self.ns = int(self._lib.csf_num_states(self._h))         # sets of several vehicle classes: the widest state
        if cls is not None and not rows_first:
            self.set_agent_class(np.arange(self.n) if idx is None else idx, cls)

    def set_agent_class(self, idx, cls):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        cls = np.ascontiguousarray(np.broadcast_to(np.asarray(cls, dtype=np.int32), idx.shape))
        self._ck(self._lib.csf_set_agent_class(self._h, idx.size, _ptr(idx), _ptr(cls)))

    def set_v_desired(self, idx, v):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        v = _f64(np.broadcast_to(np.asarray(v, dtype=np.float64), idx.shape))
        self._ck(self._lib.csf_set_v_desired(self._h, idx.size, _ptr(idx), _ptr(v)))

    def set_priority_rule(self, rule):
        self._ck(self._lib.csf_set_priority_rule(self._h, int(rule)))

    def push_state(self, idx, s):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        s = _f64(s).reshape(idx.size, self.ns)
        self._ck(self._lib.csf_push_state(self._h, idx.size, _ptr(idx), _ptr(s)))

    def integrator_state(self):
        """(vehicle.x [n, 5] - InvPendulum: delta, ddelta, theta, dtheta, psi unwrapped, vehicle.py:1728-1733 -, the unwrapped
        yaw of the PlanarPoint / PlanarBicycle integrators [n], vehicle.zrid [n, 2]): what vehicle.s does not carry"""
        n = self.n
        x = np.zeros((n, 5)); psi = np.zeros(n); z = np.zeros((n, 2), dtype=np.uint8)
        self._ck(self._lib.csf_get_integrator_state(self._h, _ptr(x), _ptr(psi), _ptr(z)))
        return x, psi, z.astype(bool)

    def set_integrator_state(self, idx, x=None, psi_unwrapped=None, zrid=None):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        x = None if x is None else _f64(x).reshape(idx.size, 5)
        psi = None if psi_unwrapped is None else _f64(psi_unwrapped).reshape(idx.size)
        z = None if zrid is None else np.ascontiguousarray(zrid, dtype=np.uint8).reshape(idx.size, 2)
        self._ck(self._lib.csf_set_integrator_state(self._h, idx.size, _ptr(idx), *[None if a is None else _ptr(a) for a in (x, psi, z)]))

    # -- hot path ---------------------------------------------------------------------------
    def step(self, n_ticks=1, sync=False):
        self._ck(self._lib.csf_step(self._h, int(n_ticks)))
        if sync:
            self.sync()

    def sync(self):
        self._ck(self._lib.csf_sync(self._h))

    def calc_forces(self):
        self._ck(self._lib.csf_calc_forces(self._h))
        return self.forces()

    def apply_forces(self, Fx, Fy):
        Fx = _f64(Fx)
        Fy = _f64(Fy)
        if Fx.size != self.n or Fy.size != self.n:
            raise ValueError("Fx, Fy must have one entry per agent")
        self._ck(self._lib.csf_apply_forces(self._h, _ptr(Fx), _ptr(Fy)))

    def replay_forces(self, Fx, Fy, lengths=None, fix_speed=False, stride=1, return_states=True):
        """Calibration replay (calibration.py:438-460): Fx, Fy are [T, n]; returns states [T // stride, n, n_states]."""
        Fx = _f64(Fx)
        Fy = _f64(Fy)
        if Fx.ndim != 2 or Fx.shape != Fy.shape or Fx.shape[1] != self.n:
            raise ValueError("Fx, Fy must be [n_ticks, n_agents]")
        T = Fx.shape[0]
        out = np.zeros((T // stride, self.n, self.ns)) if return_states else None
        ln = None if lengths is None else np.ascontiguousarray(lengths, dtype=np.int32)
        if ln is not None and ln.shape != (self.n,):
            raise ValueError("lengths must have one entry per agent")
        self._ck(self._lib.csf_replay_forces(self._h, T, _ptr(Fx), _ptr(Fy), None if ln is None else _ptr(ln),
                                             int(bool(fix_speed)), int(stride), None if out is None else _ptr(out)))
        return out

    def dest_force(self):
        fx = np.zeros(self.n)
        fy = np.zeros(self.n)
        self._ck(self._lib.csf_dest_force(self._h, _ptr(fx), _ptr(fy)))
        return fx, fy

    # -- read-back --------------------------------------------------------------------------
    def state(self, with_nav=False):
        """[n, n_states] state (and destination pointers, one-hot navigation state, tick count): one packed transfer
        (csf_get_tick; csf_get_state copies component by component)"""
        s, ptr, zn, _, _, tick = self.tick_snapshot(forces=False)
        return (s, ptr, zn, tick) if with_nav else s

    def state_by_component(self, with_nav=False):
        """the same through csf_get_state"""
        n = self.n
        s = np.zeros((n, self.ns))
        if not with_nav:
            self._ck(self._lib.csf_get_state(self._h, _ptr(s), None, None, None))
            return s
        ptr = np.zeros(n, dtype=np.int32)
        zn = np.zeros((n, 3), dtype=np.uint8)
        tick = C.c_int64(0)
        self._ck(self._lib.csf_get_state(self._h, _ptr(s), _ptr(ptr), _ptr(zn), C.byref(tick)))
        return s, ptr, zn.astype(bool), tick.value

    def tick_snapshot(self, forces=True):
        """state [n, n_states], destination pointers, one-hot navigation state, total forces and the tick count in one
        device-to-host transfer (csf_get_tick)"""
        n = self.n
        s = np.zeros((n, self.ns))
        ptr = np.zeros(n, dtype=np.int32)
        zn = np.zeros((n, 3), dtype=np.uint8)
        fx = np.zeros(n) if forces else None
        fy = np.zeros(n) if forces else None
        tick = C.c_int64(0)
        self._ck(self._lib.csf_get_tick(self._h, _ptr(s), _ptr(ptr), _ptr(zn), None if fx is None else _ptr(fx),
                                        None if fy is None else _ptr(fy), C.byref(tick)))
        return s, ptr, zn.astype(bool), fx, fy, tick.value

    def step_snapshot(self, n_ticks=1, forces=True, reuse=False):
        """step(n_ticks) and tick_snapshot() in one call (csf_step_get_tick): a handful of road users then cost one launch and
        one wait per call.  reuse: the arrays returned are the engine's own buffers, overwritten by the next call (the host
        mirror copies them into its bulk arrays at once) - at three road users the allocations are a fifth of the call."""
        n = self.n
        if reuse:
            key = (n, self.ns)
            if getattr(self, "_snap_key", None) != key:
                bufs = (np.zeros((n, self.ns)), np.zeros(n, dtype=np.int32), np.zeros((n, 3), dtype=np.uint8), np.zeros(n), np.zeros(n))
                self._snap_key, self._snap_bufs, self._snap_ptrs, self._snap_tick = key, bufs, [_ptr(b) for b in bufs], C.c_int64(0)
            s, ptr, zn, fx, fy = self._snap_bufs
            p = self._snap_ptrs
            self._ck(self._lib.csf_step_get_tick(self._h, int(n_ticks), p[0], p[1], p[2], p[3] if forces else None,
                                                 p[4] if forces else None, C.byref(self._snap_tick)))
            return s, ptr, zn, (fx if forces else None), (fy if forces else None), self._snap_tick.value
        s = np.zeros((n, self.ns))
        ptr = np.zeros(n, dtype=np.int32)
        zn = np.zeros((n, 3), dtype=np.uint8)
        fx = np.zeros(n) if forces else None
        fy = np.zeros(n) if forces else None
        tick = C.c_int64(0)
        self._ck(self._lib.csf_step_get_tick(self._h, int(n_ticks), _ptr(s), _ptr(ptr), _ptr(zn), None if fx is None else _ptr(fx),
                                             None if fy is None else _ptr(fy), C.byref(tick)))
        return s, ptr, zn.astype(bool), fx, fy, tick.value

    def step_into(self, n_ticks, s, ptr, zn, fx=None, fy=None):
        """step_snapshot straight into the caller's arrays: s [n, n_states] float64, ptr [n] int32, zn [n, 3] bool or uint8
        (one-hot), fx / fy [n] float64 or None - all C-contiguous (the host mirror's own bulk arrays: no copy in between)"""
        n = self.n
        key = (s.ctypes.data, ptr.ctypes.data, zn.ctypes.data, 0 if fx is None else fx.ctypes.data, 0 if fy is None else fy.ctypes.data, n)
        if getattr(self, "_into_key", None) != key:
            if not (s.flags.c_contiguous and ptr.flags.c_contiguous and zn.flags.c_contiguous and s.shape == (n, self.ns) and s.dtype == np.float64
                    and ptr.shape == (n,) and ptr.dtype == np.int32 and zn.shape == (n, 3) and zn.dtype.itemsize == 1):
                raise ValueError("step_into: s [n, n_states] float64, ptr [n] int32, zn [n, 3] of one byte each, C-contiguous")
            for f in (fx, fy):
                if f is not None and not (f.flags.c_contiguous and f.shape == (n,) and f.dtype == np.float64):
                    raise ValueError("step_into: fx, fy [n] float64, C-contiguous")
            self._into_key = key
            self._into_ptrs = [_ptr(s), _ptr(ptr), _ptr(zn), None if fx is None else _ptr(fx), None if fy is None else _ptr(fy)]
            self._into_tick = C.c_int64(0)
        p = self._into_ptrs
        self._ck(self._lib.csf_step_get_tick(self._h, int(n_ticks), p[0], p[1], p[2], p[3], p[4], C.byref(self._into_tick)))
        return self._into_tick.value

    @property
    def tick(self):
        t = C.c_int64(0)
        self._ck(self._lib.csf_get_state(self._h, None, None, None, C.byref(t)))
        return t.value

    def forces(self):
        fx = np.zeros(self.n)
        fy = np.zeros(self.n)
        self._ck(self._lib.csf_get_forces(self._h, _ptr(fx), _ptr(fy)))
        return fx, fy

    def force_parts(self):
        a = [np.zeros(self.n) for _ in range(4)]
        self._ck(self._lib.csf_get_force_parts(self._h, *[_ptr(x) for x in a]))
        return a

    def status(self):
        st = np.zeros(self.n, dtype=np.uint32)
        self._ck(self._lib.csf_status(self._h, _ptr(st)))
        return st

    def enable_history(self, stride=1, capacity=3000):
        self._ck(self._lib.csf_enable_history(self._h, int(stride), int(capacity)))

    def history(self, first, count):
        out = np.zeros((count, self.n, self.ns))
        self._ck(self._lib.csf_get_history(self._h, int(first), int(count), _ptr(out)))
        return out

    def pair_force(self, src, x, y, psi, apply_fov=False):
        src = _f64(src).reshape(4)
        x = _f64(x); y = _f64(y); psi = _f64(psi)
        fx = np.zeros(x.size)
        fy = np.zeros(x.size)
        self._ck(self._lib.csf_pair_force(self._h, _ptr(src), x.size, _ptr(x), _ptr(y), _ptr(psi), int(apply_fov), _ptr(fx), _ptr(fy)))
        return fx, fy

    def untracked(self):
        """get_untracked_foes() (intersection.py:690-745): bool [n, n], row = source, column = receiver"""
        n = self.n
        out = np.zeros((n, n), dtype=np.uint8)
        self._ck(self._lib.csf_untracked(self._h, _ptr(out)))
        return out.astype(bool)

    def update_destination(self, idx):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        self._ck(self._lib.csf_update_destination(self._h, idx.size, _ptr(idx)))

    def update_nav_state(self, idx, stop=None):
        """(vd, ddest) of Vehicle.updateNavState(stop) for the listed agents; stop None reads the queue's stop flags"""
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        vd = np.zeros(idx.size)
        dd = np.zeros(idx.size)
        st = None if stop is None else np.ascontiguousarray(np.broadcast_to(np.asarray(stop, dtype=np.int32), idx.shape))
        self._ck(self._lib.csf_update_nav_state(self._h, idx.size, _ptr(idx), None if st is None else _ptr(st), _ptr(vd), _ptr(dd)))
        return vd, dd

    def set_dest_pointer(self, idx, ptr):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        ptr = np.ascontiguousarray(np.broadcast_to(np.asarray(ptr, dtype=np.int32), idx.shape))
        self._ck(self._lib.csf_set_dest_pointer(self._h, idx.size, _ptr(idx), _ptr(ptr)))

    # -- sharding ---------------------------------------------------------------------------
    @staticmethod
    def comm_unique_id():
        lib = _ffi.load()
        buf = (C.c_uint8 * _ffi.UNIQUE_ID_BYTES)()
        rc = lib.csf_comm_unique_id(buf)
        if rc != 0:
            raise EngineError(f"[{rc}] " + lib.csf_last_error(None).decode())
        return bytes(buf)

    def comm_init(self, unique_id, rank, world):
        if unique_id is None:
            self._ck(self._lib.csf_comm_init(self._h, None, int(rank), int(world)))
            return
        buf = (C.c_uint8 * _ffi.UNIQUE_ID_BYTES).from_buffer_copy(bytes(unique_id))
        self._ck(self._lib.csf_comm_init(self._h, buf, int(rank), int(world)))

    @staticmethod
    def loopback_group(engines):
        """One-device rehearsal of the sharded path (include/csf.h: csf_comm_init_loopback): the engines, all holding the
        same population, become ranks 0 .. len-1; step them together with Engine.step_group."""
        lib = _ffi.load()
        arr = (C.c_void_p * len(engines))(*[e._h for e in engines])
        rc = lib.csf_comm_init_loopback(arr, len(engines))
        if rc != 0:
            raise EngineError(f"[{rc}] " + "; ".join(lib.csf_last_error(e._h).decode() for e in engines))

    @staticmethod
    def step_group(engines, n_ticks=1, sync=False):
        lib = _ffi.load()
        arr = (C.c_void_p * len(engines))(*[e._h for e in engines])
        rc = lib.csf_step_group(arr, len(engines), int(n_ticks))
        if rc != 0:
            raise EngineError(f"[{rc}] " + "; ".join(lib.csf_last_error(e._h).decode() for e in engines))
        if sync:
            engines[0].sync()

    def shard_range(self):
        lo, hi = C.c_int64(0), C.c_int64(0)
        self._ck(self._lib.csf_shard_range(self._h, C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def far_radius(self):
        """Radius (m) beyond which batches of sources are skipped (inf: every pair is evaluated); include/csf.h."""
        r = C.c_double(0)
        self._ck(self._lib.csf_far_radius(self._h, C.byref(r)))
        return r.value

    # -- measurement ------------------------------------------------------------------------
    def profile(self, every=1):
        """HIP events around the pair kernel on every `every`-th tick (0 / False: off)."""
        self._ck(self._lib.csf_profile_enable(self._h, int(every)))

    def profile_read(self):
        a, b, n = C.c_double(0), C.c_double(0), C.c_int64(0)
        self._ck(self._lib.csf_profile_read(self._h, C.byref(a), C.byref(b), C.byref(n)))
        return a.value, b.value, n.value

    def profile_kernels(self):
        """{"pair" | "road" | "agent" | "gather": (accumulated ms, launches)} over the sampled ticks; resets the sums.
        The pair kernel is timed on every sampled tick, the others on every 8th of them."""
        ms = (C.c_double * 4)()
        cnt = (C.c_int64 * 4)()
        self._ck(self._lib.csf_profile_kernels(self._h, ms, cnt))
        return {k: (ms[i], cnt[i]) for i, k in enumerate(("pair", "road", "agent", "gather"))}

    def profile_samples(self, capacity=65536, kernel="pair"):
        """microseconds of every sampled launch of `kernel` (pair, road, agent, gather) since the last reset (does not reset)"""
        out = np.zeros(int(capacity))
        n = C.c_int64(0)
        which = {"pair": 0, "road": 1, "agent": 2, "gather": 3}[kernel]
        self._ck(self._lib.csf_profile_samples_of(self._h, which, _ptr(out), int(capacity), C.byref(n)))
        return out[: n.value].copy()

    def profile_stats(self):
        """{kernel: {"median", "min", "max", "mean", "n"}} in microseconds over the sampled launches since the last reset -
        call BEFORE profile_kernels(), which resets.  Medians: one launch that met a clock step moves a mean of forty."""
        out = {}
        for k in ("pair", "road", "agent", "gather"):
            v = self.profile_samples(kernel=k)
            out[k] = None if v.size == 0 else {"median": float(np.median(v)), "min": float(v.min()), "max": float(v.max()),
                                               "mean": float(v.mean()), "n": int(v.size)}
        return out

    def count_pairs(self, detail=False):
        """(pair evaluations of one launch on the current snapshot or None, name of the engine's pair kernel); with
        detail=True the first item is the dict {evaluated, tested, full_passes, partial_passes}"""
        n = (C.c_int64 * 4)()
        name = C.c_char_p()
        self._ck(self._lib.csf_count_pairs(self._h, n, C.byref(name)))
        kernel = (name.value or b"").decode()
        if n[0] < 0:
            return None, kernel
        if detail:
            return dict(evaluated=n[0], tested=n[1], full_passes=n[2], partial_passes=n[3]), kernel
        return n[0], kernel

    def near_dropped(self):
        """near / field-of-view-edge pairs the pair kernels could not hand to their exact path since creation (expected 0)"""
        n = C.c_int64(0)
        self._ck(self._lib.csf_near_dropped(self._h, C.byref(n)))
        return n.value

    def small_ticks(self):
        """ticks run by the one-wave kernel of small populations (csf.h: csf_small_ticks)"""
        n = C.c_int64(0)
        self._ck(self._lib.csf_small_ticks(self._h, C.byref(n)))
        return n.value

    def mid_ticks(self):
        """ticks run as one launch each (csf_mid.hip: mid-size populations)"""
        v = C.c_int64()
        self._ck(self._lib.csf_mid_ticks(self._h, C.byref(v)))
        return v.value

    def chase_ticks(self):
        """ticks whose per-agent launch ran beside the pair launch (include/csf.h: csf_chase_ticks)"""
        v = C.c_int64()
        self._ck(self._lib.csf_chase_ticks(self._h, C.byref(v)))
        return v.value

    def chase_calibration(self):
        """(1 side by side / -1 in turn / 0 not measured yet, [us per tick in turn, side by side]) - include/csf.h: csf_chase_calibration"""
        st = C.c_int32()
        us = np.zeros(2)
        self._ck(self._lib.csf_chase_calibration(self._h, C.byref(st), _ptr(us)))
        return st.value, us.tolist()

    def holes_taken(self):
        """arrivals that took the slot of a road user who had left from nearby (include/csf.h: csf_holes_taken)"""
        v = C.c_int64()
        self._ck(self._lib.csf_holes_taken(self._h, C.byref(v)))
        return v.value

    def comm_stream_order(self):
        """('main' | 'second', [us per tick in stream order, on the second stream]) - where a sharded engine issues its
        all-gather, and what its communicator measured when it chose (zeros: CSF_COMM_STREAM decided, or not sharded)"""
        second = C.c_int32(0)
        us = np.zeros(2)
        self._ck(self._lib.csf_comm_stream_order(self._h, C.byref(second), _ptr(us)))
        return ("second" if second.value else "main"), [float(us[0]), float(us[1])]

    def profile_gather(self):
        """all-gather milliseconds accumulated over the launches of the last profile_read() (sharded engines)"""
        g = C.c_double(0)
        self._ck(self._lib.csf_profile_gather(self._h, C.byref(g)))
        return g.value
