"""A caller that gets its arguments wrong, through the raw C ABI (ctypes on libcsf_hip.so; run as a script in a process of its own by
tests/test_gpu_abi_errors.py, so that a crash is a failed test and not the end of the test run).

Two engines of the same population: one receives every hostile call of the table below, the other none.  Every hostile call must
come back with a negative code (include/csf.h: CSF_E_*) and leave a message in csf_last_error; afterwards both engines are stepped
and must hold the same states bit for bit - a refused call has no side effect.  Prints one JSON object.

NOT in the table, because the ABI accepts them by design: csf_destroy(NULL) (as free(NULL)); read-backs whose output pointers are all
NULL (each is optional); an index listed twice in csf_remove_agents (sorted and made unique); csf_comm_init(id = NULL) for a world of
one; lengths beyond n_ticks in csf_replay_forces (the sequence ends with the call); and numbers that are not numbers - a NaN position, speed, destination, force or road vertex is DATA: the reference takes it too,
and it comes back as CSF_ST_NAN in csf_status (tests/test_gpu_parity.py).  A NaN inside csf_params is refused."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cyclistsocialforce_amd import _ffi, parameters  # noqa: E402

N, CAP = 48, 64
L = _ffi.load()
pod = parameters.default_pod(sys.argv[1] if len(sys.argv) > 1 else "twod")
NS = _ffi.N_STATES[pod.model]


def arr(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


_alive = []                               # (every array a pointer is taken of stays alive to the end of the script)


def ptr(a):
    if a is None:
        return None
    _alive.append(a)
    return a.ctypes.data_as(C.c_void_p)


def make():
    h = L.csf_create_v(C.byref(pod), C.sizeof(pod), _ffi.ABI_VERSION, CAP, 0)
    assert h, L.csf_last_error(None)
    rng = np.random.default_rng(5)
    s0 = np.zeros((N, NS))
    s0[:, 0], s0[:, 1] = rng.uniform(0, 30, N), rng.uniform(0, 30, N)
    s0[:, 2], s0[:, 3] = rng.uniform(-np.pi, np.pi, N), rng.uniform(3, 6, N)
    vd = np.full(N, 5.0)
    assert L.csf_add_agents(h, N, ptr(s0), ptr(vd)) == 0
    off = arr(np.arange(N + 1) * 3, np.int64)
    d = np.array([20.0, 39.0, 40.0])
    q = np.zeros((N, 3, 3))
    q[:, :, 0] = s0[:, 0:1] + d * np.cos(s0[:, 2:3])
    q[:, :, 1] = s0[:, 1:2] + d * np.sin(s0[:, 2:3])
    assert L.csf_set_dest_queue(h, N, ptr(arr(np.arange(N), np.int32)), ptr(off), ptr(arr(q.reshape(-1, 3), np.float64)), 1) == 0
    assert L.csf_step(h, 3) == 0 and L.csf_sync(h) == 0
    return h


def state(h):
    s = np.zeros((N, NS))
    dp, zn, tk = np.zeros(N, np.int32), np.zeros((N, 3), np.uint8), C.c_int64(0)      # (znav: one-hot, [n, 3])
    rc = L.csf_get_state(h, ptr(s), ptr(dp), ptr(zn), C.byref(tk))
    assert rc == 0, rc
    return s, dp, zn, tk.value


victim, twin = make(), make()
i32a = lambda *v: arr(v, np.int32)        # noqa: E731
i64a = lambda *v: arr(v, np.int64)        # noqa: E731
f64 = lambda *shape: np.zeros(shape)      # noqa: E731
good_idx, s1, big = i32a(0), f64(1, NS), 2 ** 31 - 1
out_n, out_ns = f64(N), f64(N, NS)
i64v, i32v, dbl = C.c_int64(0), C.c_int32(0), C.c_double(0)
nullh = None
pod_bad = type(pod).from_buffer_copy(pod)
pod_bad.t_s = -1.0
pod_nan = type(pod).from_buffer_copy(pod)
pod_nan.f_0 = float("nan")
pod_model = type(pod).from_buffer_copy(pod)
pod_model.model = 77

# (name, arguments) - every one of them wrong in exactly one way
calls = [
    # a null handle, everything else in order
    ("csf_add_agents", (nullh, 1, ptr(s1), ptr(f64(1)))), ("csf_remove_agents", (nullh, 1, ptr(good_idx))),
    ("csf_step", (nullh, 1)), ("csf_sync", (nullh,)), ("csf_calc_forces", (nullh,)), ("csf_get_state", (nullh, ptr(out_ns), None, None, None)),
    ("csf_get_forces", (nullh, ptr(out_n), ptr(out_n))), ("csf_status", (nullh, ptr(np.zeros(N, np.uint32)))),
    ("csf_set_params", (nullh, C.byref(pod))), ("csf_push_state", (nullh, 1, ptr(good_idx), ptr(s1))),
    ("csf_set_dest_queue", (nullh, 1, ptr(good_idx), ptr(i64a(0, 1)), ptr(f64(1, 3)), 1)),
    ("csf_step_get_tick", (nullh, 1, ptr(out_ns), None, None, ptr(out_n), ptr(out_n), None)),
    ("csf_profile_read", (nullh, C.byref(dbl), C.byref(dbl), C.byref(i64v))), ("csf_chase_ticks", (nullh, C.byref(i64v))),
    ("csf_shard_range", (nullh, C.byref(i64v), C.byref(i64v))), ("csf_far_radius", (nullh, C.byref(dbl))),
    # counts and sizes
    ("csf_add_agents", (victim, -1, ptr(s1), ptr(f64(1)))), ("csf_add_agents", (victim, CAP, ptr(f64(CAP, NS)), ptr(f64(CAP)))),   # beyond the capacity
    ("csf_add_agents", (victim, 1, None, ptr(f64(1)))), ("csf_remove_agents", (victim, -3, ptr(good_idx))), ("csf_remove_agents", (victim, 1, None)),
    ("csf_step", (victim, -1)), ("csf_enable_history", (victim, 0, 16)), ("csf_enable_history", (victim, 1, -2)),
    ("csf_profile_samples_of", (victim, 9, ptr(f64(8)), 8, C.byref(i64v))), ("csf_profile_samples_of", (victim, 0, None, 8, C.byref(i64v))),
    ("csf_profile_samples", (victim, ptr(f64(8)), -1, C.byref(i64v))),
    # indices out of range, twice the same, negative
    ("csf_remove_agents", (victim, 1, ptr(i32a(N)))), ("csf_remove_agents", (victim, 1, ptr(i32a(-1)))),
    ("csf_remove_agents", (victim, 1, ptr(i32a(big)))),
    ("csf_push_state", (victim, 1, ptr(i32a(N)), ptr(s1))), ("csf_push_state", (victim, 1, ptr(i32a(-7)), ptr(s1))), ("csf_push_state", (victim, 1, ptr(good_idx), None)),
    ("csf_set_v_desired", (victim, 1, ptr(i32a(N + 5)), ptr(f64(1)))), ("csf_set_agent_class", (victim, 1, ptr(good_idx), ptr(i32a(3)))),   # no such parameter set
    ("csf_set_agent_class", (victim, 1, ptr(i32a(-1)), ptr(i32a(0)))),
    ("csf_update_destination", (victim, 1, ptr(i32a(N)))), ("csf_set_dest_pointer", (victim, 1, ptr(good_idx), ptr(i32a(99)))),
    ("csf_set_dest_pointer", (victim, 1, ptr(i32a(-2)), ptr(i32a(0)))),
    ("csf_update_nav_state", (victim, 1, ptr(i32a(N)), ptr(i32a(0)), ptr(f64(1)), ptr(f64(1)))),
    ("csf_set_integrator_state", (victim, 1, ptr(i32a(N)), ptr(f64(1, 8)), ptr(f64(1)), None)),
    # queues: a road user that is not there, offsets that run backwards, an empty queue, no rows
    ("csf_set_dest_queue", (victim, 1, ptr(i32a(N)), ptr(i64a(0, 1)), ptr(f64(1, 3)), 1)),
    ("csf_set_dest_queue", (victim, 2, ptr(i32a(0, 1)), ptr(i64a(0, 2, 1)), ptr(f64(2, 3)), 1)),
    ("csf_set_dest_queue", (victim, 1, ptr(good_idx), ptr(i64a(0, 0)), ptr(f64(1, 3)), 1)),
    ("csf_set_dest_queue", (victim, 1, ptr(good_idx), ptr(i64a(0, 1 << 60)), ptr(f64(1, 3)), 1)),     # more rows than a vector can hold: refused, or caught at the boundary
    ("csf_set_dest_queue", (victim, 1, ptr(good_idx), ptr(i64a(0, 1)), None, 1)), ("csf_set_dest_queue", (victim, 1, ptr(good_idx), None, ptr(f64(1, 3)), 1)),
    # road: offsets that run backwards, an edge of one vertex ... none, a NaN coordinate, a negative exponent base
    ("csf_set_road_vertices", (victim, 1, ptr(i64a(2, 0)), ptr(f64(2, 2)), ptr(np.full(1, .05)), ptr(np.full(1, 3.0)))),
    ("csf_set_road_vertices", (victim, -1, ptr(i64a(0, 2)), ptr(f64(2, 2)), ptr(np.full(1, .05)), ptr(np.full(1, 3.0)))),
    ("csf_set_road_vertices", (victim, 1, ptr(i64a(0, 2)), None, ptr(np.full(1, .05)), ptr(np.full(1, 3.0)))),
    # parameters: a time step below zero, a NaN, a vehicle class that does not exist, none at all; a rule that does not exist
    ("csf_set_params", (victim, C.byref(pod_bad))), ("csf_set_params", (victim, C.byref(pod_nan))), ("csf_set_params", (victim, C.byref(pod_model))),
    ("csf_set_params", (victim, None)), ("csf_set_param_classes", (victim, 0, C.byref(pod))), ("csf_set_param_classes", (victim, 2, None)),
    ("csf_set_param_classes", (victim, 100000, C.byref(pod))), ("csf_set_priority_rule", (victim, 5)),
    # forces from elsewhere: none, or not numbers
    ("csf_apply_forces", (victim, None, ptr(out_n))),
    ("csf_replay_forces", (victim, -1, ptr(f64(1, N)), ptr(f64(1, N)), None, 0, 0, ptr(f64(1, N, NS)))),
    ("csf_replay_forces", (victim, 2, None, ptr(f64(2, N)), None, 0, 0, ptr(f64(2, N, NS)))),
    # read-backs with nowhere to write
    ("csf_status", (victim, None)),
    ("csf_dest_force", (victim, None, ptr(out_n))), ("csf_untracked", (victim, None)), ("csf_far_radius", (victim, None)),
    ("csf_pair_force", (victim, None, 1, ptr(f64(1)), ptr(f64(1)), ptr(f64(1)), 0, ptr(f64(1)), ptr(f64(1)))),
    ("csf_pair_force", (victim, ptr(f64(4)), -1, ptr(f64(1)), ptr(f64(1)), ptr(f64(1)), 0, ptr(f64(1)), ptr(f64(1)))),
    ("csf_count_pairs", (victim, None, None)), ("csf_chase_ticks", (victim, None)),
    # communicators: a rank outside the world, a world of none, a group that is no group
    ("csf_comm_init", (victim, ptr(np.zeros(128, np.uint8)), 3, 2)), ("csf_comm_init", (victim, ptr(np.zeros(128, np.uint8)), 0, 0)),
    ("csf_comm_init", (victim, None, 0, 2)), ("csf_comm_unique_id", (None,)), ("csf_comm_init_loopback", (None, 2)), ("csf_step_group", (None, 2, 1)),
    # creation: no parameters, a struct of another size, another ABI, no capacity, a device that is not there
    ("csf_create_v", (None, C.sizeof(pod), _ffi.ABI_VERSION, CAP, 0)), ("csf_create_v", (C.byref(pod), C.sizeof(pod) - 8, _ffi.ABI_VERSION, CAP, 0)),
    ("csf_create_v", (C.byref(pod), C.sizeof(pod), _ffi.ABI_VERSION - 1, CAP, 0)), ("csf_create_v", (C.byref(pod), C.sizeof(pod), _ffi.ABI_VERSION, 0, 0)),
    ("csf_create_v", (C.byref(pod), C.sizeof(pod), _ffi.ABI_VERSION, -5, 0)), ("csf_create_v", (C.byref(pod), C.sizeof(pod), _ffi.ABI_VERSION, 1 << 62, 0)),
    ("csf_create_v", (C.byref(pod), C.sizeof(pod), _ffi.ABI_VERSION, (1 << 30) + 1, 0)), ("csf_create_v", (C.byref(pod), C.sizeof(pod), _ffi.ABI_VERSION, CAP, 4096)),
    ("csf_create_v", (C.byref(pod_model), C.sizeof(pod), _ffi.ABI_VERSION, CAP, 0)), ("csf_create", (None, CAP, 0)),
]
L.csf_create_v.restype = C.c_void_p
L.csf_create_v.argtypes = [C.c_void_p, C.c_size_t, C.c_int32, C.c_int64, C.c_int32]
L.csf_create.argtypes = [C.c_void_p, C.c_int64, C.c_int32]

# ---- wrong in its context: the victim has a history ring of 8 samples (stride 2) from here on, the twin too -------------------
for h in (victim, twin):
    assert L.csf_enable_history(h, 2, 8) == 0 and L.csf_step(h, 6) == 0          # three samples recorded
eng2 = make()                                   # a third engine, one road user short: for groups that are no groups
assert L.csf_remove_agents(eng2, 1, ptr(i32a(N - 1))) == 0
grp_same = (C.c_void_p * 2)(victim, victim)
grp_null = (C.c_void_p * 2)(victim, None)
grp_diff = (C.c_void_p * 2)(victim, eng2)       # (the members of a group hold the same population)
calls += [
    ("csf_get_history", (victim, -1, 2, ptr(f64(2, N, NS)))), ("csf_get_history", (victim, 0, 50, ptr(f64(50, N, NS)))),       # more than was recorded
    ("csf_get_history", (victim, 2, 5, ptr(f64(5, N, NS)))), ("csf_get_history", (victim, 0, -1, ptr(f64(1, N, NS)))), ("csf_get_history", (victim, 0, 2, None)),
    ("csf_enable_history", (victim, 2, 0)), ("csf_get_history", (eng2, 0, 1, ptr(f64(1, N, NS)))),                             # no history there
    ("csf_comm_init_loopback", (grp_same, 2)), ("csf_comm_init_loopback", (grp_null, 2)), ("csf_comm_init_loopback", (grp_diff, 2)),
    ("csf_comm_init_loopback", (grp_diff, 0)), ("csf_comm_init_loopback", (grp_diff, -2)),
    ("csf_step_group", (grp_diff, 2, 1)),                                        # no group was formed
    ("csf_replay_forces", (victim, 2, ptr(f64(2, N)), ptr(f64(2, N)), ptr(i32a(*([-1] * N))), 0, 1, None)),
    ("csf_replay_forces", (victim, 2, ptr(f64(2, N)), ptr(f64(2, N)), None, 0, -1, ptr(f64(2, N, NS)))),   # a stride below zero with somewhere to write
    ("csf_set_script", (victim, 1, ptr(good_idx), ptr(i64a(0, 2)), ptr(f64(2, 4)))),                      # not an UncontrolledVehicle
    ("csf_replace_agents", (victim, 1, ptr(i32a(N)), 0, None, None, None, None)),
    ("csf_replace_agents", (victim, 0, None, 1, ptr(s1), ptr(f64(1)), ptr(i64a(0, 0)), ptr(f64(1, 3)))),    # an arrival without a queue row
    ("csf_replace_agents", (victim, 0, None, 1, ptr(s1), ptr(f64(1)), None, None)),
    ("csf_replace_agents", (victim, 0, None, CAP, ptr(f64(CAP, NS)), ptr(f64(CAP)), ptr(i64a(*range(CAP + 1))), ptr(f64(CAP, 3)))),   # beyond the capacity
    ("csf_replace_agents", (victim, 0, None, 2, ptr(f64(2, NS)), ptr(f64(2)), ptr(i64a(0, 2, 1)), ptr(f64(2, 3)))),
    ("csf_replace_agents", (victim, -1, ptr(good_idx), 0, None, None, None, None)),
]

accepted, silent = [], []
only = [int(t) for t in sys.argv[2].split(",")] if len(sys.argv) > 2 else None     # (to find the call behind a crash: its number)
for k, (name, args) in enumerate(calls):
    if only is not None and k not in only:
        continue
    rc = getattr(L, name)(*args)
    if rc is None:
        rc = 0                                  # (a NULL handle from csf_create*)
    if name.startswith("csf_create"):
        if rc:                               # a handle: the call was accepted
            accepted.append([k, name, "a handle"])
            L.csf_destroy(rc)
        elif not L.csf_last_error(None):
            silent.append([k, name])
        continue
    if rc >= 0:
        accepted.append([k, name, int(rc)])
    elif args and args[0] == victim and not L.csf_last_error(victim):
        silent.append([k, name])

# the engine that was called names still works, and holds what its twin holds
for h in (victim, twin, eng2):
    assert L.csf_step(h, 7) == 0 and L.csf_sync(h) == 0, L.csf_last_error(h)
sv, st = state(victim), state(twin)
same = bool(all(np.array_equal(a, b) for a, b in zip(sv[:3], st[:3])) and sv[3] == st[3])
flags = np.zeros(N, np.uint32)
L.csf_status(victim, ptr(flags))
print(json.dumps({"calls": len(calls), "accepted": accepted, "silent": silent, "same_as_twin": same, "finite": bool(np.isfinite(sv[0]).all()),
                  "status_flags": int((flags != 0).sum()), "tick": sv[3], "eng2_tick": state(eng2)[3], "agents": int(L.csf_num_agents(victim))}))
L.csf_destroy(victim)
L.csf_destroy(twin)
L.csf_destroy(eng2)
