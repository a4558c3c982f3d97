"""ctypes binding of libcsf_hip.so (include/csf.h).

The library is the only compute path of this package: if it is missing or cannot be loaded the import of
the engine fails loudly — there is no NumPy/CPU fallback.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# CSF_LIB selects another build of the same library (kernel A/B measurements, tools/ab.sh)
LIB_PATH = os.environ.get("CSF_LIB") or os.path.join(HERE, "libcsf_hip.so")

BICYCLE, TWOD, INVPEND, PLANARPOINT, PLANARBIKE, UNCONTROLLED, BALANCINGRIDER = 0, 1, 2, 3, 4, 5, 6
UNREGULATED, P2R = 0, 1
N_STATES = {BICYCLE: 5, TWOD: 5, INVPEND: 6, PLANARPOINT: 4, PLANARBIKE: 5, UNCONTROLLED: 4, BALANCINGRIDER: 8}
UNIQUE_ID_BYTES = 128
ST_SPLINE, ST_NAN, ST_NAVSTATE, ST_UNCONTROLLABLE = 1, 2, 4, 8

# every symbol include/csf.h declares (tests check that the library exports all of them)
SYMBOLS = (
    "csf_create", "csf_destroy", "csf_last_error", "csf_abi_version", "csf_add_agents", "csf_remove_agents",
    "csf_set_dest_queue", "csf_set_road_vertices", "csf_set_params", "csf_set_param_classes", "csf_set_agent_class", "csf_set_v_desired",
    "csf_set_priority_rule", "csf_push_state", "csf_num_agents", "csf_num_states", "csf_step", "csf_sync",
    "csf_calc_forces", "csf_apply_forces", "csf_replay_forces", "csf_dest_force", "csf_get_state", "csf_get_forces",
    "csf_get_force_parts", "csf_status", "csf_enable_history", "csf_get_history", "csf_pair_force",
    "csf_comm_unique_id", "csf_comm_init", "csf_shard_range", "csf_profile_enable", "csf_profile_read",
    "csf_far_radius", "csf_get_tick", "csf_profile_gather", "csf_profile_kernels", "csf_profile_samples",
    "csf_count_pairs", "csf_comm_init_loopback", "csf_step_group", "csf_untracked", "csf_update_destination",
    "csf_update_nav_state", "csf_set_dest_pointer", "csf_set_incremental", "csf_set_script", "csf_near_dropped", "csf_comm_stream_order", "csf_small_ticks", "csf_step_get_tick",
    "csf_get_integrator_state", "csf_set_integrator_state", "csf_mid_ticks", "csf_holes_taken",
    "csf_create_v", "csf_params_size", "csf_profile_samples_of", "csf_chase_ticks", "csf_chase_calibration", "csf_replace_agents",
)
ABI_VERSION = 9


class Params(C.Structure):
    """csf_params of include/csf.h (field order is ABI)."""

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


class EngineError(RuntimeError):
    pass


_lib = None


def load():
    """Load libcsf_hip.so; raises EngineError if the HIP extension is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EngineError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C cyclistsocialforce_amd/csrc`). This package has no CPU fallback."
        )
    try:
        L = C.CDLL(LIB_PATH)
    except OSError as exc:  # pragma: no cover - depends on the machine
        raise EngineError(f"cannot load {LIB_PATH}: {exc}") from exc
    vp, i32, i64, dp = C.c_void_p, C.c_int32, C.c_int64, C.c_void_p
    L.csf_create.restype = vp
    L.csf_create.argtypes = [C.POINTER(Params), i64, i32]
    L.csf_create_v.restype = vp
    L.csf_create_v.argtypes = [C.POINTER(Params), C.c_size_t, i32, i64, i32]
    L.csf_params_size.restype = C.c_size_t
    L.csf_params_size.argtypes = []
    L.csf_destroy.argtypes = [vp]
    L.csf_last_error.restype = C.c_char_p
    L.csf_last_error.argtypes = [vp]
    L.csf_abi_version.restype = i32
    L.csf_add_agents.argtypes = [vp, i64, dp, dp]
    L.csf_remove_agents.argtypes = [vp, i64, vp]
    L.csf_replace_agents.argtypes = [vp, i64, vp, i64, dp, dp, vp, dp]
    L.csf_set_dest_queue.argtypes = [vp, i64, vp, vp, dp, i32]
    L.csf_set_road_vertices.argtypes = [vp, i32, vp, dp, dp, dp]
    L.csf_set_params.argtypes = [vp, C.POINTER(Params)]
    L.csf_set_param_classes.argtypes = [vp, i32, C.POINTER(Params)]
    L.csf_set_agent_class.argtypes = [vp, i64, vp, vp]
    L.csf_set_v_desired.argtypes = [vp, i64, vp, dp]
    L.csf_set_priority_rule.argtypes = [vp, i32]
    L.csf_push_state.argtypes = [vp, i64, vp, dp]
    L.csf_num_agents.restype = i64
    L.csf_num_agents.argtypes = [vp]
    L.csf_num_states.restype = i32
    L.csf_num_states.argtypes = [vp]
    L.csf_step.argtypes = [vp, i64]
    L.csf_sync.argtypes = [vp]
    L.csf_calc_forces.argtypes = [vp]
    L.csf_apply_forces.argtypes = [vp, dp, dp]
    L.csf_replay_forces.argtypes = [vp, i64, dp, dp, vp, i32, i32, dp]
    L.csf_dest_force.argtypes = [vp, dp, dp]
    L.csf_get_state.argtypes = [vp, dp, vp, vp, C.POINTER(i64)]
    L.csf_get_forces.argtypes = [vp, dp, dp]
    L.csf_get_force_parts.argtypes = [vp, dp, dp, dp, dp]
    L.csf_status.argtypes = [vp, vp]
    L.csf_enable_history.argtypes = [vp, i32, i32]
    L.csf_get_history.argtypes = [vp, i64, i64, dp]
    L.csf_pair_force.argtypes = [vp, dp, i64, dp, dp, dp, i32, dp, dp]
    L.csf_comm_unique_id.argtypes = [vp]
    L.csf_comm_init.argtypes = [vp, vp, i32, i32]
    L.csf_shard_range.argtypes = [vp, C.POINTER(i64), C.POINTER(i64)]
    L.csf_profile_enable.argtypes = [vp, i32]
    L.csf_profile_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(i64)]
    L.csf_far_radius.argtypes = [vp, C.POINTER(C.c_double)]
    L.csf_profile_gather.argtypes = [vp, C.POINTER(C.c_double)]
    L.csf_get_tick.argtypes = [vp, dp, vp, vp, dp, dp, C.POINTER(i64)]
    pd = C.POINTER(C.c_double)
    L.csf_profile_kernels.argtypes = [vp, pd, C.POINTER(i64)]
    L.csf_profile_samples.argtypes = [vp, dp, i64, C.POINTER(i64)]
    L.csf_profile_samples_of.argtypes = [vp, i32, dp, i64, C.POINTER(i64)]
    L.csf_count_pairs.argtypes = [vp, C.POINTER(i64), C.POINTER(C.c_char_p)]   # int64 counts[4]
    L.csf_comm_init_loopback.argtypes = [C.POINTER(vp), i32]
    L.csf_step_group.argtypes = [C.POINTER(vp), i32, i64]
    L.csf_untracked.argtypes = [vp, vp]
    L.csf_update_destination.argtypes = [vp, i64, vp]
    L.csf_update_nav_state.argtypes = [vp, i64, vp, vp, dp, dp]
    L.csf_set_dest_pointer.argtypes = [vp, i64, vp, vp]
    L.csf_set_incremental.argtypes = [vp, i32]
    L.csf_set_script.argtypes = [vp, i64, vp, vp, dp]
    L.csf_near_dropped.argtypes = [vp, C.POINTER(i64)]
    L.csf_comm_stream_order.argtypes = [vp, C.POINTER(i32), dp]
    L.csf_small_ticks.argtypes = [vp, C.POINTER(i64)]
    L.csf_step_get_tick.argtypes = [vp, i64, vp, vp, vp, vp, vp, C.POINTER(i64)]
    L.csf_mid_ticks.argtypes = [vp, C.POINTER(i64)]
    L.csf_holes_taken.argtypes = [vp, C.POINTER(i64)]
    L.csf_chase_ticks.argtypes = [vp, C.POINTER(i64)]
    L.csf_chase_calibration.argtypes = [vp, C.POINTER(i32), dp]
    L.csf_get_integrator_state.argtypes = [vp, dp, dp, vp]
    L.csf_set_integrator_state.argtypes = [vp, i64, vp, dp, dp, vp]
    if L.csf_abi_version() != ABI_VERSION:
        raise EngineError(f"libcsf_hip.so has ABI {L.csf_abi_version()}, expected {ABI_VERSION}")
    if L.csf_params_size() != C.sizeof(Params):
        raise EngineError(f"csf_params: the library's has {L.csf_params_size()} bytes, this binding's {C.sizeof(Params)} "
                          "(include/csf.h and _ffi.Params are out of step)")
    _lib = L
    return L
