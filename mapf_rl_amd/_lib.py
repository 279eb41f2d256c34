"""ctypes binding of libmapf_env.so (include/mapf_env.h).  There is NO fallback: if the HIP library is
missing or fails to load, importing this module raises -- the product path never silently degrades to a
CPU implementation."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MAPF_LIB_OVERRIDE") or os.path.join(_HERE, "libmapf_env.so")  # override: A/B tuning of two builds only

OK = 0
ERR_INVALID_ARG, ERR_ACTION, ERR_OVERLAP, ERR_HIP, ERR_UNSUPPORTED, ERR_NO_SPACE, ERR_NOT_READY, ERR_TIMEOUT = (
    -1, -2, -3, -4, -5, -6, -7, -8)

# every symbol include/mapf_env.h declares: (name, restype, argtypes)
_vp, _i, _u64, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_float
SYMBOLS = [
    ("mapf_abi_version", _i, []),
    ("mapf_strerror", ctypes.c_char_p, [_i]),
    ("mapf_device_count", _i, []),
    ("mapf_create", _i, [_i, _i, _i, _i, _i, ctypes.POINTER(_vp)]),
    ("mapf_destroy", _i, [_vp]),
    ("mapf_set_reward_table", _i, [_vp, ctypes.POINTER(_f)]),
    ("mapf_load", _i, [_vp, _vp, _vp, _vp, _i, _vp]),
    ("mapf_set_agents", _i, [_vp, _vp, _vp]),
    ("mapf_reset_envs", _i, [_vp, _vp, _f, _u64, _vp]),
    ("mapf_stage_next", _i, [_vp, _f, _u64, _vp]),
    ("mapf_build_navi", _i, [_vp, _vp]),
    ("mapf_step", _i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("mapf_observe", _i, [_vp, _vp, _vp, _vp, _vp]),
    ("mapf_observe_masked", _i, [_vp, _vp, _vp, _vp, _vp, _vp]),
    ("mapf_obs_bits_row_dwords", _i, [_vp]),
    ("mapf_multi_create", _i, [_i] + [ctypes.POINTER(_vp)] * 9 + [ctypes.POINTER(_u64), ctypes.POINTER(_vp)]),
    ("mapf_multi_destroy", _i, [_vp]),
    ("mapf_multi_num_workgroups", _i, [_vp]),
    ("mapf_multi_step", _i, [_vp, _vp]),
    ("mapf_multi_reset", _i, [_vp, _f, _vp, _vp]),
    ("mapf_multi_observe_masked", _i, [_vp, _vp]),
    ("mapf_load_envs", _i, [_vp, _vp, _i, _vp, _vp, _vp, _vp]),
    ("mapf_get_navi", _i, [_vp, _vp, _vp]),
    ("mapf_get_agents", _i, [_vp, _vp, _vp]),
    ("mapf_get_goals", _i, [_vp, _vp, _vp]),
    ("mapf_get_maps", _i, [_vp, _vp, _vp]),
    ("mapf_get_steps", _i, [_vp, _vp, _vp]),
    ("mapf_check_status", _i, [_vp, _vp]),
    ("mapf_num_envs", _i, [_vp]),
    ("mapf_map_len", _i, [_vp]),
    ("mapf_num_agents", _i, [_vp]),
    ("mapf_obs_radius", _i, [_vp]),
    ("mapf_generate", _i, [_i, _i, _i, _f, _u64, _vp, _vp, _vp, ctypes.POINTER(ctypes.c_int32)]),
    # include/mapf_replay.h
    ("mapf_replay_create", _i, [_i, _i, _i, ctypes.POINTER(_vp)]),
    ("mapf_replay_destroy", _i, [_vp]),
    ("mapf_replay_row_dwords", _i, [_vp]),
    ("mapf_replay_capacity", _i, [_vp]),
    ("mapf_replay_ptr", _i, [_vp]),
    ("mapf_replay_size", ctypes.c_int64, [_vp]),
    ("mapf_replay_counter", ctypes.c_int64, [_vp, _i]),
    ("mapf_replay_tree_update", _i, [_vp, _vp, _vp, _i, ctypes.c_double, _vp]),
    ("mapf_replay_state", _i, [_vp, ctypes.POINTER(ctypes.c_int64), _vp]),
    ("mapf_replay_tree_sample", _i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    ("mapf_replay_tree_read", _i, [_vp, _vp, _vp]),
    ("mapf_replay_add", _i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("mapf_replay_add_many", _i, [_vp, _i, _i, _i] + [_vp] * 10),
    ("mapf_replay_add_many_env", _i, [_vp, _i, _vp, _i] + [_vp] * 10),
    ("mapf_actor_explore_multi", _i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("mapf_actor_record_multi", _i, [_i, _i, _i, _i] + [_vp] * 17),
    ("mapf_actor_rewind_multi", _i, [_i, _i, _i] + [_vp] * 9),
    ("mapf_actor_log_multi", _i, [_i, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(_vp), ctypes.POINTER(_vp), _i, _vp, _vp, _vp, _vp]),
    ("mapf_actor_record", _i, [_i] * 6 + [_vp] * 16),
    ("mapf_actor_rewind", _i, [_i] * 5 + [_vp] * 6),
    ("mapf_actor_explore", _i, [_i, _i, _vp, _vp, _vp, _vp, ctypes.c_uint64, ctypes.c_uint64, _vp]),
    ("mapf_actor_explore_dev", _i, [_i, _i, _vp, _vp, _vp, _vp, ctypes.c_uint64, ctypes.c_uint64, _vp, _vp]),
    ("mapf_actor_iteration_tail", _i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_float, ctypes.c_uint64, _vp]),
    ("mapf_actor_log", _i, [_i, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    ("mapf_obs_changed", _i, [_vp, _vp, ctypes.c_int64, _vp, _vp, _vp, _vp]),
    ("mapf_replay_is_weights", _i, [_vp, _i, ctypes.c_double, _vp, _vp]),
    ("mapf_replay_sample", _i, [_vp, _vp, _i, _i] + [_vp] * 12),
    ("mapf_replay_update_priorities", _i, [_vp, _vp, _vp, _i, _vp, _vp]),
    # include/mapf_dqn.h
    ("mapf_bias_res_relu_fwd", _i, [_vp, _vp, _vp, ctypes.c_int64, _i, _vp]),
    ("mapf_bias_res_relu_bwd", _i, [_vp, _vp, _vp, _vp, ctypes.c_int64, _i, _vp]),
    ("mapf_recurrent_infer", _i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, ctypes.c_int64, _vp]),
    ("mapf_recurrent_infer_multi", _i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    ("mapf_recurrent_forward_save", _i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, ctypes.POINTER(_vp), _vp, ctypes.c_int64, _vp]),
    ("mapf_recurrent_backward", _i, [ctypes.POINTER(_vp), _vp, _vp, _vp, _i, _i, _i, ctypes.POINTER(_vp), _vp, ctypes.c_int64, _vp]),
    ("mapf_recurrent_infer_packed", _i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, ctypes.c_int64, _i, _vp]),
    ("mapf_recurrent_forward_save_packed", _i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, ctypes.POINTER(_vp), _vp, ctypes.c_int64, _i, _vp]),
    ("mapf_recurrent_backward_packed", _i, [ctypes.POINTER(_vp), _vp, _vp, _vp, _i, _i, _i, ctypes.POINTER(_vp), _vp, ctypes.c_int64, _i, _vp]),
    ("mapf_comm_mask", _i, [_vp, _i, _i, _i, _i, _vp, _vp, _i, _vp]),
    ("mapf_q_head", _i, [_vp, ctypes.c_int64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("mapf_input_proj_pack", _i, [_vp, _vp, _vp]),
    ("mapf_input_proj_rows", _i, [_vp, ctypes.c_int64, _vp, _vp, _vp, _vp, _vp]),
    ("mapf_comm_mask_multi", _i, [_vp, _i, _vp, _i, _i, _vp, _vp, _i, _vp]),
    ("mapf_window_relevance", _i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    ("mapf_encoder_pack", _i, [ctypes.POINTER(_vp), ctypes.POINTER(_vp), _i, _vp, _vp, _vp]),
    ("mapf_encoder_forward", _i, [_vp, _i, ctypes.c_int64, _vp, _vp, _vp, _vp]),
    ("mapf_encoder_forward_rows", _i, [_vp, ctypes.c_int64, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("mapf_encoder_pack_bwd", _i, [ctypes.POINTER(_vp), _i, _vp, _vp]),
    ("mapf_encoder_backward_data", _i, [_vp, ctypes.c_int64, _vp, _vp, _vp, _vp, _vp]),
    ("mapf_encoder_wgrad0", _i, [_vp, _vp, _i, ctypes.c_int64, _vp, _vp, _vp]),
    ("mapf_encoder_backward", _i, [_vp, _vp, ctypes.c_int64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("mapf_encoder_wgrad", _i, [_vp, _vp, ctypes.c_int64, _vp, _vp, _vp]),
    ("mapf_encoder_wgrad_multi", _i, [_vp, ctypes.c_int64, _vp, ctypes.c_int64, _i, _i, ctypes.c_int64, _vp, _vp, _vp, _vp]),
    ("mapf_encoder_forward_bounded", _i, [_vp, _i, ctypes.c_int64, _vp, _vp, _vp, _vp, _vp]),
    ("mapf_encoder_forward_save_bounded", _i, [_vp, _i, ctypes.c_int64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("mapf_encoder_backward_bounded", _i, [_vp, _vp, ctypes.c_int64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("mapf_encoder_wgrad0_bounded", _i, [_vp, _vp, _i, ctypes.c_int64, _vp, _vp, _vp, _vp]),
    ("mapf_plan_totals", _i, [_vp, _i, _i, _vp, _vp]),
    ("mapf_encoder_forward_save", _i, [_vp, _i, ctypes.c_int64, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("mapf_plan_mark", _i, [_vp, ctypes.c_int64, ctypes.c_int64, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("mapf_obs_dup", _i, [_i, _i, _i, _i, _vp, ctypes.c_int64, ctypes.c_int64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("mapf_dedup_sum", _i, [_i, _i, _i, ctypes.c_int64, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("mapf_plan_rows", _i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, ctypes.c_int64, ctypes.c_int64, _vp, _i, _vp, ctypes.c_int64, ctypes.c_int64,
                            _vp, _vp, _vp, ctypes.c_int64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("mapf_plan_rows_padded", _i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, ctypes.c_int64, ctypes.c_int64, _vp, _i, _vp, ctypes.c_int64, ctypes.c_int64,
                            _vp, _vp, _vp, ctypes.c_int64, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int64, ctypes.c_int64, _vp]),
    ("mapf_rows_scatter", _i, [_vp, _vp, _vp, ctypes.c_int64, _i, _i, _vp]),
    ("mapf_dqn_head_loss", _i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), _f,
                                _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.POINTER(_vp), _vp]),
    ("mapf_recurrent_pack", _i, [ctypes.POINTER(_vp), _vp, _vp, _vp, _vp]),
    ("mapf_recurrent_bias_grads", _i, [_vp, _i, ctypes.POINTER(_vp), _vp]),
    ("mapf_adam_step", _i, [ctypes.c_int64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _f, ctypes.c_int64, _f, _vp]),
    ("mapf_adam_step_dev", _i, [ctypes.c_int64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _vp, _f, _vp]),
    ("mapf_to_bf16", _i, [_vp, _vp, ctypes.c_int64, _vp]),
    ("mapf_zero_rows", _i, [ctypes.POINTER(_vp), ctypes.POINTER(_i), _i, ctypes.c_int64, ctypes.c_int64, _vp]),
    ("mapf_zero_rows_from", _i, [ctypes.POINTER(_vp), ctypes.POINTER(_i), _i, _vp, ctypes.c_int64, _vp]),
    ("mapf_tall_tn_plan", _i, [ctypes.c_int64, _i, _i, ctypes.POINTER(_i), ctypes.POINTER(_i), ctypes.POINTER(ctypes.c_int64)]),
    ("mapf_tall_tn", _i, [_vp, ctypes.c_int64, _vp, ctypes.c_int64, ctypes.c_int64, _i, _i, _i, _vp, _vp, _i, _vp, ctypes.c_int64, _vp]),
    ("mapf_sum_parts", _i, [ctypes.POINTER(_vp), ctypes.POINTER(_vp), _i, _i, ctypes.c_int64, _vp, _vp]),
    ("mapf_encoder_small_grads", _i, [_vp, ctypes.c_int64, _vp, _vp, ctypes.c_int64, _vp, _vp, _i, _vp, _vp, _vp]),
    ("mapf_latent_grad_pack", _i, [_vp, _vp, _vp]),
    ("mapf_latent_grad_rows", _i, [_vp, ctypes.c_int64, _vp, _vp, _vp]),
    # include/mapf_search.h
    ("mapf_find_path", _i, [_i, _i, _vp, _vp, _vp, ctypes.c_double, _i, _vp, ctypes.POINTER(_i), ctypes.POINTER(_i)]),
    ("mapf_distance_field", _i, [_i, _vp, _i, _i, _vp]),
]


class MapfError(RuntimeError):
    def __init__(self, status, where=""):
        self.status = status
        msg = lib.mapf_strerror(status).decode() if lib is not None else str(status)
        super().__init__("%s%s (status %d)" % (where + ": " if where else "", msg, status))


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "mapf_rl_amd: %s is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        fn = getattr(L, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if L.mapf_abi_version() != 2:
        raise ImportError("mapf_rl_amd: ABI version mismatch")
    return L


lib = None
lib = _load()


def check(status, where=""):
    if status != OK:
        raise MapfError(status, where)


class ActorState(ctypes.Structure):
    """include/mapf_replay.h: mapf_actor_state (the persistent device buffers of one vectorised actor)."""
    _fields_ = [(k, ctypes.c_int32) for k in ("num_envs", "num_agents", "local_steps", "env_row_dwords", "row_dwords", "max_agents", "log_size", "reserved")] + \
               [(k, ctypes.c_void_p) for k in ("lb_q", "lb_act", "lb_rew", "lb_hid", "lb_comm", "lb_obs", "t", "finished", "obs_bits", "stat_mask", "stat_log",
                                               "counters", "eps", "policy_actions", "act8", "obs", "pos", "reward_class", "reward", "done")]
