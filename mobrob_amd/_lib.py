"""ctypes binding of libmobrob_ppo.so (C ABI declared in include/mobrob_ppo.h).

There is NO CPU fallback: if the shared library is missing or no gfx950 device is visible the
product raises.  (The CPU oracle under oracle/ is test infrastructure and is never imported here.)
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MOBROB_PPO_LIB: another build of the same library (A/B timing of a compile-time switch, stamp / ablation builds)
LIB_PATH = os.environ.get("MOBROB_PPO_LIB") or os.path.join(_HERE, "libmobrob_ppo.so")
ABI_VERSION = 3

OK, ERR_INVALID, ERR_HIP, ERR_STATE, ERR_NO_DEVICE = 0, -1, -2, -3, -4

BUF = dict(obs=0, actions=1, rewards=2, episode_starts=3, values=4, log_probs=5, advantages=6, returns=7,
           params=8, grads=9, advstat=10, last_values=11, last_dones=12, clipped_actions=13, episode_start_state=14,
           terminal_obs=15, terminal_values=16, truncated=17, env_state=18, grad_exchange=19, sde_noise=20)
HYPER = dict(learning_rate=0, clip_range=1, clip_range_vf=2, target_kl=3, ent_coef=4, vf_coef=5, epoch_kernel=6)
KERNEL_IDS = dict(act=0, gae=1, train_grad=2, apply=3, env=4, grad_reduce=5, allreduce=6)


class Config(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("obs_dim", C.c_int32), ("act_dim", C.c_int32),
        ("pi_hidden", C.c_int32 * 2), ("vf_hidden", C.c_int32 * 2),
        ("n_envs", C.c_int32), ("n_steps", C.c_int32), ("batch_size", C.c_int32), ("n_epochs", C.c_int32),
        ("gamma", C.c_double), ("gae_lambda", C.c_double), ("clip_range", C.c_double), ("ent_coef", C.c_double),
        ("vf_coef", C.c_double), ("max_grad_norm", C.c_double), ("learning_rate", C.c_double),
        ("adam_beta1", C.c_double), ("adam_beta2", C.c_double), ("adam_eps", C.c_double),
        ("action_low", C.c_double), ("action_high", C.c_double),
        ("normalize_advantage", C.c_int32), ("seed", C.c_uint64), ("device_id", C.c_int32),
        ("rank", C.c_int32), ("world_size", C.c_int32), ("fast_kernels", C.c_int32), ("rollout_graph", C.c_int32), ("rollout_persistent", C.c_int32),
        ("activation", C.c_int32), ("forward_x3", C.c_int32), ("pi_hidden3", C.c_int32), ("vf_hidden3", C.c_int32),
        ("reserved", C.c_int32 * 1), ("pi_hidden_ext", C.c_int32 * 5), ("vf_hidden_ext", C.c_int32 * 5),
        ("use_sde", C.c_int32), ("sde_sample_freq", C.c_int32), ("sde_full_std", C.c_int32), ("sde_use_expln", C.c_int32),
    ]


class GoalEnv(C.Structure):  # mobrob_goal_env_t
    _fields_ = [("pos_dim", C.c_int32), ("terminate_on_goal", C.c_int32), ("time_limit", C.c_int32),
                ("dt", C.c_float), ("extent", C.c_float), ("reach_radius", C.c_float), ("goal_bonus", C.c_float),
                ("extra_bonus", C.c_float), ("obs_noise", C.c_float), ("mix", (C.c_float * 32) * 3)]


class DroneParams(C.Structure):  # mobrob_drone_params_t
    _fields_ = [(k, C.c_float) for k in ("mass", "g", "dt", "max_thrust", "max_xy_torque", "max_z_torque",
                                         "max_roll_pitch", "tune_fac")]


class EpisodeStats(C.Structure):  # mobrob_episode_stats_t
    _fields_ = [("episodes", C.c_int64), ("return_sum", C.c_double), ("length_sum", C.c_double), ("goals", C.c_int64)]


class TrainStats(C.Structure):
    _fields_ = [(k, C.c_float) for k in
                ("policy_loss", "value_loss", "entropy_loss", "loss", "approx_kl", "clip_fraction", "grad_norm")] + \
               [("n_minibatches", C.c_int32)]


# every symbol include/mobrob_ppo.h declares: name -> (restype, argtypes)
_P, _F, _U8, _I64 = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.POINTER(C.c_int64)
SYMBOLS = {
    "mobrob_ppo_default_config": (None, [C.POINTER(Config)]),
    "mobrob_ppo_create": (C.c_int, [C.POINTER(Config), C.POINTER(_P)]),
    "mobrob_ppo_destroy": (None, [_P]),
    "mobrob_ppo_device_bytes": (C.c_int, [C.POINTER(Config), C.POINTER(C.c_size_t)]),
    "mobrob_ppo_create_in_arena": (C.c_int, [C.POINTER(Config), _P, C.c_size_t, C.POINTER(_P)]),
    "mobrob_ppo_device_alloc": (_P, [C.c_int32, C.c_size_t]),
    "mobrob_ppo_device_free": (None, [_P]),
    "mobrob_ppo_last_error": (C.c_char_p, []),
    "mobrob_ppo_abi_version": (C.c_int, []),
    "mobrob_ppo_set_stream": (C.c_int, [_P, _P]),
    "mobrob_ppo_synchronize": (C.c_int, [_P]),
    "mobrob_ppo_host_alloc": (_P, [C.c_size_t]),
    "mobrob_ppo_host_free": (None, [_P]),
    "mobrob_ppo_host_register": (C.c_int, [_P, C.c_size_t]),
    "mobrob_ppo_host_unregister": (C.c_int, [_P]),
    "mobrob_ppo_param_count": (C.c_int64, [_P]),
    "mobrob_ppo_get_params": (C.c_int, [_P, _F, C.c_int64]),
    "mobrob_ppo_set_params": (C.c_int, [_P, _F, C.c_int64]),
    "mobrob_ppo_get_optimizer_state": (C.c_int, [_P, _F, _F, C.c_int64, _I64]),
    "mobrob_ppo_set_optimizer_state": (C.c_int, [_P, _F, _F, C.c_int64, C.c_int64]),
    "mobrob_ppo_rollout_begin": (C.c_int, [_P]),
    "mobrob_ppo_act": (C.c_int, [_P, _F, _F, _F, _F, _F, _F]),
    "mobrob_ppo_store": (C.c_int, [_P, _F, _U8, _U8, _F]),
    "mobrob_ppo_act_part": (C.c_int, [_P, C.c_int32, C.c_int32, _F, _F]),
    "mobrob_ppo_wait_part": (C.c_int, [_P, C.c_int32]),
    "mobrob_ppo_store_part": (C.c_int, [_P, C.c_int32, C.c_int32, _F, _U8, _U8, _F, _F]),
    "mobrob_ppo_collect_host": (C.c_int, [_P, _P, _P, C.c_int32, _F, _F, _F, _U8, _U8, _F]),
    "mobrob_ppo_finish_rollout": (C.c_int, [_P, _F, _U8]),
    "mobrob_ppo_collect_synthetic": (C.c_int, [_P, C.c_float, C.c_int32]),
    "mobrob_ppo_collect_goal_env": (C.c_int, [_P, C.POINTER(GoalEnv)]),
    "mobrob_ppo_episode_stats": (C.c_int, [_P, C.POINTER(EpisodeStats), C.c_int32]),
    "mobrob_ppo_episode_records": (C.c_int, [_P, _F, C.c_int32]),
    "mobrob_ppo_train": (C.c_int, [_P, _I64, C.POINTER(TrainStats)]),
    "mobrob_ppo_train_enqueue": (C.c_int, [_P, _I64]),
    "mobrob_ppo_set_hyper": (C.c_int, [_P, C.c_int32, C.c_double]),
    "mobrob_ctrl_turtlebot3": (C.c_int, [_P, C.c_int32, C.c_int32, _F, _F, _F, _F, _F]),
    "mobrob_ctrl_drone_pid": (C.c_int, [_P, C.c_int32, C.c_int32, C.POINTER(DroneParams), _F, _F, _F, _F, _F, _F]),
    "mobrob_ppo_last_train_info": (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "mobrob_ppo_comm_unique_id": (C.c_int, [_U8]),
    "mobrob_ppo_comm_prepare": (C.c_int, [_P]),
    "mobrob_ppo_comm_init": (C.c_int, [_P, _U8]),
    "mobrob_ppo_comm_init_rank": (C.c_int, [_P, _U8, C.c_int32, C.c_int32]),
    "mobrob_ppo_comm_info": (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "mobrob_ppo_allreduce_counters": (C.c_int, [_P, _I64, _I64, C.c_int32]),
    "mobrob_ppo_oneshot_export": (C.c_int, [_P, _U8]),
    "mobrob_ppo_oneshot_open": (C.c_int, [_P, _U8, C.c_int32, C.c_int32]),
    "mobrob_ppo_oneshot_close": (C.c_int, [_P]),
    "mobrob_ppo_exchange_selfcheck": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_int32)]),
    "mobrob_ppo_comm_destroy": (C.c_int, [_P]),
    "mobrob_ppo_train_dp": (C.c_int, [_P, _I64, _P, _P]),
    "mobrob_ppo_epoch_begin": (C.c_int, [_P, _I64]),
    "mobrob_ppo_num_minibatches": (C.c_int, [_P]),
    "mobrob_ppo_minibatch_grad": (C.c_int, [_P, C.c_int32]),
    "mobrob_ppo_minibatch_apply": (C.c_int, [_P]),
    "mobrob_ppo_minibatch_apply_checked": (C.c_int, [_P, C.POINTER(C.c_int32)]),
    "mobrob_ppo_fetch_step_stats": (C.c_int, [_P, _F, C.c_int32]),
    "mobrob_ppo_predict": (C.c_int, [_P, _F, C.c_int32, C.c_int32, _F, _F, _F]),
    "mobrob_ppo_sde_reset_noise": (C.c_int, [_P]),
    "mobrob_ppo_sde_set_noise": (C.c_int, [_P, _F]),
    "mobrob_ppo_buffer_info": (C.c_int, [_P, C.c_int32, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    "mobrob_ppo_read_buffer": (C.c_int, [_P, C.c_int32, _P, C.c_size_t]),
    "mobrob_ppo_write_buffer": (C.c_int, [_P, C.c_int32, _P, C.c_size_t]),
    "mobrob_ppo_mark_rollout_ready": (C.c_int, [_P]),
    "mobrob_ppo_compute_gae": (C.c_int, [_P]),
    "mobrob_ppo_explained_variance": (C.c_int, [_P, C.POINTER(C.c_double)]),
    "mobrob_ppo_x3_mode": (C.c_int, [_P]),
    "mobrob_ppo_update_mode": (C.c_int, [_P]),
    "mobrob_ppo_feistel_permutation": (C.c_int, [_P, C.c_int64, C.c_uint64, _I64]),
    "mobrob_ppo_profile_enable": (C.c_int, [_P, C.c_int32]),
    "mobrob_ppo_profile_read": (C.c_int, [_P, C.POINTER(C.c_double), _I64]),
}

ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p)  # mobrob_allreduce_fn

_lib = None


class EngineError(RuntimeError):
    pass


def load():
    """dlopen the in-tree HIP library; raises ImportError with the build hint if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950). mobrob_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype, fn.argtypes = res, args
    if lib.mobrob_ppo_abi_version() != ABI_VERSION:
        raise ImportError(f"ABI mismatch: library {lib.mobrob_ppo_abi_version()} != binding {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int):
    """C return code -> the exception class the reference would raise at this boundary."""
    if rc >= 0:
        return rc
    msg = load().mobrob_ppo_last_error().decode()
    if rc == ERR_INVALID:
        raise ValueError(msg)
    raise EngineError(f"[{rc}] {msg}")
