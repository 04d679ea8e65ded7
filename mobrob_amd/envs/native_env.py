"""Native vectorised host environment: ctypes binding of csrc/host_env.c (libmobrob_hostenv.so).

`NativeGoalVecEnv` is the host-side VecEnv for large env counts: one C call steps all environments on a thread pool
and writes observations directly into (pinned) staging arrays, so that the per-step Python work of the rollout
collector is O(1) instead of O(n_envs) (no per-env objects, no list of info dicts on the fast path).  It implements
both the SB3 VecEnv contract (`reset` / `step` -> obs, rewards, dones, infos) and the array protocol the collector
prefers (`step_arrays`)."""
from __future__ import annotations

import ctypes as C
import os
import time

import numpy as np

from .vec_env import VecEnvBase
from .wrapper import ROBOT_DIMS, KinematicSim, observation_space_of

_LIB = None
LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "libmobrob_hostenv.so")


def _load():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'`")
        lib = C.CDLL(LIB_PATH)
        F, U8, D = C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.POINTER(C.c_double)
        lib.mobrob_hostenv_create.restype = C.c_void_p
        lib.mobrob_hostenv_create.argtypes = [C.c_int32] * 6 + [C.c_double] * 6 + [D, C.c_uint64]
        lib.mobrob_hostenv_destroy.argtypes = [C.c_void_p]
        lib.mobrob_hostenv_reset.argtypes = [C.c_void_p, F]
        lib.mobrob_hostenv_step.restype = C.c_int32
        lib.mobrob_hostenv_step.argtypes = [C.c_void_p, F, F, F, U8, U8, F]
        lib.mobrob_hostenv_step_range.restype = C.c_int32
        lib.mobrob_hostenv_step_range.argtypes = [C.c_void_p, C.c_int32, C.c_int32, F, F, F, U8, U8, F]
        lib.mobrob_hostenv_episode_stats.argtypes = [C.c_void_p, D, C.c_int32]
        lib.mobrob_hostenv_episode_records.restype = C.c_int32
        lib.mobrob_hostenv_episode_records.argtypes = [C.c_void_p, D, C.c_int32]
        lib.mobrob_hostenv_get_state.argtypes = [C.c_void_p, C.c_int32, D]
        lib.mobrob_hostenv_set_threads.argtypes = [C.c_void_p, C.c_int32]
        lib.mobrob_hostenv_get_threads.restype = C.c_int32
        lib.mobrob_hostenv_get_threads.argtypes = [C.c_void_p]
        _LIB = lib
    return _LIB


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


class NativeGoalVecEnv(VecEnvBase):
    """The goal-reaching task of `KinematicGoalEnv` (envs/wrapper.py) for `n_envs` robots, stepped natively."""

    def __init__(self, n_envs, obs_dim, act_dim, pos_dim, time_limit=1000, terminate_on_goal=True, extra_bonus=0.0, seed=0):
        self.lib = _load()
        self.num_envs, self.obs_dim, self.act_dim, self.pos_dim = int(n_envs), int(obs_dim), int(act_dim), int(pos_dim)
        self.time_limit, self.terminate_on_goal = int(time_limit), bool(terminate_on_goal)
        sim = KinematicSim(obs_dim, act_dim, pos_dim)
        self.mix, self.dt, self.extent = np.ascontiguousarray(sim._mix, np.float64), sim.dt, sim.extent
        self._h = self.lib.mobrob_hostenv_create(self.num_envs, self.obs_dim, self.act_dim, self.pos_dim,
                                                 int(self.terminate_on_goal), self.time_limit, self.dt, self.extent, 0.3,
                                                 5.0, float(extra_bonus), 0.1,
                                                 self.mix.ctypes.data_as(C.POINTER(C.c_double)), int(seed or 0))
        if not self._h:
            raise ValueError("mobrob_hostenv_create rejected the arguments")
        n, d = self.num_envs, self.obs_dim
        self._obs = np.zeros((n, d), np.float32)
        self._rew = np.zeros(n, np.float32)
        self._done = np.zeros(n, np.uint8)
        self._trunc = np.zeros(n, np.uint8)
        self._term = np.zeros((n, d), np.float32)
        self._t0 = time.time()
        self._act_arr = self._act_ptr = self._out_ptrs = None  # pointer cache of step_range

    @classmethod
    def for_robot(cls, env_name, n_envs, time_limit=1000, seed=0, terminate_on_goal=True):
        if env_name not in ROBOT_DIMS:
            raise ValueError(f"Env {env_name} not found")
        d, a, p = ROBOT_DIMS[env_name]
        env = cls(n_envs, d, a, p, time_limit, terminate_on_goal, 10.0 if env_name == "drone" else 0.0, seed)
        env.observation_space = observation_space_of(env_name)
        return env

    def use_buffers(self, obs=None, rewards=None, dones=None, truncated=None, terminal_obs=None):
        """Write step results into caller-owned arrays (e.g. `engine.pinned(...)` staging: no extra copy before DMA)."""
        for name, arr in (("_obs", obs), ("_rew", rewards), ("_done", dones), ("_trunc", truncated), ("_term", terminal_obs)):
            if arr is not None:
                cur = getattr(self, name)
                if arr.shape != cur.shape or arr.dtype != cur.dtype or not arr.flags.c_contiguous:
                    raise ValueError(f"buffer for {name[1:]} must be C-contiguous {cur.dtype}{cur.shape}")
                setattr(self, name, arr)
        self._act_arr = None

    def seed(self, seed=None):
        pass  # seeded at construction (make_vec_env semantics: env i <- seed + i)

    def reset(self):
        self.lib.mobrob_hostenv_reset(self._h, _fp(self._obs))
        return self._obs

    def step_arrays(self, actions):
        """-> (obs, rewards, dones u8, truncated u8, terminal_obs, n_truncated); arrays are reused between calls."""
        a = np.ascontiguousarray(actions, np.float32)
        nt = self.lib.mobrob_hostenv_step(self._h, _fp(a), _fp(self._obs), _fp(self._rew), _u8(self._done), _u8(self._trunc),
                                          _fp(self._term))
        return self._obs, self._rew, self._done, self._trunc, self._term, int(nt)

    def step_range(self, i0, i1, actions):
        """Step the envs [i0, i1) only (actions is the full [n, act_dim] float32 array; result rows outside the range
        are untouched) -> number of truncated envs in the range.  Used by the pipelined collector."""
        if actions is not self._act_arr:
            if actions.dtype != np.float32 or not actions.flags.c_contiguous or actions.shape != (self.num_envs, self.act_dim):
                raise ValueError("step_range needs the full C-contiguous float32 [n_envs, act_dim] action array")
            self._act_arr, self._act_ptr = actions, _fp(actions)
            self._out_ptrs = (_fp(self._obs), _fp(self._rew), _u8(self._done), _u8(self._trunc), _fp(self._term))
        return self.lib.mobrob_hostenv_step_range(self._h, i0, i1, self._act_ptr, *self._out_ptrs)

    @property
    def step_range_fn(self):
        """Address of mobrob_hostenv_step_range (a mobrob_env_step_range_fn) for mobrob_ppo_collect_host; the env
        argument is `self.handle`.  The native collector writes into the arrays given to `use_buffers`."""
        return C.cast(self.lib.mobrob_hostenv_step_range, C.c_void_p).value

    @property
    def handle(self):
        return self._h

    def step(self, actions):
        """SB3 VecEnv contract (infos only carry what the collector reads; per-episode Monitor values are aggregated in
        `episode_stats`)."""
        obs, rew, done, trunc, term, _ = self.step_arrays(actions)
        infos = [{} for _ in range(self.num_envs)]
        for i in np.nonzero(done)[0]:
            infos[i] = {"TimeLimit.truncated": bool(trunc[i])}
            if trunc[i]:
                infos[i]["terminal_observation"] = term[i].copy()
        return obs.copy(), rew.copy(), done.astype(bool), infos

    def episode_stats(self, reset=True):
        out = np.zeros(4, np.float64)
        self.lib.mobrob_hostenv_episode_stats(self._h, out.ctypes.data_as(C.POINTER(C.c_double)), int(bool(reset)))
        n = int(out[0])
        return {"episodes": n, "goals": int(out[1]), "ep_rew_mean": out[2] / n if n else float("nan"),
                "ep_len_mean": out[3] / n if n else float("nan")}

    def pop_episodes(self, max_records=100):
        """Monitor records {r, l, t} of the episodes finished since the last call (oldest first, newest 100 at most)."""
        out = np.zeros((int(max_records), 2), np.float64)
        n = self.lib.mobrob_hostenv_episode_records(self._h, out.ctypes.data_as(C.POINTER(C.c_double)), int(max_records))
        now = round(time.time() - self._t0, 6)
        return [{"r": float(r), "l": int(l), "t": now} for r, l in out[:n]]

    @property
    def threads(self):
        return int(self.lib.mobrob_hostenv_get_threads(self._h))

    def set_threads(self, n):
        self.lib.mobrob_hostenv_set_threads(self._h, int(n))

    def state(self, i):
        out = np.zeros(9, np.float64)
        self.lib.mobrob_hostenv_get_state(self._h, int(i), out.ctypes.data_as(C.POINTER(C.c_double)))
        return out[:3], out[3:6], out[6:]

    def close(self):
        if getattr(self, "_h", None):
            self.lib.mobrob_hostenv_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
