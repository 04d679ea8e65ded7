"""Vectorised environment layer feeding the rollout collector (the VecEnv contract of SB3 that the
reference obtains from `make_vec_env(get_env, n_envs, env_kwargs, vec_env_cls, seed)`,
/root/reference/src/mobrob/rl_control/ppo.py:37-48).

Contract (SURVEY.md §8b "what sits below"):
    reset() -> obs[N, D] float32
    step(actions[N, A]) -> (obs[N, D] float32, rewards[N] float32, dones[N] bool, infos[N])
        infos[i]["terminal_observation"], infos[i]["TimeLimit.truncated"], infos[i]["episode"] = {r, l, t}
        environments auto-reset when done.
Three sources:
  * HostVecEnv           in-process loop over `EnvWrapper` instances (replaces Dummy/SubprocVecEnv: the pipes
                         between processes are gone; thousands of light envs step in one address space)
  * SyntheticVecEnv      host NumPy statistical env source with the VecEnv contract (no simulator needed)
  * DeviceSyntheticVecEnv  marker: the engine's device-resident synthetic source (no host round trip at all)
  * DeviceGoalVecEnv     marker: the goal-reaching task of envs/wrapper.py stepped on the GPU (reward, termination,
                         lazy reset, time limit of the reference's EnvWrapper evaluated inside the rollout loop)
"""
from __future__ import annotations

import functools
import time

import numpy as np

from .wrapper import MEAN_EPISODE_LEN, ROBOT_DIMS, KinematicSim, observation_space_of


class VecEnvBase:
    num_envs: int
    obs_dim: int
    act_dim: int
    action_low = -1.0
    action_high = 1.0
    observation_space = None  # a Box when the robot is known (bounds go into the checkpoint)

    def close(self):
        pass


class HostVecEnv(VecEnvBase):
    def __init__(self, env_fns, seed=None):
        self.envs = [fn() for fn in env_fns]
        self.num_envs = len(self.envs)
        e0 = self.envs[0]
        self.obs_dim = int(e0.observation_space.shape[0])
        self.act_dim = int(e0.action_space.shape[0])
        self.observation_space, self.action_space = e0.observation_space, e0.action_space
        self._seeds = [None] * self.num_envs
        if seed is not None:
            self.seed(seed)
        self._ep_ret = np.zeros(self.num_envs, np.float64)
        self._ep_len = np.zeros(self.num_envs, np.int64)
        self._t0 = time.time()

    def seed(self, seed=None):
        # make_vec_env: env i gets seed + i, applied on its next reset (Appendix A.4)
        self._seeds = [None if seed is None else seed + i for i in range(self.num_envs)]
        for i, e in enumerate(self.envs):
            e.action_space.seed(None if seed is None else seed + i)

    def _reset_one(self, i):
        kw = {}
        if self._seeds[i] is not None:
            kw["seed"] = self._seeds[i]
            self._seeds[i] = None
        obs, _ = self.envs[i].reset(**kw)
        return np.asarray(obs, np.float32)

    def reset(self):
        self._ep_ret[:] = 0
        self._ep_len[:] = 0
        return np.stack([self._reset_one(i) for i in range(self.num_envs)])

    def step(self, actions):
        obs = np.empty((self.num_envs, self.obs_dim), np.float32)
        rews = np.empty(self.num_envs, np.float32)
        dones = np.zeros(self.num_envs, bool)
        infos = []
        for i, e in enumerate(self.envs):
            o, r, term, trunc, info = e.step(actions[i])
            info = dict(info)
            self._ep_ret[i] += r
            self._ep_len[i] += 1
            done = bool(term or trunc)
            info["TimeLimit.truncated"] = bool(trunc and not term)
            if done:
                info["terminal_observation"] = np.asarray(o, np.float32)
                info["episode"] = {"r": float(self._ep_ret[i]), "l": int(self._ep_len[i]),
                                   "t": round(time.time() - self._t0, 6)}
                self._ep_ret[i], self._ep_len[i] = 0.0, 0
                o = self._reset_one(i)
            obs[i], rews[i], dones[i] = o, r, done
            infos.append(info)
        return obs, rews, dones, infos

    def close(self):
        for e in self.envs:
            e.close()


class SyntheticVecEnv(VecEnvBase):
    """obs ~ N(0,1); reward ~ N(0.03, 0.1^2) + 5*terminated; terminated ~ Bernoulli(p_term);
    truncation after `time_limit` steps with a terminal observation."""

    def __init__(self, n_envs, obs_dim, act_dim, p_term=1 / 107.0, time_limit=1000, seed=0):
        self.num_envs, self.obs_dim, self.act_dim = int(n_envs), int(obs_dim), int(act_dim)
        self.p_term, self.time_limit = float(p_term), int(time_limit)
        self.rng = np.random.default_rng(seed)
        self._ep_len = np.zeros(self.num_envs, np.int64)
        self._ep_ret = np.zeros(self.num_envs, np.float64)
        self._t0 = time.time()

    @classmethod
    def for_robot(cls, env_name, n_envs, time_limit=1000, seed=0):
        if env_name not in ROBOT_DIMS:
            raise ValueError(f"Env {env_name} not found")
        d, a, _ = ROBOT_DIMS[env_name]
        env = cls(n_envs, d, a, 1.0 / MEAN_EPISODE_LEN[env_name], time_limit, seed)
        env.observation_space = observation_space_of(env_name)  # what PPO.save records for this robot
        return env

    def seed(self, seed=None):
        self.rng = np.random.default_rng(seed)

    def reset(self):
        self._ep_len[:] = 0
        self._ep_ret[:] = 0
        return self.rng.standard_normal((self.num_envs, self.obs_dim), dtype=np.float32)

    def step(self, actions):
        n = self.num_envs
        obs = self.rng.standard_normal((n, self.obs_dim), dtype=np.float32)
        term = self.rng.random(n) < self.p_term
        rew = (0.03 + 0.1 * self.rng.standard_normal(n, dtype=np.float32) + 5.0 * term).astype(np.float32)
        self._ep_len += 1
        self._ep_ret += rew
        trunc = (self._ep_len >= self.time_limit) & ~term
        dones = term | trunc
        infos = [{} for _ in range(n)]
        if dones.any():
            idx = np.nonzero(dones)[0]
            now = round(time.time() - self._t0, 6)
            for i in idx:
                infos[i] = {"terminal_observation": obs[i].copy(), "TimeLimit.truncated": bool(trunc[i]),
                            "episode": {"r": float(self._ep_ret[i]), "l": int(self._ep_len[i]), "t": now}}
            obs[idx] = self.rng.standard_normal((len(idx), self.obs_dim), dtype=np.float32)
            self._ep_len[idx] = 0
            self._ep_ret[idx] = 0
        return obs, rew, dones, infos


class DeviceSyntheticVecEnv(VecEnvBase):
    """Marker for the engine's device-resident generator (same distribution as SyntheticVecEnv, Philox
    streams); `PPO.learn` then runs whole rollouts on the GPU via mobrob_ppo_collect_synthetic."""

    def __init__(self, n_envs, obs_dim, act_dim, p_term=1 / 107.0, time_limit=1000, seed=0):
        self.num_envs, self.obs_dim, self.act_dim = int(n_envs), int(obs_dim), int(act_dim)
        self.p_term, self.time_limit, self._seed = float(p_term), int(time_limit), seed

    @classmethod
    def for_robot(cls, env_name, n_envs, time_limit=1000, seed=0):
        if env_name not in ROBOT_DIMS:
            raise ValueError(f"Env {env_name} not found")
        d, a, _ = ROBOT_DIMS[env_name]
        env = cls(n_envs, d, a, 1.0 / MEAN_EPISODE_LEN[env_name], time_limit, seed)
        env.observation_space = observation_space_of(env_name)  # what PPO.save records for this robot
        return env

    def seed(self, seed=None):
        self._seed = seed

    def reset(self):
        raise RuntimeError("DeviceSyntheticVecEnv lives on the GPU; it is stepped by the engine, not from the host")

    step = reset


class DeviceGoalVecEnv(VecEnvBase):
    """The goal-reaching task of `KinematicGoalEnv` (same kinematics, reward, termination, lazy reset and time
    limit as the host classes in envs/wrapper.py) stepped entirely on the GPU by
    mobrob_ppo_collect_goal_env -- the device counterpart of make_vec_env(get_env, ...) for learner-bound runs."""

    def __init__(self, n_envs, obs_dim, act_dim, pos_dim, time_limit=1000, terminate_on_goal=True, extra_bonus=0.0,
                 seed=0):
        self.num_envs, self.obs_dim, self.act_dim, self.pos_dim = int(n_envs), int(obs_dim), int(act_dim), int(pos_dim)
        self.time_limit, self.terminate_on_goal, self.extra_bonus, self._seed = int(time_limit), bool(terminate_on_goal), float(extra_bonus), seed
        sim = KinematicSim(obs_dim, act_dim, pos_dim)  # same action read-out and constants as the host stand-in
        self.mix, self.dt, self.extent = sim._mix.astype(np.float32), sim.dt, sim.extent

    @classmethod
    def for_robot(cls, env_name, n_envs, time_limit=1000, seed=0, terminate_on_goal=True):
        if env_name not in ROBOT_DIMS:
            raise ValueError(f"Env {env_name} not found")
        d, a, p = ROBOT_DIMS[env_name]
        env = cls(n_envs, d, a, p, time_limit, terminate_on_goal, 10.0 if env_name == "drone" else 0.0, seed)
        env.observation_space = observation_space_of(env_name)
        return env

    def collect(self, engine):
        engine.collect_goal_env(self.pos_dim, self.mix, self.time_limit, self.terminate_on_goal, dt=self.dt,
                                extent=self.extent, extra_bonus=self.extra_bonus)

    def seed(self, seed=None):
        self._seed = seed

    def reset(self):
        raise RuntimeError("DeviceGoalVecEnv lives on the GPU; it is stepped by the engine, not from the host")

    step = reset


def make_vec_env(env_fn, n_envs, env_kwargs=None, vec_env_cls=None, seed=None):
    """Signature of SB3's helper as the reference calls it (ppo.py:37-48)."""
    cls = vec_env_cls or HostVecEnv
    thunk = functools.partial(env_fn, **dict(env_kwargs or {}))  # picklable: worker processes rebuild it
    return cls([thunk] * n_envs, seed=seed)
