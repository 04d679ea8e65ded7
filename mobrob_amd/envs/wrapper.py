"""Goal-conditioned robot environment surface (the `EnvWrapper` contract of the reference).

Reference: /root/reference/src/mobrob/envs/wrapper.py -- `EnvWrapper` :15-228 (goal sampling,
distance-delta reward with +5 on reach :137-154, terminate-on-goal :156-171, lazy reset :173-201,
reach radius 0.3 :203-207), five concrete robots :293-546 and the `get_env` factory :549-571.

Scope (SURVEY.md §2 rows 3-5, §8b "what sits below"): the *surface* -- observation/action shapes and
dtypes, the gymnasium 5-tuple step/reset contract, goal API, reward and termination rules -- is what
feeds the PPO collector, so it is reproduced here.  The physics behind it (MuJoCo 2.1 via mujoco-py,
Bullet via pybullet) is third-party native code that exists on neither the build nor the GPU box and
is OUT OF SCOPE; `build_env` therefore returns a light kinematic stand-in (`KinematicSim`) with the
right dimensions.  A real simulator can be plugged in by subclassing `EnvWrapper` exactly as the
reference's README describes (nine abstract methods).
"""
from __future__ import annotations

from abc import ABC, abstractmethod

import numpy as np

from .spaces import Box

# obs/act dims per robot: reference wrapper.py:293-299 (point 14/2), :309-318 (car 26/2),
# :330-346 (doggo 58/12), :417-489 (drone 12/18), :509-546 (turtlebot3 43/2)
ROBOT_DIMS = {"point": (14, 2, 2), "car": (26, 2, 2), "doggo": (58, 12, 2), "drone": (12, 18, 3),
              "turtlebot3": (43, 2, 2)}  # (obs_dim, act_dim, position_dim)
# mean episode lengths recorded in the reference checkpoints -> synthetic p_term
MEAN_EPISODE_LEN = {"point": 119, "car": 91, "doggo": 107, "drone": 568, "turtlebot3": 131}


class KinematicSim:
    """Stand-in for the Engine/BulletEnv object behind `EnvWrapper.env`: a velocity-controlled point
    whose command is a fixed linear read-out of the action; observations are [pos-dependent features,
    velocity, sensor noise] padded to the robot's obs_dim.  NOT a physics model."""

    def __init__(self, obs_dim, act_dim, pos_dim, dt=0.05, extent=3.0):
        self.obs_dim, self.act_dim, self.pos_dim, self.dt, self.extent = obs_dim, act_dim, pos_dim, dt, extent
        self.observation_space = Box(-np.inf, np.inf, (obs_dim,), np.float32)
        self.action_space = Box(-1.0, 1.0, (act_dim,), np.float32)
        self.placements_extents = (-extent, -extent, extent, extent)
        self._rng = np.random.default_rng(0)
        mix = np.random.default_rng(12345 + obs_dim).standard_normal((pos_dim, act_dim))
        self._mix = (mix / np.linalg.norm(mix, axis=1, keepdims=True)).astype(np.float64)
        self.pos = np.zeros(pos_dim)
        self.vel = np.zeros(pos_dim)
        self.goal = np.zeros(pos_dim)

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)

    def reset(self):
        self.vel[:] = 0.0
        return self.obs(), {}

    def step(self, action):
        a = np.clip(np.asarray(action, np.float64), -1.0, 1.0)
        self.vel = 0.8 * self.vel + 0.2 * (self._mix @ a)
        self.pos = np.clip(self.pos + self.dt * self.vel, -self.extent, self.extent)
        return self.obs(), 0.0, False, False, {}

    def obs(self):
        o = np.zeros(self.obs_dim, np.float32)
        rel = (self.goal - self.pos)[: self.pos_dim]
        feat = np.concatenate([rel / (np.linalg.norm(rel) + 1e-6), self.vel, self.pos])
        k = min(len(feat), self.obs_dim)
        o[:k] = feat[:k]
        if self.obs_dim > k:
            o[k:] = 0.1 * self._rng.standard_normal(self.obs_dim - k)
        return o

    def render(self, mode="human"):
        return None

    def close(self):
        pass


class EnvWrapper(ABC):
    """gymnasium-style goal environment; see module docstring for the reference lines mirrored."""

    def __init__(self, enable_gui: bool = False, terminate_on_goal: bool = False):
        self.enable_gui = enable_gui
        self.terminate_on_goal = terminate_on_goal
        self._goal = None
        self._prev_pos = None
        self.env = self.build_env()
        self.observation_space = self.get_observation_space()
        self.action_space = self.get_action_space()
        self.init_space = self.get_init_space()
        self.goal_space = self.get_goal_space()
        self._first_reset = True
        self.render_mode = "human"

    # -- the nine methods a new robot implements (reference README.md:82-92, wrapper.py:39-93) --
    @abstractmethod
    def _set_goal(self, goal): ...
    @abstractmethod
    def build_env(self): ...
    @abstractmethod
    def get_pos(self): ...
    @abstractmethod
    def set_pos(self, pos): ...
    @abstractmethod
    def get_obs(self) -> np.ndarray: ...
    @abstractmethod
    def get_observation_space(self) -> Box: ...
    @abstractmethod
    def get_action_space(self) -> Box: ...
    @abstractmethod
    def get_init_space(self) -> Box: ...
    @abstractmethod
    def get_goal_space(self) -> Box: ...

    def seed(self, seed=None):
        self.env.seed(seed)
        self.init_space.seed(seed)
        self.goal_space.seed(seed + 1 if seed is not None else None)  # avoid init on goal
        self.action_space.seed(seed)
        self.observation_space.seed(seed)

    def toggle_render_mode(self):
        self.render_mode = "human" if self.render_mode == "rgb_array" else "rgb_array"

    def set_goal(self, goal):
        self._set_goal(goal)
        self._goal = np.array(goal)

    def reset_random_goal(self):
        self.set_goal(self.goal_space.sample())

    def get_goal(self) -> np.ndarray:
        return np.array([]) if self._goal is None else self._goal

    def reward_fn(self) -> float:
        """progress towards the goal since the previous step, +5 when inside the reach radius"""
        cur = self.get_pos()
        if self._goal is None or self._prev_pos is None:
            reward = 0.0
        else:
            reward = float(np.linalg.norm(self._goal - self._prev_pos) - np.linalg.norm(self._goal - cur))
        self._prev_pos = cur
        if self.reached():
            reward += 5.0
        return reward

    def step(self, action):
        obs, _, _, truncated, info = self.env.step(action)  # inner reward/termination are discarded
        reward = self.reward_fn()
        terminated = self.terminate_on_goal and self.reached()
        return obs, reward, terminated, truncated, info

    def reset(self, init_pos=None, *args, **kwargs):
        if "seed" in kwargs:
            seed = kwargs.pop("seed")
            if seed is not None:
                self.seed(seed)
        if self._first_reset or not self.reached():
            # lazy reset: a robot that has just reached its goal keeps its pose and only gets a new goal
            self.env.reset()
            self.set_pos(self.init_space.sample())
        if init_pos is not None:
            self.set_pos(init_pos)
        self.reset_random_goal()
        self._prev_pos = self.get_pos()
        self._first_reset = False
        return self.get_obs(), {}

    def reached(self, reach_radius: float = 0.3) -> bool:
        return bool(np.linalg.norm(self.get_pos() - self.get_goal()) < reach_radius)

    def reset_init_space(self, init_space: Box):
        self.init_space = init_space

    def reset_goal_space(self, goal_space: Box):
        self.goal_space = goal_space

    def render(self):
        return self.env.render(mode=self.render_mode)

    def close(self):
        self.env.close()


class KinematicGoalEnv(EnvWrapper):
    """Concrete env for one of the five robot names, backed by `KinematicSim`."""
    robot = "point"

    def build_env(self):
        d, a, p = ROBOT_DIMS[self.robot]
        return KinematicSim(d, a, p)

    def _set_goal(self, goal):
        g = np.zeros(self.env.pos_dim)
        g[: len(goal)] = np.asarray(goal, np.float64)[: self.env.pos_dim]
        self.env.goal = g

    def get_pos(self):
        return np.array(self.env.pos)

    def set_pos(self, pos):
        p = np.zeros(self.env.pos_dim)
        p[: len(pos)] = np.asarray(pos, np.float64)[: self.env.pos_dim]
        self.env.pos = p

    def get_obs(self):
        return self.env.obs()

    def get_observation_space(self):
        return self.env.observation_space

    def get_action_space(self):
        return self.env.action_space

    def get_init_space(self):
        e = self.env.extent
        return Box(np.full(self.env.pos_dim, -e / 2, np.float32), np.full(self.env.pos_dim, e / 2, np.float32))

    def get_goal_space(self):
        e = self.env.extent
        return Box(np.full(self.env.pos_dim, -e, np.float32), np.full(self.env.pos_dim, e, np.float32))


class PointEnv(KinematicGoalEnv):
    robot = "point"


class CarEnv(KinematicGoalEnv):
    robot = "car"


class DoggoEnv(KinematicGoalEnv):
    robot = "doggo"


class DroneEnv(KinematicGoalEnv):
    robot = "drone"

    def reward_fn(self) -> float:  # reference wrapper.py:491-496: extra +10 on reach
        r = super().reward_fn()
        return r + 10.0 if self.reached() else r


class Turtlebot3Env(KinematicGoalEnv):
    robot = "turtlebot3"


class TimeLimit:
    """gymnasium.wrappers.TimeLimit: truncated=True once `max_episode_steps` steps have elapsed."""

    def __init__(self, env, max_episode_steps):
        self.env = env
        self._max, self._elapsed = int(max_episode_steps), 0

    def __getattr__(self, name):
        return getattr(self.env, name)

    def step(self, action):
        obs, r, term, trunc, info = self.env.step(action)
        self._elapsed += 1
        if self._elapsed >= self._max:
            trunc = True
        return obs, r, term, trunc, info

    def reset(self, *a, **k):
        self._elapsed = 0
        return self.env.reset(*a, **k)


_ENVS = {"point": PointEnv, "car": CarEnv, "doggo": DoggoEnv, "drone": DroneEnv, "turtlebot3": Turtlebot3Env}


def get_env(env_name: str, enable_gui: bool = False, terminate_on_goal: bool = False, time_limit: int | None = None):
    """reference wrapper.py:549-571 -- same signature, same ValueError for an unknown name."""
    if env_name not in _ENVS:
        raise ValueError(f"Env {env_name} not found")
    env = _ENVS[env_name](enable_gui=enable_gui, terminate_on_goal=terminate_on_goal)
    if time_limit is not None:
        env = TimeLimit(env, max_episode_steps=time_limit)
    return env
