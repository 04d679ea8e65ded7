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

from . import goal_rules as rules
from .spaces import Box

# obs/act dims per robot: reference wrapper.py:293-299 (point 14/2), :309-318 (car 26/2),
# :330-346 (doggo 58/12), :417-489 (drone 12/18), :509-546 (turtlebot3 43/2)
ROBOT_DIMS = {"point": (14, 2, 2), "car": (26, 2, 2), "doggo": (58, 12, 2), "drone": (12, 18, 3),
              "turtlebot3": (43, 2, 2)}  # (obs_dim, act_dim, position_dim)
# Finite observation bounds two robots declare (reference wrapper.py:423-468 drone: pose / Euler angles / velocities /
# body rates; :515-529 turtlebot3: sin, cos, x, y scaled by sqrt 2, velocity limits 0.26 / 0.26 / 1.82, 36 lidar rays of
# length 1).  The numbers are also inside the reference checkpoints' `observation_space` blobs, which
# tests/test_checkpoint.py decodes and compares with these.  The MuJoCo robots are unbounded.
_PI = float(np.float32(np.pi))
OBS_BOUNDS = {
    "drone": (np.array([-10, -10, -50] + [-_PI] * 3 + [-15] * 3 + [-0.2 * np.pi] * 3, np.float32),
              np.array([10, 10, 5] + [_PI] * 3 + [15] * 3 + [0.2 * np.pi] * 3, np.float32)),
    "turtlebot3": (-np.array([1, 1, 2 ** 0.5, 2 ** 0.5, 0.26, 0.26, 1.82] + [1.0] * 36, np.float32),
                   np.array([1, 1, 2 ** 0.5, 2 ** 0.5, 0.26, 0.26, 1.82] + [1.0] * 36, np.float32)),
}


def observation_space_of(env_name: str) -> Box:
    """`env.observation_space` of a robot without building the robot (vector layers that live on the GPU or in C)."""
    d = ROBOT_DIMS[env_name][0]
    low, high = OBS_BOUNDS.get(env_name, (np.full(d, -np.inf, np.float32), np.full(d, np.inf, np.float32)))
    return Box(low, high, (d,), np.float32)


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
    """One goal-conditioned robot behind the gymnasium 5-tuple API -- the class a user subclasses to add a robot
    (nine abstract methods, reference README.md:82-92 / wrapper.py:39-93) and the unit `get_env` hands to the vector
    layers (`HostVecEnv` in-process, `ShmVecEnv` across worker processes).

    The task rules themselves live in `goal_rules` (array form); this class binds them to ONE simulator instance:
    it keeps the goal, the goal distance measured at the end of the previous step and whether the robot was ever
    reset, and asks the subclass for positions.  Public names, signatures and return conventions are the
    reference's (`step`, `reset(init_pos=None, seed=...)`, `reward_fn`, `reached`, `set_goal`, ...), because callers
    and subclasses use them (e.g. a subclass may extend `reward_fn`, which `step` therefore always goes through)."""

    reach_radius = rules.REACH_RADIUS
    goal_bonus = rules.GOAL_BONUS

    def __init__(self, enable_gui: bool = False, terminate_on_goal: bool = False):
        self.enable_gui = bool(enable_gui)
        self.terminate_on_goal = bool(terminate_on_goal)
        self.render_mode = "human"
        self._goal = None               # ndarray once set_goal() ran
        self._dist_before = np.nan      # goal distance at the end of the previous step (NaN: not measured yet)
        self._ever_reset = False
        self.env = self.build_env()
        self.observation_space, self.action_space = self.get_observation_space(), self.get_action_space()
        self.init_space, self.goal_space = self.get_init_space(), self.get_goal_space()

    # ---- what a robot implements ---------------------------------------------------------------------------
    @abstractmethod
    def build_env(self):
        """Create the simulator object kept in `self.env` (needs seed/reset/step/render/close)."""

    @abstractmethod
    def _set_goal(self, goal):
        """Move the goal marker inside the simulator."""

    @abstractmethod
    def get_pos(self):
        """Robot position in goal coordinates."""

    @abstractmethod
    def set_pos(self, pos):
        """Teleport the robot."""

    @abstractmethod
    def get_obs(self) -> np.ndarray:
        """Current observation vector."""

    @abstractmethod
    def get_observation_space(self) -> Box: ...

    @abstractmethod
    def get_action_space(self) -> Box: ...

    @abstractmethod
    def get_init_space(self) -> Box: ...

    @abstractmethod
    def get_goal_space(self) -> Box: ...

    # ---- goal bookkeeping ----------------------------------------------------------------------------------
    def _distance_now(self) -> float:
        return float(rules.goal_distance(self.get_goal(), self.get_pos()))

    def set_goal(self, goal):
        self._set_goal(goal)
        self._goal = np.array(goal)
        if not np.isnan(self._dist_before):  # progress is always measured against the goal in force
            self._dist_before = self._distance_now()

    def get_goal(self) -> np.ndarray:
        return self._goal if self._goal is not None else np.array([])

    def reset_random_goal(self):
        self.set_goal(self.goal_space.sample())

    def reached(self, reach_radius: float | None = None) -> bool:
        radius = self.reach_radius if reach_radius is None else reach_radius
        return bool(rules.inside_goal(self._distance_now(), radius))

    def reward_fn(self) -> float:
        if self._goal is None:
            return 0.0
        dist = self._distance_now()
        reward, _ = rules.progress_reward(self._dist_before, dist, self.reach_radius, self.goal_bonus)
        self._dist_before = dist
        return float(reward)

    # ---- gymnasium API -------------------------------------------------------------------------------------
    def seed(self, seed=None):
        """Simulator and the four spaces; the goal space is offset by one so that the first start pose and the first
        goal of an environment are not the same draw."""
        self.env.seed(seed)
        for space, offset in ((self.init_space, 0), (self.goal_space, 1), (self.action_space, 0),
                              (self.observation_space, 0)):
            space.seed(None if seed is None else seed + offset)

    def step(self, action):
        # the simulator's own reward / termination are not the task's: only its observation, truncation and info count
        obs, _, _, truncated, info = self.env.step(action)
        reward = self.reward_fn()
        terminated = bool(rules.episode_over(self.reached(), self.terminate_on_goal))
        return obs, reward, terminated, truncated, info

    def reset(self, init_pos=None, *args, **kwargs):
        seed = kwargs.pop("seed", None)
        if seed is not None:
            self.seed(seed)
        if rules.must_respawn(self._ever_reset, self._ever_reset and self.reached()):
            self.env.reset()
            self.set_pos(self.init_space.sample())
        if init_pos is not None:
            self.set_pos(init_pos)
        self._ever_reset = True
        self.reset_random_goal()
        self._dist_before = self._distance_now()
        return self.get_obs(), {}

    # ---- odds and ends of the surface ----------------------------------------------------------------------
    def reset_init_space(self, init_space: Box):
        self.init_space = init_space

    def reset_goal_space(self, goal_space: Box):
        self.goal_space = goal_space

    def toggle_render_mode(self):
        self.render_mode = {"human": "rgb_array", "rgb_array": "human"}[self.render_mode]

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
        return observation_space_of(self.robot)

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

    extra_goal_bonus = 10.0  # the drone moves fast: a larger arrival bonus (reference wrapper.py:491-496)

    def reward_fn(self) -> float:
        return super().reward_fn() + (self.extra_goal_bonus if self.reached() else 0.0)


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
