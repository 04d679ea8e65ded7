"""The goal task's rules as array functions over a batch of robots.

What the reference evaluates one robot at a time inside `EnvWrapper` (/root/reference/src/mobrob/envs/wrapper.py:
reward :137-154, termination :156-171, lazy reset :173-201, reach test :203-207) is stated here once, over `[n]` /
`[n, p]` arrays, so that every consumer applies literally the same rules: the single-robot `EnvWrapper` (n = 1),
the worker processes of `ShmVecEnv`, and the tests that compare the C (`csrc/host_env.c`) and device
(`csrc/kernels_env.h`) implementations against them.

The reward is stated on DISTANCES: a robot carries the distance to its goal measured at the end of its previous
step (`NaN` = nothing measured yet), and the step reward is the decrease of that distance -- the same number as
`|goal - prev_pos| - |goal - pos|`, because the goal only changes inside `reset`, which re-measures.
"""
from __future__ import annotations

import numpy as np

REACH_RADIUS = 0.3   # wrapper.py:203
GOAL_BONUS = 5.0     # wrapper.py:151-152


def goal_distance(goal, pos):
    """Euclidean distance robot -> goal, row-wise: [n, p], [n, p] -> [n] (float64)."""
    delta = np.asarray(goal, np.float64) - np.asarray(pos, np.float64)
    return np.sqrt(np.einsum("...i,...i->...", delta, delta))


def inside_goal(dist, radius=REACH_RADIUS):
    """Strictly inside the reach radius."""
    return np.asarray(dist) < radius


def progress_reward(prev_dist, dist, radius=REACH_RADIUS, bonus=GOAL_BONUS):
    """Decrease of the goal distance since the previous step (zero where no previous distance exists) plus the
    reach bonus -> (reward [n] float64, reached [n] bool)."""
    prev_dist, dist = np.asarray(prev_dist, np.float64), np.asarray(dist, np.float64)
    gain = np.where(np.isnan(prev_dist), 0.0, prev_dist - dist)
    hit = inside_goal(dist, radius)
    return gain + bonus * hit, hit


def episode_over(reached, terminate_on_goal):
    """`terminated` of the gymnasium 5-tuple: only goal arrival ends an episode, and only if asked to."""
    return np.logical_and(bool(terminate_on_goal), reached)


def must_respawn(ever_reset, reached):
    """Lazy reset: a robot is put back to a sampled start pose at its first reset and whenever the episode ended
    WITHOUT reaching the goal (time limit: it may be stuck); one that has just arrived keeps its pose and only
    receives a new goal."""
    return np.logical_or(np.logical_not(ever_reset), np.logical_not(reached))
