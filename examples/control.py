#!/usr/bin/env python3
"""Run a trained policy in its environment and report the reward it collects.

Same protocol and command line as the reference script (/root/reference/examples/control.py:11-82), so that numbers
from the two are comparable (last row):

* `--epochs` evaluation epochs (default 5) of exactly STEPS_PER_EPOCH = 1000 environment steps each (:36-39);
* the policy acts deterministically, `policy.predict(obs, deterministic=True)` (:40);
* reaching the goal (`terminated`) RESETS the environment and the epoch goes on -- the epoch's figure is the reward
  accumulated over all 1000 steps, usually several episodes, not the return of one episode (:41-46);
* three report lines: `average reward`, `reward stds`, `rewards` (:61-63);
* defaults `--env-name point --policy-name ppo` (:68-69).

GUI rendering, the 5 ms sleep that paces the Bullet GUI and video recording (:24-33, :48-52) need the real MuJoCo / Bullet
simulators; the kinematic stand-in robots have nothing to draw, so `--no-gui` / `--video-path` are accepted and ignored.
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

STEPS_PER_EPOCH = 1000


def evaluation_epoch(env, policy, steps=STEPS_PER_EPOCH):
    """Reward collected over `steps` consecutive environment steps, restarting the episode whenever the goal is reached."""
    collected = 0.0
    obs, _ = env.reset()
    for _ in range(steps):
        action, _state = policy.predict(obs, deterministic=True)
        obs, reward, terminated, _truncated, _info = env.step(action)
        if terminated:
            obs, _ = env.reset()
        collected += reward
    return collected


def simulate(env_name, policy_name="ppo", epochs=5, no_gui=True, video_path=None, env=None, policy=None):
    """env / policy: injected instances (tests); by default `get_env(env_name, terminate_on_goal=True)` and the
    checkpoint `data/policies/<env_name>-<policy_name>.zip`, exactly what the reference script builds (:19-20)."""
    if env is None or policy is None:
        from mobrob_amd import get_env, load_policy
        env = get_env(env_name, enable_gui=not no_gui, terminate_on_goal=True) if env is None else env
        policy = load_policy(env_name, policy_name) if policy is None else policy
    rewards = [evaluation_epoch(env, policy) for _ in range(epochs)]
    print(f"average reward: {np.mean(rewards)}")
    print(f"reward stds: {np.std(rewards)}")
    print(f"rewards: {rewards}")
    return rewards


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--env-name", type=str, default="point")
    ap.add_argument("--policy-name", type=str, default="ppo")
    ap.add_argument("--epochs", type=int, default=5)
    ap.add_argument("--no-gui", action="store_true", default=False)
    ap.add_argument("--video-path", type=str, default=None)
    a = ap.parse_args()
    simulate(env_name=a.env_name, policy_name=a.policy_name, epochs=a.epochs, no_gui=a.no_gui, video_path=a.video_path)
