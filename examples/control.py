#!/usr/bin/env python3
"""Roll out a trained policy (deterministic) and print mean/std episode reward.

CLI-compatible with the reference's examples/control.py (:66-82): --env-name, --policy-name, --epochs,
--no-gui, --video-path (GUI and video need a real simulator and are ignored by the kinematic stand-in)."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from mobrob_amd import get_env, load_policy  # noqa: E402


def simulate(env_name, policy_name="ppo", epochs=5, steps=1000, gui=False, video_path=None):
    env = get_env(env_name, enable_gui=gui, terminate_on_goal=True)
    policy = load_policy(env_name, policy_name)
    returns = []
    for ep in range(epochs):
        obs, _ = env.reset(seed=ep)
        total = 0.0
        for _ in range(steps):
            action, _ = policy.predict(obs, deterministic=True)
            obs, reward, terminated, truncated, _ = env.step(action)
            total += reward
            if terminated or truncated:
                break
        returns.append(total)
    print(f"mean reward: {np.mean(returns):.3f}, std reward: {np.std(returns):.3f}")
    return returns


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--env-name", type=str, default="drone")
    ap.add_argument("--policy-name", type=str, default="ppo")
    ap.add_argument("--epochs", type=int, default=5)
    ap.add_argument("--no-gui", action="store_true", default=False)
    ap.add_argument("--video-path", type=str, default=None)
    a = ap.parse_args()
    simulate(a.env_name, a.policy_name, a.epochs, gui=not a.no_gui, video_path=a.video_path)
