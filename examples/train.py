#!/usr/bin/env python3
"""Train a goal-conditioned PPO policy on the MI355X engine.

CLI-compatible with the reference's examples/train.py (:52-61): --env-name, --finetune, --save-freq.
YAML -> PPOCtrl.from_config -> optional weight-only finetune -> periodic checkpoints -> learn -> save zip.
Extra (build-only) flag: --vec-env-type overrides the YAML's value (e.g. `device` for the GPU-resident
synthetic env source when no simulator is installed)."""
import argparse
import os
import sys

import yaml

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from mobrob_amd.rl_control.ppo import PPO, CheckpointCallback, PPOCtrl  # noqa: E402
from mobrob_amd.utils import DATA_DIR  # noqa: E402


def train_with_ppo(env_name, finetune=False, save_freq=1_000_000, vec_env_type=None, total_timesteps=None):
    with open(f"{DATA_DIR}/configs/{env_name}-ppo.yaml", "r") as f:
        config = yaml.load(f, Loader=yaml.FullLoader)
    if vec_env_type is not None:
        config["vec_env_type"] = vec_env_type
    ppo_ctrl = PPOCtrl.from_config(config=config)

    if finetune:  # weights only: optimizer state, counters and RNG start fresh (reference train.py:30-33)
        ppo_ctrl.ppo.policy.load_state_dict(PPO.load(f"{DATA_DIR}/policies/{env_name}-ppo.zip").policy.state_dict())

    temp_dir = f"{DATA_DIR}/policies/tmp/{env_name}-ppo"
    save_callback = CheckpointCallback(save_freq=save_freq // config["n_envs"], save_path=f"{temp_dir}/models",
                                       name_prefix="timestep", verbose=1)
    ppo_ctrl.learn(total_timesteps=total_timesteps or config["total_timesteps"], callback=save_callback,
                   progress_bar=True)
    os.makedirs(f"{DATA_DIR}/policies", exist_ok=True)
    ppo_ctrl.save_model(f"{DATA_DIR}/policies/{env_name}-ppo.zip")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--env-name", type=str, default="drone")
    ap.add_argument("--finetune", action="store_true", default=False)
    ap.add_argument("--save-freq", type=int, default=1_000_000)
    ap.add_argument("--vec-env-type", type=str, default=None, help="override the YAML (subproc|dummy|synthetic|native|device|device_goal)")
    ap.add_argument("--total-timesteps", type=int, default=None, help="override the YAML")
    a = ap.parse_args()
    train_with_ppo(a.env_name, a.finetune, a.save_freq, a.vec_env_type, a.total_timesteps)
