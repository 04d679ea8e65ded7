"""DATA_DIR / load_policy -- mirrors /root/reference/src/mobrob/utils.py:11-16 (path + checkpoint convention).

Video recording and stdout suppression helpers of the reference (utils.py:19-57) are GUI/IO cosmetics and
out of scope."""
import os
from os.path import abspath, dirname

PROJ_DIR = dirname(abspath(__file__))
DATA_DIR = os.environ.get("MOBROB_DATA_DIR", os.path.join(dirname(PROJ_DIR), "data"))


def load_policy(env_name: str, policy_name: str):
    """`PPO.load(f"{DATA_DIR}/policies/{env_name}-{policy_name}.zip")` (reference utils.py:15-16)."""
    from .rl_control.ppo import PPO
    return PPO.load(f"{DATA_DIR}/policies/{env_name}-{policy_name}.zip")
