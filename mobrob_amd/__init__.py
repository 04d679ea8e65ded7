"""mobrob_amd -- MI355X-native goal-conditioned PPO training path of ZikangXiong/mobrob.

Mirrors the reference package surface (`/root/reference/src/mobrob/__init__.py:1-4`):
`get_env`, `load_policy`.  Importing the package does not need a GPU; constructing a PPO does.
"""
from .envs.wrapper import get_env  # noqa: F401
from .utils import load_policy  # noqa: F401

__all__ = ["get_env", "load_policy"]
