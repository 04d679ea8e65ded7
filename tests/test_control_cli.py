"""CPU: examples/control.py follows the reference script's evaluation protocol (/root/reference/examples/control.py:35-63):
1000 steps per epoch whatever happens, reset-and-continue when the goal is reached, the reward accumulated over the whole
epoch, the three report lines, default `--env-name point`.  The policy is the reference's point checkpoint (weights from the
committed fixture tests/golden/point.npz) evaluated by the oracle -- `load_policy` itself needs the GPU engine and is
exercised in tests/test_goal_env_gpu.py."""
import importlib.util
import os
import subprocess
import sys

import numpy as np

from oracle import ppo_oracle as O
from tests.util import golden_params, load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = os.path.join(ROOT, "examples", "control.py")


def _load_script():
    spec = importlib.util.spec_from_file_location("control_cli", SCRIPT)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class _OraclePolicy:
    """`.predict(obs, deterministic=True) -> (action, state)` with the checkpoint's weights (SB3 clips to the Box)."""

    def __init__(self, params):
        self.p, self.calls = params, 0

    def predict(self, obs, deterministic=False):
        assert deterministic is True and np.asarray(obs).shape == (14,)
        self.calls += 1
        mean, _ = O.policy_outputs(self.p, np.asarray(obs, np.float32)[None])
        return np.clip(mean[0], -1.0, 1.0), None


class _GoalSeeker:
    """Scripted policy for the kinematic stand-in: observation[:2] is the unit vector to the goal, the simulator's command
    is a fixed linear read-out of the action -> drive straight at the goal (guarantees the reset-and-continue branch runs)."""

    def __init__(self, env):
        self.pinv, self.calls = np.linalg.pinv(env.env._mix), 0

    def predict(self, obs, deterministic=False):
        assert deterministic is True
        self.calls += 1
        a = self.pinv @ np.asarray(obs[:2], np.float64)
        return (a / max(1.0, np.max(np.abs(a)))).astype(np.float32), None


class _CountingEnv:
    def __init__(self, env, seed=7):
        self.env, self.resets, self.steps, self.terminations, self.log, self._seed = env, 0, 0, 0, [], seed

    def reset(self, *a, **k):
        self.resets += 1
        if self.resets == 1:          # the script resets without a seed, like the reference: fix the draw sequence here
            k.setdefault("seed", self._seed)
        return self.env.reset(*a, **k)

    def step(self, action):
        out = self.env.step(action)
        self.steps += 1
        self.terminations += int(bool(out[2]))
        self.log.append(float(out[1]))
        return out


def test_evaluation_protocol_is_the_reference_scripts(capsys):
    from mobrob_amd import get_env
    cli = _load_script()
    assert cli.STEPS_PER_EPOCH == 1000
    # (a) the reference's point checkpoint through the loop: 1000 predict calls per epoch whatever the episodes do
    policy = _OraclePolicy(golden_params(load_golden("point")))
    env = _CountingEnv(get_env("point", enable_gui=False, terminate_on_goal=True))
    assert len(cli.simulate("point", epochs=2, env=env, policy=policy)) == 2
    assert policy.calls == 2 * 1000 and env.steps == 2 * 1000 and env.resets == 2 + env.terminations
    capsys.readouterr()
    # (b) a policy that does reach goals: every goal restarts the episode and the epoch goes on accumulating
    raw = get_env("point", enable_gui=False, terminate_on_goal=True)
    policy, env = _GoalSeeker(raw), _CountingEnv(raw)
    epochs = 3
    rewards = cli.simulate("point", epochs=epochs, env=env, policy=policy)
    assert policy.calls == epochs * 1000 and env.steps == epochs * 1000      # never cut short by a termination
    assert env.terminations >= 3 * epochs                                      # the robot reaches goal after goal ...
    assert env.resets == epochs + env.terminations                             # ... and every one restarts the episode
    assert min(rewards) > 5.0 * 3                                              # several +5 arrival bonuses in ONE epoch figure
    per_epoch = np.array(env.log).reshape(epochs, 1000).sum(axis=1)
    assert np.allclose(rewards, per_epoch)                                     # cumulative over the epoch, goal bonuses included
    out = capsys.readouterr().out.strip().splitlines()
    assert out[0] == f"average reward: {np.mean(rewards)}"
    assert out[1] == f"reward stds: {np.std(rewards)}"
    assert out[2] == f"rewards: {rewards}"


def test_command_line_matches_the_reference():
    r = subprocess.run([sys.executable, SCRIPT, "--help"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0
    for flag in ("--env-name", "--policy-name", "--epochs", "--no-gui", "--video-path"):
        assert flag in r.stdout
    src = open(SCRIPT).read()
    assert 'add_argument("--env-name", type=str, default="point")' in src
    assert 'add_argument("--policy-name", type=str, default="ppo")' in src
    assert 'add_argument("--epochs", type=int, default=5)' in src
