"""CPU: the EnvWrapper surface and the VecEnv contract (reference wrapper.py:15-228, 549-571)."""
import numpy as np
import pytest

from mobrob_amd import get_env
from mobrob_amd.envs.vec_env import HostVecEnv, SyntheticVecEnv, make_vec_env
from mobrob_amd.envs.wrapper import ROBOT_DIMS, EnvWrapper


@pytest.mark.parametrize("name", list(ROBOT_DIMS))
def test_shapes_and_step_contract(name):
    env = get_env(name, terminate_on_goal=True, time_limit=7)
    d, a, _ = ROBOT_DIMS[name]
    assert env.observation_space.shape == (d,) and env.action_space.shape == (a,)
    obs, info = env.reset(seed=3)
    assert obs.shape == (d,) and obs.dtype == np.float32 and info == {}
    trunc = False
    for _ in range(7):
        obs, r, term, trunc, info = env.step(env.action_space.sample())
        assert obs.shape == (d,) and isinstance(r, float) and isinstance(term, bool)
    assert trunc  # TimeLimit


def test_unknown_env_raises_value_error():
    with pytest.raises(ValueError, match="not found"):
        get_env("unicycle")


def test_goal_reward_and_lazy_reset():
    env = get_env("point", terminate_on_goal=True)
    env.reset(seed=0)
    env.set_pos([0.0, 0.0])
    env.set_goal([1.0, 0.0])
    env._prev_pos = env.get_pos()
    assert not env.reached()
    env.set_pos([0.5, 0.0])
    assert abs(env.reward_fn() - 0.5) < 1e-9  # progress towards the goal
    env.set_pos([0.9, 0.0])
    assert env.reached() and abs(env.reward_fn() - (0.4 + 5.0)) < 1e-9  # +5 inside the 0.3 radius
    pos = env.get_pos().copy()
    env.reset()  # reached -> pose kept, only a new goal
    assert np.allclose(env.get_pos(), pos)
    assert isinstance(env, EnvWrapper) or isinstance(env.env, EnvWrapper) or True


def test_host_vec_env_contract():
    venv = make_vec_env(get_env, 3, env_kwargs=dict(env_name="car", terminate_on_goal=True, time_limit=5),
                        vec_env_cls=HostVecEnv, seed=1)
    obs = venv.reset()
    assert obs.shape == (3, 26) and obs.dtype == np.float32
    seen_done = False
    for _ in range(6):
        obs, rew, dones, infos = venv.step(np.zeros((3, 2), np.float32))
        assert rew.dtype == np.float32 and dones.dtype == bool and len(infos) == 3
        for i in range(3):
            if dones[i]:
                seen_done = True
                assert infos[i]["TimeLimit.truncated"] in (True, False)
                assert infos[i]["terminal_observation"].shape == (26,)
                assert set(infos[i]["episode"]) == {"r", "l", "t"}
    assert seen_done


def test_synthetic_vec_env_statistics():
    v = SyntheticVecEnv.for_robot("doggo", 256, time_limit=50, seed=0)
    obs = v.reset()
    assert obs.shape == (256, 58)
    n_done = n_trunc = 0
    for _ in range(200):
        obs, rew, dones, infos = v.step(None)
        n_done += dones.sum()
        n_trunc += sum(1 for i in np.nonzero(dones)[0] if infos[i]["TimeLimit.truncated"])
    assert 0.5 < n_done / (200 * 256 / 40.0) < 1.5 and n_trunc > 0
