"""CPU: the native (C, OpenMP) vectorised host environment follows the EnvWrapper rules of envs/wrapper.py
(reference wrapper.py:137-207, 549-571) -- checked against the Python classes' arithmetic transition by transition."""
import numpy as np
import pytest

from mobrob_amd.envs.wrapper import ROBOT_DIMS


def _make(robot, n, tl, **kw):
    from mobrob_amd.envs.native_env import NativeGoalVecEnv
    return NativeGoalVecEnv.for_robot(robot, n, time_limit=tl, seed=3, **kw)


@pytest.mark.parametrize("robot", ["point", "drone", "turtlebot3"])
def test_native_env_follows_the_wrapper_rules(robot):
    D, A, P = ROBOT_DIMS[robot]
    N, TL, STEPS = 96, 30, 80
    env = _make(robot, N, TL)
    obs = env.reset().copy()
    assert obs.shape == (N, D) and obs.dtype == np.float32
    rng = np.random.default_rng(0)
    ep_len = np.zeros(N, int)
    n_done = n_goal = 0
    for t in range(STEPS):
        state0 = [env.state(i) for i in range(N)]
        act = rng.uniform(-1.5, 1.5, (N, A)).astype(np.float32)
        o, r, done, trunc, term, nt = env.step_arrays(act)
        o, r, done, trunc, term = o.copy(), r.copy(), done.copy(), trunc.copy(), term.copy()
        ep_len += 1
        for i in range(N):
            pos0, vel0, goal0 = (x[:P] for x in state0[i])
            cmd = env.mix @ np.clip(act[i].astype(np.float64), -1, 1)
            vel1 = 0.8 * vel0 + 0.2 * cmd
            pos1 = np.clip(pos0 + env.dt * vel1, -env.extent, env.extent)
            d0, d1 = np.linalg.norm(goal0 - pos0), np.linalg.norm(goal0 - pos1)
            reached = d1 < 0.3
            bonus = (5.0 + (10.0 if robot == "drone" else 0.0)) if reached else 0.0
            assert abs(r[i] - (d0 - d1 + bonus)) < 1e-5
            is_trunc = ep_len[i] >= TL and not reached
            assert bool(done[i]) == bool(reached or is_trunc) and bool(trunc[i]) == bool(is_trunc)
            pos2, vel2, goal2 = (x[:P] for x in env.state(i))
            if not done[i]:
                assert np.allclose(pos2, pos1, atol=1e-12) and np.allclose(vel2, vel1, atol=1e-12) and np.array_equal(goal2, goal0)
                rel = goal0 - pos1
                assert np.allclose(o[i, :P], rel / (np.linalg.norm(rel) + 1e-6), atol=1e-6)
                assert np.allclose(o[i, P:2 * P], vel1, atol=1e-6) and np.allclose(o[i, 2 * P:3 * P], pos1, atol=1e-6)
            else:
                n_done += 1
                n_goal += int(reached)
                ep_len[i] = 0
                assert not np.array_equal(goal2, goal0)                      # always a new goal
                if reached:                                                  # lazy reset: pose kept
                    assert np.allclose(pos2, pos1, atol=1e-12) and np.allclose(vel2, vel1, atol=1e-12)
                else:                                                        # time limit: pose from the init space
                    assert np.all(np.abs(pos2) <= env.extent / 2 + 1e-9) and np.all(vel2 == 0.0)
                    rel = goal0 - pos1                                       # terminal observation = pre-reset state
                    assert np.allclose(term[i, :P], rel / (np.linalg.norm(rel) + 1e-6), atol=1e-6)
                    assert np.allclose(term[i, 2 * P:3 * P], pos1, atol=1e-6)
                assert np.allclose(o[i, 2 * P:3 * P], pos2, atol=1e-6)       # returned obs is the post-reset one
        assert nt == int(trunc.sum())
    st = env.episode_stats()
    assert st["episodes"] == n_done and st["goals"] == n_goal and n_done > N
    if D > 3 * P:
        pad = env.reset()[:, 3 * P:]
        assert abs(pad.std() - 0.1) < 0.01
    env.close()


def test_native_env_vecenv_contract_and_buffers():
    env = _make("car", 8, 5, terminate_on_goal=False)
    obs = env.reset()
    o, r, d, infos = env.step(np.zeros((8, 2), np.float32))
    assert o.shape == (8, 26) and r.shape == (8,) and d.dtype == bool and len(infos) == 8
    for _ in range(4):
        o, r, d, infos = env.step(np.zeros((8, 2), np.float32))
    assert d.all() and all(i["TimeLimit.truncated"] and i["terminal_observation"].shape == (26,) for i in infos)
    buf = np.zeros((8, 26), np.float32)
    env.use_buffers(obs=buf)
    env.step_arrays(np.zeros((8, 2), np.float32))
    assert np.abs(buf).sum() > 0                                             # results land in the caller's array
    with pytest.raises(ValueError):
        env.use_buffers(obs=np.zeros((8, 25), np.float32))
    from mobrob_amd.envs.native_env import NativeGoalVecEnv
    with pytest.raises(ValueError):
        NativeGoalVecEnv.for_robot("submarine", 4)
    env.close()
