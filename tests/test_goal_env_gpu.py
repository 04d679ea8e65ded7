"""GPU: device-resident goal environment (reference EnvWrapper rules on the GPU) and closed-loop learning."""
import numpy as np
import pytest

from oracle import ppo_oracle as O
from tests.util import scaled_err

pytestmark = pytest.mark.gpu


def _env(robot, n, tl, **kw):
    from mobrob_amd.envs.vec_env import DeviceGoalVecEnv
    return DeviceGoalVecEnv.for_robot(robot, n, time_limit=tl, **kw)


@pytest.mark.parametrize("robot", ["point", "drone", "doggo"])
def test_goal_env_rollout_follows_the_wrapper_rules(robot):
    from mobrob_amd.engine import PPOEngine
    from mobrob_amd.envs.wrapper import ROBOT_DIMS
    D, A, P = ROBOT_DIMS[robot]
    N, T, TL, H = 128, 64, 25, 64
    env = _env(robot, N, TL)
    p = O.init_params(D, A, (H, H), (H, H), seed=4)
    p["value_net.bias"] = np.array([3.0], np.float32)
    p["value_net.weight"] *= 0.0  # V == 3 everywhere: a bootstrapped reward is progress + 0.99 * 3
    e = PPOEngine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=1024, n_epochs=1, seed=11)
    e.set_params(p)
    env.collect(e)
    e.synchronize()
    obs, act, rew, es = e.read("obs"), e.read("actions"), e.read("rewards"), e.read("episode_starts")
    assert np.all(es[0] == 1.0)
    unit, vel, pos = obs[..., :P], obs[..., P:2 * P], obs[..., 2 * P:3 * P]
    assert np.allclose(np.linalg.norm(unit, axis=-1), 1.0, atol=1e-3)
    assert np.all(np.abs(pos) <= env.extent + 1e-6)
    if D > 3 * P:  # padding features: N(0, 0.1^2)
        pad = obs[..., 3 * P:]
        assert abs(pad.mean()) < 0.01 and abs(pad.std() - 0.1) < 0.01
    # kinematics on every transition that did not end an episode: vel' = 0.8 vel + 0.2 mix.clip(a), pos' = clip(pos + dt vel')
    cont = es[1:] == 0.0                                   # [T-1, N]: step t did not finish an episode
    cmd = np.clip(act, -1.0, 1.0) @ env.mix.T               # [T, N, P]
    v_next = 0.8 * vel[:-1] + 0.2 * cmd
    p_next = np.clip(pos[:-1] + env.dt * v_next, -env.extent, env.extent)
    m = cont[..., None] & np.ones(P, bool)
    assert np.max(np.abs(v_next[:-1][m] - vel[1:-1][m])) < 1e-5
    assert np.max(np.abs(p_next[:-1][m] - pos[1:-1][m])) < 1e-5
    # reward on those transitions = progress towards the goal (|goal-prev| - |goal-cur|): first order  unit . (pos' - pos)
    dpos = pos[1:-1] - pos[:-2]
    prog = np.sum(unit[:-2] * dpos, axis=-1)
    r = rew[:-1]
    assert np.max(np.abs((r - prog)[cont])) < 0.02
    assert np.all(np.abs(r[cont]) <= np.linalg.norm(dpos, axis=-1)[cont] + 1e-5)   # triangle inequality
    # episode ends: either the goal was reached (bonus) or the time limit hit (bootstrap with V ~ 3)
    ended = ~cont
    assert ended.any()
    bonus = 5.0 + (10.0 if robot == "drone" else 0.0)
    re = r[ended]
    near_bonus = np.abs(re - bonus) < 0.2
    near_boot = np.abs(re - 0.99 * 3.0) < 0.1
    assert np.all(near_bonus | near_boot), re[~(near_bonus | near_boot)][:5]
    # a lazily reset robot (reached its goal) keeps its pose; a timed-out one restarts inside the init space
    t_idx, n_idx = np.nonzero(ended)
    for t, n, b in zip(t_idx, n_idx, near_bonus):
        new = pos[t + 1, n]
        if not b:
            assert np.all(np.abs(new) <= env.extent / 2 + 1e-6) and np.all(vel[t + 1, n] == 0.0)
    # Monitor statistics match the stored episode boundaries
    st = e.episode_stats(reset=True)
    n_done = int(ended.sum() + e.read("last_dones").sum())
    assert st["episodes"] == n_done and 0 < st["ep_len_mean"] <= TL
    assert st["goals"] <= st["episodes"]
    assert e.episode_stats()["episodes"] == 0
    # GAE on the stored rollout is the oracle's
    adv, _ = O.gae(rew, e.read("values"), es, e.read("last_values"), e.read("last_dones") > 0, 0.99, 0.95)
    assert np.array_equal(e.read("advantages"), adv)
    # stored values / log-probs are the policy's on the stored observations
    mean, val = O.policy_outputs(p, obs[:T].reshape(T * N, D))
    assert np.max(np.abs(e.read("values").reshape(-1) - val)) < 1e-4
    e.close()


def test_truncation_bootstrap_uses_the_terminal_observation():
    from mobrob_amd.engine import PPOEngine
    from mobrob_amd.envs.wrapper import ROBOT_DIMS
    D, A, P = ROBOT_DIMS["car"]
    N, T, TL = 64, 8, 4
    env = _env("car", N, TL, terminate_on_goal=False)   # never terminates: every env is truncated at steps 3 and 7
    p = O.init_params(D, A, seed=6)
    e = PPOEngine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=128, n_epochs=1, seed=3)
    e.set_params(p)
    env.collect(e)
    e.synchronize()
    assert np.all(e.read("truncated") == 1)
    tobs = e.read("terminal_obs")[:, :D]
    _, v = O.policy_outputs(p, tobs)
    assert scaled_err(e.read("terminal_values"), v) < 1e-4
    # the terminal observation continues the trajectory (pre-reset pose), the stored next observation is the reset one
    obs = e.read("obs")
    pos_prev, vel_prev = obs[T - 1][:, 2 * P:3 * P], obs[T - 1][:, P:2 * P]
    cmd = np.clip(e.read("actions")[T - 1], -1, 1) @ env.mix.T
    v_next = 0.8 * vel_prev + 0.2 * cmd
    assert np.max(np.abs(tobs[:, P:2 * P] - v_next)) < 1e-5
    assert np.max(np.abs(tobs[:, 2 * P:3 * P] - np.clip(pos_prev + env.dt * v_next, -env.extent, env.extent))) < 1e-5
    assert np.all(obs[T][:, P:2 * P] == 0.0)  # reset: velocity zero, pose from the init space
    e.close()


def test_ppo_learns_to_reach_goals_on_the_device_env():
    """Closed loop through every kernel of the path: rollout forward + sampling, env rules, storage, bootstrap, GAE,
    minibatch update.  With correct gradients the point robot learns to drive to its goal within ~1.5 M steps."""
    from mobrob_amd.rl_control.ppo import PPOCtrl
    cfg = {"ppo_kwargs": {"policy": "MlpPolicy", "n_steps": 128, "batch_size": 4096, "n_epochs": 10, "gamma": 0.99,
                          "gae_lambda": 0.95, "ent_coef": 0.0, "clip_range": 0.2,
                          "policy_kwargs": {"net_arch": {"pi": [64, 64], "vf": [64, 64]}}},
           "env_name": "point", "time_limit": 200, "n_envs": 256, "vec_env_type": "device_goal", "enable_gui": False,
           "seed": 0}
    ctrl = PPOCtrl.from_config(cfg)
    hist = []

    ppo = ctrl.ppo
    for it in range(45):
        ppo.learn(total_timesteps=128 * 256, reset_num_timesteps=False)
        st = ppo.device_episode_stats
        hist.append((st["episodes"], st["goals"], st["ep_rew_mean"], st["ep_len_mean"]))
    first = np.array(hist[:5], dtype=np.float64)
    last = np.array(hist[-5:], dtype=np.float64)
    goal_rate_first = first[:, 1].sum() / max(first[:, 0].sum(), 1)
    goal_rate_last = last[:, 1].sum() / max(last[:, 0].sum(), 1)
    # untrained: most episodes run into the time limit; trained: nearly all end at the goal, and quickly
    assert goal_rate_last > 0.9 and goal_rate_last > goal_rate_first + 0.3, (goal_rate_first, goal_rate_last, hist[-1])
    assert np.nanmean(last[:, 3]) < 0.6 * np.nanmean(first[:, 3]), (first[:, 3], last[:, 3])
    assert np.nanmean(last[:, 2]) > np.nanmean(first[:, 2]) + 1.0


def test_policy_trained_on_the_device_env_solves_the_host_env(tmp_path, monkeypatch):
    """Device goal env == host `KinematicGoalEnv` task: train on the GPU env, save an SB3 zip, load it through
    `load_policy` (examples/control.py path) and roll it out deterministically on the HOST environment."""
    import mobrob_amd.utils as U
    from mobrob_amd import get_env
    from mobrob_amd.rl_control.ppo import PPOCtrl
    cfg = {"ppo_kwargs": {"policy": "MlpPolicy", "n_steps": 128, "batch_size": 4096, "n_epochs": 10, "gamma": 0.99,
                          "gae_lambda": 0.95, "ent_coef": 0.0, "clip_range": 0.2,
                          "policy_kwargs": {"net_arch": {"pi": [64, 64], "vf": [64, 64]}}},
           "env_name": "point", "time_limit": 200, "n_envs": 512, "vec_env_type": "device_goal", "enable_gui": False,
           "seed": 1}
    ctrl = PPOCtrl.from_config(cfg)
    ctrl.learn(total_timesteps=30 * 128 * 512)
    monkeypatch.setattr(U, "DATA_DIR", str(tmp_path))
    ctrl.save_model(f"{tmp_path}/policies/point-ppo.zip")
    policy = U.load_policy("point", "ppo")
    env = get_env("point", terminate_on_goal=True, time_limit=200)
    reached, lengths = 0, []
    for ep in range(12):
        obs, _ = env.reset(seed=100 + ep)
        for t in range(200):
            action, _ = policy.predict(obs, deterministic=True)
            obs, reward, terminated, truncated, _ = env.step(action)
            if terminated or truncated:
                break
        reached += int(terminated)
        lengths.append(t + 1)
    assert reached >= 10, (reached, lengths)          # an untrained policy reaches ~40 % of the goals within 200 steps
    assert np.mean(lengths) < 120, lengths
    # the evaluation script itself (examples/control.py, the reference's protocol: 1000 steps per epoch, reset and go on at
    # every goal): a policy that reaches a goal in < 120 steps collects >= 8 arrival bonuses of +5 per epoch
    from tests.test_control_cli import _load_script
    rewards = _load_script().simulate("point", "ppo", epochs=2)
    assert len(rewards) == 2 and min(rewards) > 30.0, rewards


def test_ppo_learns_with_the_native_host_env_through_pinned_staging():
    """Host-env path at scale: the multi-threaded C environment steps 512 robots, observations / rewards / flags
    travel through pinned staging on the side stream, actions come back once per step -- and PPO learns the task."""
    from mobrob_amd.rl_control.ppo import PPOCtrl
    cfg = {"ppo_kwargs": {"policy": "MlpPolicy", "n_steps": 128, "batch_size": 4096, "n_epochs": 10, "gamma": 0.99,
                          "gae_lambda": 0.95, "ent_coef": 0.0, "clip_range": 0.2,
                          "policy_kwargs": {"net_arch": {"pi": [64, 64], "vf": [64, 64]}}},
           "env_name": "point", "time_limit": 200, "n_envs": 512, "vec_env_type": "native", "enable_gui": False, "seed": 0}
    ctrl = PPOCtrl.from_config(cfg)
    ppo = ctrl.ppo
    hist = []
    for it in range(24):
        ppo.learn(total_timesteps=128 * 512, reset_num_timesteps=False)
        st = ppo.device_episode_stats
        hist.append((st["episodes"], st["goals"], st["ep_len_mean"]))
    h = np.array(hist, dtype=np.float64)
    first, last = h[:4].sum(0), h[-4:].sum(0)
    assert last[1] / max(last[0], 1) > 0.9 and last[1] / max(last[0], 1) > first[1] / max(first[0], 1) + 0.3, (first, last)
    assert ppo.num_timesteps == 24 * 128 * 512


def test_the_reference_doggo_yaml_on_native_host_envs_is_served_by_the_rollout_kernel(monkeypatch):
    """data/configs/doggo-ppo.yaml as the reference ships it (16 environments, n_steps 1000, batch 100, 2x64) with the environments on
    the host (`vec_env_type: native`): PPO.learn runs the whole collector loop in C as ONE row range of half a tile, served by
    k_rollout64_tile<.., 3> -- MOBROB_COLLECT_SERVER=2 makes any fallback an error -- and the loop's counters are the reference's
    (/root/reference/data/configs/doggo-ppo.yaml:3-23 through /root/reference/src/mobrob/rl_control/ppo.py:61-74)."""
    import yaml
    from mobrob_amd.rl_control.ppo import PPOCtrl
    from mobrob_amd.utils import DATA_DIR
    with open(f"{DATA_DIR}/configs/doggo-ppo.yaml") as f:
        cfg = yaml.safe_load(f)
    cfg["vec_env_type"] = "native"
    monkeypatch.setenv("MOBROB_COLLECT_SERVER", "2")
    monkeypatch.setenv("MOBROB_SERVER_TIMEOUT_S", "10")
    ctrl = PPOCtrl.from_config(cfg)
    ppo = ctrl.ppo
    n_envs, n_steps = cfg["n_envs"], cfg["ppo_kwargs"]["n_steps"]
    assert n_envs < 32                                    # half a tile: the shape this test is about
    ppo.learn(total_timesteps=2 * n_envs * n_steps)
    assert ppo.num_timesteps == 2 * n_envs * n_steps and ppo._n_updates == 2 * cfg["ppo_kwargs"]["n_epochs"]
    st = ppo.device_episode_stats
    assert st["episodes"] > 0 and np.isfinite(st["ep_rew_mean"])
    assert all(np.isfinite(v).all() for v in ppo.engine.get_params().values())
    ppo.engine.close()


def test_ep_info_buffer_holds_real_monitor_records_and_the_zip_keeps_the_robot_bounds(tmp_path):
    """Device goal env and native host env: `ep_info_buffer` gets the (return, length) of individual finished episodes
    (it used to get the rollout mean repeated), consistent with the aggregated counters; a saved drone model carries
    the drone's finite observation bounds and PPO.load restores them."""
    from mobrob_amd import checkpoint as ck
    from mobrob_amd.envs.wrapper import observation_space_of
    from mobrob_amd.rl_control.ppo import PPO, PPOCtrl
    for kind in ("device_goal", "native"):
        cfg = {"ppo_kwargs": {"policy": "MlpPolicy", "n_steps": 64, "batch_size": 1024, "n_epochs": 2},
               "env_name": "drone", "time_limit": 30, "n_envs": 64, "vec_env_type": kind, "enable_gui": False, "seed": 3}
        ctrl = PPOCtrl.from_config(cfg)
        ppo = ctrl.ppo
        ppo.learn(total_timesteps=64 * 64)
        st = ppo.device_episode_stats
        recs = list(ppo.ep_info_buffer)
        assert st["episodes"] >= 64 * 2 and len(recs) == 100 and ppo._episode_num == st["episodes"]
        lens = np.array([r["l"] for r in recs]); rets = np.array([r["r"] for r in recs])
        assert lens.min() >= 1 and lens.max() <= 30 and np.all(lens == lens.astype(int))
        assert len(set(np.round(rets, 5))) > 20, kind              # individual episodes, not one mean repeated
        assert (lens == 30).any()                                   # most episodes of an untrained policy hit the time limit
        assert abs(rets.mean() - st["ep_rew_mean"]) < 3 * rets.std() + 1e-3
        path = str(tmp_path / f"drone-{kind}.zip")
        ctrl.save_model(path)
        c = ck.load_zip(path)
        sp = observation_space_of("drone")
        assert np.array_equal(c["data"]["observation_space"]["low"], sp.low) and np.array_equal(c["data"]["observation_space"]["high"], sp.high)
        assert [e["l"] for e in c["data"]["ep_info_buffer"]] == [r["l"] for r in recs]
        again = PPO.load(path)
        assert np.array_equal(again.obs_bounds[0], sp.low)
        again.save(str(tmp_path / "again.zip"))                     # no env attached: the bounds come from the checkpoint
        assert np.array_equal(ck.load_zip(str(tmp_path / "again.zip"))["data"]["observation_space"]["high"], sp.high)
        ppo.engine.close()
