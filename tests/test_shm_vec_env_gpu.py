"""GPU: the worker-process environment layer (`ShmVecEnv`, `vec_env_type: subproc`) through the engine: the shared
block is pinned + mapped in place (mobrob_ppo_host_register), the rollout kernels read observations / write clipped
actions in the memory the workers use, and the pipelined collector overlaps worker stepping with the policy kernel.
Reference: SubprocVecEnv behind make_vec_env, /root/reference/src/mobrob/rl_control/ppo.py:30-48."""
import functools

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cfg(vec_env_type, n_envs=24, n_steps=40, env_name="point", time_limit=15, seed=2, hidden=64):
    return {"ppo_kwargs": {"policy": "MlpPolicy", "n_steps": n_steps, "batch_size": 240, "n_epochs": 2, "gamma": 0.99,
                           "gae_lambda": 0.95, "ent_coef": 0.0, "clip_range": 0.2,
                           "policy_kwargs": {"net_arch": {"pi": [hidden, hidden], "vf": [hidden, hidden]}}},
            "env_name": env_name, "time_limit": time_limit, "n_envs": n_envs, "vec_env_type": vec_env_type,
            "enable_gui": False, "seed": seed}


BUFS = ("obs", "actions", "rewards", "episode_starts", "values", "log_probs", "advantages", "returns", "last_values")


def _rollout_buffers(ppo):
    ppo.engine.synchronize()
    return {k: ppo.engine.read(k) for k in BUFS}


def test_registered_shared_block_is_accepted_as_pinned_memory():
    from mobrob_amd.engine import PPOEngine
    from mobrob_amd.envs.shm_vec_env import ShmVecEnv
    from mobrob_amd.envs.wrapper import get_env
    env = ShmVecEnv([functools.partial(get_env, env_name="point", terminate_on_goal=True, time_limit=10)] * 8, seed=0, n_workers=2)
    e = PPOEngine(obs_dim=14, act_dim=2, n_envs=8, n_steps=4, batch_size=16, n_epochs=1)
    b = env.buffers()
    with pytest.raises(ValueError):   # not yet device visible: the pipelined calls refuse pageable memory
        e.rollout_begin()
        e.part_pipeline(2, b["obs"], b["clip"], b["rew"], b["done"], b["trunc"], b["term"]).act(0)
    e.register_host(*env.shared_block())
    env.reset()
    e.rollout_begin()
    pipe = e.part_pipeline(2, b["obs"], b["clip"], b["rew"], b["done"], b["trunc"], b["term"])
    pipe.act(0); pipe.act(1); pipe.wait(0); pipe.wait(1)
    assert np.all(np.abs(b["clip"]) <= 1.0) and np.any(b["clip"] != 0.0)   # the kernel wrote into the workers' block
    e.unregister_host(env.shared_block()[0])
    env.close()
    e.close()


@pytest.mark.parametrize("hidden,n_envs", [(64, 24), (256, 24), (64, 96)])   # 96 envs: two pipelined row ranges; 24: one range
def test_subproc_rollout_and_update_equal_the_in_process_path(hidden, n_envs):
    """Same seeds -> the worker-process layer (pipelined over two row ranges, shared block read in place) and the
    in-process loop (`dummy`: HostVecEnv, staged copies, info dicts) fill bit-identical rollout buffers, and the
    update that follows leaves identical parameters."""
    from mobrob_amd.rl_control.ppo import BaseCallback, PPOCtrl
    out = {}
    for kind in ("dummy", "subproc"):
        ctrl = PPOCtrl.from_config(_cfg(kind, hidden=hidden, n_envs=n_envs))
        ppo = ctrl.ppo
        ppo.learn(total_timesteps=n_envs * 40)
        out[kind] = (ppo.engine.get_flat_params(), list(ppo.ep_info_buffer), ppo.num_timesteps)
        cb = BaseCallback()
        cb.init_callback(ppo)
        ppo._collect_rollouts(cb)     # a second rollout, with the updated policy, left in the buffers
        out[kind] += (_rollout_buffers(ppo),)
        ppo.env.close()
        ppo.engine.close()
    (p_d, ep_d, n_d, buf_d), (p_s, ep_s, n_s, buf_s) = out["dummy"], out["subproc"]
    assert n_d == n_s == n_envs * 40
    for k in BUFS:
        assert np.array_equal(buf_d[k], buf_s[k]), k
    assert np.array_equal(p_d, p_s)
    assert len(ep_d) > 0 and [(e["r"], e["l"]) for e in ep_d] == [(e["r"], e["l"]) for e in ep_s]   # real Monitor records


def test_subproc_training_learns_and_saves(tmp_path):
    from mobrob_amd.rl_control.ppo import PPO, PPOCtrl, CheckpointCallback
    cfg = _cfg("subproc", n_envs=64, n_steps=64, time_limit=100, seed=0)
    cfg["ppo_kwargs"].update(batch_size=1024, n_epochs=10)
    ctrl = PPOCtrl.from_config(cfg)
    ppo = ctrl.ppo
    assert type(ppo.env).__name__ == "ShmVecEnv" and ppo.env.n_workers >= 1
    lens = []
    cb = CheckpointCallback(save_freq=64 * 5, save_path=str(tmp_path), name_prefix="timestep")
    for it in range(30):
        ppo.learn(total_timesteps=64 * 64, reset_num_timesteps=False, callback=cb if it == 0 else None)
        lens.append(np.mean([e["l"] for e in ppo.ep_info_buffer]))
    assert lens[-1] < 0.75 * lens[2], lens                     # episodes get shorter: the robots learn to reach goals
    ctrl.save_model(str(tmp_path / "point-ppo.zip"))
    again = PPO.load(str(tmp_path / "point-ppo.zip"))
    assert np.array_equal(again.engine.get_flat_params(), ppo.engine.get_flat_params())
    assert [e["l"] for e in again.ep_info_buffer] == [e["l"] for e in ppo.ep_info_buffer]
    ppo.env.close()


def test_set_env_rebinds_the_staging_of_array_protocol_envs():
    """after set_env() the new native env must write into the staging the engine reads (it used to keep
    writing into its own unpinned arrays while the engine read stale pinned ones)."""
    from mobrob_amd.envs.native_env import NativeGoalVecEnv
    from mobrob_amd.rl_control.ppo import PPOCtrl
    cfg = _cfg("native", n_envs=32, n_steps=16, time_limit=50)
    ppo = PPOCtrl.from_config(cfg).ppo
    ppo.learn(total_timesteps=32 * 16)
    new_env = NativeGoalVecEnv.for_robot("point", 32, time_limit=50, seed=77)
    ppo.set_env(new_env)
    assert ppo._host_bufs is None
    ppo.learn(total_timesteps=32 * 16)
    assert new_env._obs is ppo._host_bufs["obs"]               # the env writes where the engine reads
    ppo.engine.synchronize()
    stored = ppo.engine.read("obs")[16]                        # last_obs slot = what the env wrote last
    assert np.array_equal(stored, ppo._host_bufs["obs"])
    # a fresh reference env with the same seed went through the same states only if ITS observations were used
    twin = NativeGoalVecEnv.for_robot("point", 32, time_limit=50, seed=77)
    first = twin.reset().copy()
    assert np.array_equal(ppo.engine.read("obs")[0], first)
    ppo.engine.close()
    assert np.isfinite(ppo._host_bufs["obs"]).all()            # views outlive the engine (no use-after-free)
