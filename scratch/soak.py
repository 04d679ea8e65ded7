"""Soak: long closed-loop training on the device goal env for every robot (numerical stability, no NaN, learning)."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobrob_amd.rl_control.ppo import PPOCtrl
for robot, H, iters in (("point", 64, 150), ("car", 64, 150), ("turtlebot3", 64, 150), ("drone", 64, 250), ("doggo", 256, 150)):
    cfg = {"ppo_kwargs": {"policy": "MlpPolicy", "n_steps": 128, "batch_size": 8192, "n_epochs": 10, "gamma": 0.99,
                          "gae_lambda": 0.95, "ent_coef": 0.0, "clip_range": 0.2,
                          "policy_kwargs": {"net_arch": {"pi": [H, H], "vf": [H, H]}}},
           "env_name": robot, "time_limit": 200, "n_envs": 2048, "vec_env_type": "device_goal", "enable_gui": False, "seed": 0}
    ctrl = PPOCtrl.from_config(cfg)
    ppo = ctrl.ppo
    t0 = time.time()
    first = None
    for it in range(iters):
        ppo.learn(total_timesteps=128 * 2048, reset_num_timesteps=False)
        st = ppo.device_episode_stats
        if it == 3:
            first = st
    p = ppo.engine.get_flat_params()
    assert np.isfinite(p).all()
    print(f"{robot:11s} 2x{H}: {ppo.num_timesteps/1e6:6.1f} M steps in {time.time()-t0:5.1f} s | goal rate {first['goals']/max(first['episodes'],1):.2f} -> "
          f"{st['goals']/max(st['episodes'],1):.3f}, ep_len {first['ep_len_mean']:.0f} -> {st['ep_len_mean']:.1f}, ep_rew {first['ep_rew_mean']:.2f} -> {st['ep_rew_mean']:.2f}, "
          f"|params|max {np.abs(p).max():.2f}", flush=True)
    ppo.engine.close()
