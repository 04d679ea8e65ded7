"""Closed-loop training at the reference YAML shape (16 envs x 1000 steps, batch 100, 2x64: k_rollout64_tile +
k_split64_train + norm records) on the device goal env: finite parameters, the task is learnt."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobrob_amd.rl_control.ppo import PPOCtrl
for robot in ("point", "car", "doggo"):
    cfg = {"ppo_kwargs": {"policy": "MlpPolicy", "n_steps": 1000, "batch_size": 100, "n_epochs": 5, "gamma": 0.99,
                          "gae_lambda": 0.95, "ent_coef": 0.0, "clip_range": 0.2,
                          "policy_kwargs": {"net_arch": {"pi": [64, 64], "vf": [64, 64]}}},
           "env_name": robot, "time_limit": 200, "n_envs": 16, "vec_env_type": "device_goal", "enable_gui": False, "seed": 0}
    ppo = PPOCtrl.from_config(cfg).ppo
    t0 = time.time()
    first = None
    for it in range(60):
        ppo.learn(total_timesteps=16000, reset_num_timesteps=False)
        st = ppo.device_episode_stats
        if it == 1:
            first = st
    p = ppo.engine.get_flat_params()
    assert np.isfinite(p).all()
    print(f"{robot:8s}: {ppo.num_timesteps/1e6:5.2f} M steps in {time.time()-t0:5.1f} s ({ppo.num_timesteps/(time.time()-t0)/1e3:.0f} k steps/s incl. Python) | "
          f"goal rate {first['goals']/max(first['episodes'],1):.2f} -> {st['goals']/max(st['episodes'],1):.3f}, ep_len {first['ep_len_mean']:.0f} -> {st['ep_len_mean']:.1f}", flush=True)
    ppo.engine.close()
