import sys, time, numpy as np
sys.path.insert(0, '.')
from mobrob_amd.rl_control.ppo import PPOCtrl
robot = sys.argv[1] if len(sys.argv) > 1 else "point"
H = int(sys.argv[2]) if len(sys.argv) > 2 else 64
cfg = {"ppo_kwargs": {"policy": "MlpPolicy", "n_steps": 128, "batch_size": 4096, "n_epochs": 10, "gamma": 0.99,
                      "gae_lambda": 0.95, "ent_coef": 0.0, "clip_range": 0.2,
                      "policy_kwargs": {"net_arch": {"pi": [H, H], "vf": [H, H]}}},
       "env_name": robot, "time_limit": 200, "n_envs": 1024, "vec_env_type": "device_goal", "enable_gui": False, "seed": 0}
ctrl = PPOCtrl.from_config(cfg)
ppo = ctrl.ppo
t0 = time.time()
for it in range(40):
    ppo.learn(total_timesteps=128 * 1024, reset_num_timesteps=False)
    st = ppo.device_episode_stats
    if it % 3 == 0 or it == 39:
        print(f"iter {it:3d} steps {ppo.num_timesteps:9d} episodes {st['episodes']:6d} goal_rate {st['goals']/max(st['episodes'],1):.3f} "
              f"ep_rew_mean {st['ep_rew_mean']:7.3f} ep_len_mean {st['ep_len_mean']:7.2f}  t={time.time()-t0:.2f}s")
