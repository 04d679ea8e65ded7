"""Closed-loop sanity of the served host collector: doggo (2x256) on the native C goal env through PPO.learn, 4096 envs; the goal rate and
the mean episode return must rise.  python scratch/soak_served.py [iterations]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MOBROB_COLLECT_SERVER", "2")
from mobrob_amd.rl_control.ppo import PPOCtrl  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
N, T = 4096, 250
cfg = {"ppo_kwargs": {"policy": "MlpPolicy", "n_steps": T, "batch_size": 65536, "n_epochs": 5, "gamma": 0.99, "gae_lambda": 0.95,
                      "ent_coef": 0.0, "clip_range": 0.2, "policy_kwargs": {"net_arch": {"pi": [256, 256], "vf": [256, 256]}}},
       "env_name": "doggo", "time_limit": 200, "n_envs": N, "vec_env_type": "native", "enable_gui": False, "seed": 0}
c = PPOCtrl.from_config(cfg)


t0 = time.time()
for it in range(iters):   # one iteration per call: the default callback keeps the native (served) collector
    c.learn(total_timesteps=N * T, reset_num_timesteps=(it == 0))
    st = c.ppo.device_episode_stats
    if st and st["episodes"] and (it % 3 == 0 or it == iters - 1):
        print(f"iteration {it:3d}  timesteps {c.ppo.num_timesteps:9d}  episodes {st['episodes']:6d}  goal rate {st['goals'] / st['episodes']:.3f}  "
              f"ep_rew_mean {st['ep_rew_mean']:.3f}  ep_len_mean {st['ep_len_mean']:.1f}", flush=True)
print("done", round(time.time() - t0, 1), "s")
