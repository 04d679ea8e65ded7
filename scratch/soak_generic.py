"""Closed-loop check of the generic chain's keyword surface: does PPO LEARN the goal task on the device env with other activations,
depths and gSDE?  (The parity tests pin arithmetic; this pins that nothing systematic -- a sign, a schedule -- is wrong with it.)

    gpurun -- python scratch/soak_generic.py > profiles/r6/soak_generic.txt
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobrob_amd.rl_control.ppo import PPOCtrl

CASES = [("tanh 2x64 (fused family, for scale)", {}, {"net_arch": [64, 64]}),
         ("ELU 3x64", {}, {"net_arch": [64, 64, 64], "activation_fn": "ELU"}),
         ("ReLU pi [128] vf [64, 64, 32, 32]", {}, {"net_arch": {"pi": [128], "vf": [64, 64, 32, 32]}, "activation_fn": "ReLU"}),
         ("SiLU 2x64", {}, {"net_arch": [64, 64], "activation_fn": "SiLU"}),
         ("gSDE tanh 2x64, sde_sample_freq 4, log_std_init -2", {"use_sde": True, "sde_sample_freq": 4}, {"net_arch": [64, 64], "log_std_init": -2.0}),
         ("gSDE ELU 2x64, full_std=False, use_expln", {"use_sde": True, "sde_sample_freq": 8},
          {"net_arch": [64, 64], "activation_fn": "ELU", "log_std_init": -2.0, "full_std": False, "use_expln": True})]
for name, kw, pk in CASES:
    cfg = {"ppo_kwargs": {"policy": "MlpPolicy", "n_steps": 128, "batch_size": 4096, "n_epochs": 10, "gamma": 0.99, "gae_lambda": 0.95,
                          "ent_coef": 0.0, "clip_range": 0.2, "policy_kwargs": pk, **kw},
           "env_name": "point", "time_limit": 200, "n_envs": 1024, "vec_env_type": "device_goal", "enable_gui": False, "seed": 0}
    ppo = PPOCtrl.from_config(cfg).ppo
    t0 = time.time()
    line = []
    for it in range(30):
        ppo.learn(total_timesteps=128 * 1024, reset_num_timesteps=False)
        st = ppo.device_episode_stats
        if it in (0, 9, 19, 29):
            line.append(f"it {it:2d}: goal rate {st['goals'] / max(st['episodes'], 1):.3f} ep_rew {st['ep_rew_mean']:7.2f} ep_len {st['ep_len_mean']:6.1f}")
    print(f"{name:58s} | " + " | ".join(line) + f" | {time.time() - t0:.1f} s", flush=True)
    ppo.engine.close()
