"""Closed-loop training of doggo (58/12, 2x256) on the device goal env with the three gradient kernels an engine can have -- k_chain_train
(default), k_fused_train<.., X3> (MOBROB_NO_CHAIN=1), all products on v_mfma_f32 (MOBROB_NO_X3=1): the task is learnt the same way
(goal rate, episode length, reward after the same number of steps)."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobrob_amd.rl_control.ppo import PPOCtrl
for mode in ("chain", "x3", "f32"):
    x3 = mode != "f32"
    os.environ.pop("MOBROB_NO_X3", None)
    os.environ.pop("MOBROB_NO_CHAIN", None)
    if mode == "f32":
        os.environ["MOBROB_NO_X3"] = "1"
    if mode == "x3":
        os.environ["MOBROB_NO_CHAIN"] = "1"
    cfg = {"ppo_kwargs": {"policy": "MlpPolicy", "n_steps": 128, "batch_size": 8192, "n_epochs": 10, "gamma": 0.99,
                          "gae_lambda": 0.95, "ent_coef": 0.0, "clip_range": 0.2,
                          "policy_kwargs": {"net_arch": {"pi": [256, 256], "vf": [256, 256]}}},
           "env_name": "doggo", "time_limit": 200, "n_envs": 2048, "vec_env_type": "device_goal", "enable_gui": False, "seed": 0}
    ctrl = PPOCtrl.from_config(cfg)
    ppo = ctrl.ppo
    assert ppo.engine.x3_mode() == {"chain": 7, "x3": 3, "f32": 0}[mode]
    t0 = time.time()
    rows = []
    for it in range(120):
        ppo.learn(total_timesteps=128 * 2048, reset_num_timesteps=False)
        st = ppo.device_episode_stats
        if it in (3, 30, 60, 119):
            rows.append((it, st["goals"] / max(st["episodes"], 1), st["ep_len_mean"], st["ep_rew_mean"]))
    p = ppo.engine.get_flat_params()
    assert np.isfinite(p).all()
    print(f"{mode:5s} (x3_mode {ppo.engine.x3_mode()}): {ppo.num_timesteps/1e6:.1f} M steps in {time.time()-t0:.1f} s; (iteration, goal rate, ep_len, ep_rew): "
          + "; ".join(f"({i}, {g:.3f}, {l:.1f}, {r:.2f})" for i, g, l, r in rows), flush=True)
    ppo.engine.close()
