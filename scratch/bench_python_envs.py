"""f3 figure: 4096 Python `EnvWrapper` instances (doggo-shaped KinematicGoalEnv behind get_env) stepped by
(a) `dummy`   = HostVecEnv, one serial loop in the learner process, and
(b) `subproc` = ShmVecEnv, worker processes on all host cores + shared GPU-visible block + two pipelined row ranges,
through PPO.learn (rollout + update), headline network (2x256).  usage: python scratch/bench_python_envs.py [n_envs] [n_steps]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def main():
    import __graft_entry__
    __graft_entry__.build()
    from mobrob_amd.rl_control.ppo import PPOCtrl
    n_envs = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    out = {"n_envs": n_envs, "n_steps": n_steps, "host_cores": os.cpu_count()}
    for kind in (("subproc",) if os.environ.get("MOBROB_ENV_WORKERS") else ("dummy", "subproc")):
        cfg = {"ppo_kwargs": {"policy": "MlpPolicy", "n_steps": n_steps, "batch_size": 65536, "n_epochs": 5, "gamma": 0.99,
                              "gae_lambda": 0.95, "ent_coef": 0.01, "clip_range": 0.2,
                              "policy_kwargs": {"net_arch": {"pi": [256, 256], "vf": [256, 256]}}},
               "env_name": "doggo", "time_limit": 1000, "n_envs": n_envs, "vec_env_type": kind, "enable_gui": False, "seed": 0}
        t0 = time.perf_counter()
        ctrl = PPOCtrl.from_config(cfg)
        t_build = time.perf_counter() - t0
        ppo = ctrl.ppo
        ppo.learn(total_timesteps=n_envs * n_steps)                                  # warm-up iteration
        t0 = time.perf_counter()
        iters = 3
        ppo.learn(total_timesteps=iters * n_envs * n_steps, reset_num_timesteps=False)
        ppo.engine.synchronize()
        dt = time.perf_counter() - t0
        out[kind] = {"env_steps_per_s": iters * n_envs * n_steps / dt, "ms_per_vector_step": 1e3 * dt / (iters * n_steps),
                     "build_s": t_build, "workers": getattr(ppo.env, "n_workers", 1)}
        ppo.env.close()
        ppo.engine.close()
    if "dummy" in out:
        out["speedup"] = out["subproc"]["env_steps_per_s"] / out["dummy"]["env_steps_per_s"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
