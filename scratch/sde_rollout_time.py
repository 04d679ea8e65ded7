"""gSDE rollout and update cost at a large shape (generic chain): python scratch/sde_rollout_time.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobrob_amd.engine import PPOEngine
from mobrob_amd.rl_control.init import policy_init
for sde in (False, True):
    D, A, N, T, H = 58, 12, 4096, 100, 256
    e = PPOEngine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=65536, n_epochs=1, pi=(H, H), vf=(H, H), activation="elu", use_sde=sde,
                  sde_sample_freq=4)
    e.set_params(policy_init(D, A, (H, H), (H, H), seed=0, log_std_init=-2.0 if sde else 0.0, use_sde=sde))
    e.collect_synthetic(p_term=0.02, time_limit=50); e.synchronize()
    t0 = time.perf_counter(); e.collect_synthetic(p_term=0.02, time_limit=50); e.synchronize(); t1 = time.perf_counter()
    e.compute_gae() if hasattr(e, "compute_gae") else None
    t2 = time.perf_counter(); e.train(None); e.synchronize(); t3 = time.perf_counter()
    print(f"use_sde={sde}: rollout {1e6 * (t1 - t0) / T:.1f} us per step ({N} envs), update {1e3 * (t3 - t2):.1f} ms for {T * N // 65536 + 1} minibatches", flush=True)
    e.close()
