"""Does the per-launch HIP-event bracketing of bench.py (engine.profile) cost throughput?  Headline shape, 4 iterations each."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobrob_amd.engine import PPOEngine
from mobrob_amd.rl_control.init import orthogonal_policy_init
D, A, H, N, T, E, B = 58, 12, 256, 4096, 1000, 5, 65536
e = PPOEngine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=B, n_epochs=E, pi=(H, H), vf=(H, H), ent_coef=0.01, seed=0)
e.set_params(orthogonal_policy_init(D, A, (H, H), (H, H), 0))
def run(k):
    for _ in range(k):
        e.collect_synthetic(p_term=1 / 107.0, time_limit=1000)
        e.train(None)
    e.synchronize()
run(2)
for mode in (False, True, False, True):
    e.profile(mode)
    t0 = time.perf_counter(); run(4); dt = (time.perf_counter() - t0) / 4
    if mode: e.profile_read()
    print(f"profile={mode}: {dt*1e3:.2f} ms/iter  {N*T/dt/1e6:.2f} M env-steps/s")
