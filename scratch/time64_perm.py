"""Does the row permutation (random gathers) pace the 64-wide gradient kernels?  Identity permutation against a random one."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mobrob_amd.engine import PPOEngine
from mobrob_amd.rl_control.init import orthogonal_policy_init
for (D, A, N, T, B) in [(14, 2, 1024, 256, 65536), (58, 12, 1024, 128, 65536)]:
    for kind in ("random", "identity"):
        e = PPOEngine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=B, n_epochs=4, pi=(64, 64), vf=(64, 64), ent_coef=0.01)
        e.set_params(orthogonal_policy_init(D, A, (64, 64), (64, 64), 0))
        e.collect_synthetic()
        rng = np.random.default_rng(0)
        perms = np.stack([rng.permutation(N * T) if kind == "random" else np.arange(N * T) for _ in range(4)])
        e.train(perms)
        e.profile(True)
        e.train(perms)
        pr = e.profile_read()
        print("%-8s %2d/%-2d: train %.1f us/launch, reduce %.1f, apply %.1f" % (kind, D, A, 1e3 * pr["train_grad"][0] / pr["train_grad"][1],
              1e3 * pr["grad_reduce"][0] / pr["grad_reduce"][1], 1e3 * pr["apply"][0] / pr["apply"][1]), flush=True)
        e.close()
