import sys, numpy as np
sys.path.insert(0, '.')
from mobrob_amd.engine import PPOEngine
from mobrob_amd.rl_control.init import orthogonal_policy_init
for (D, A, N, T, B) in [(14, 2, 1024, 512, 16384), (14, 2, 1024, 512, 65536), (14, 2, 1024, 512, 262144), (58, 12, 1024, 512, 65536), (58, 12, 1024, 512, 262144)]:
    e = PPOEngine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=B, n_epochs=2, pi=(64, 64), vf=(64, 64), ent_coef=0.01)
    e.set_params(orthogonal_policy_init(D, A, (64, 64), (64, 64), 0))
    e.collect_synthetic()
    e.train(None)
    e.profile(True)
    e.train(None)
    pr = e.profile_read()
    print(D, A, "B", B, "train us/launch", 1e3 * pr["train_grad"][0] / pr["train_grad"][1], "reduce", 1e3 * pr["grad_reduce"][0] / pr["grad_reduce"][1], "apply", 1e3 * pr["apply"][0] / pr["apply"][1], "launches", pr["train_grad"][1])
    e.close()
