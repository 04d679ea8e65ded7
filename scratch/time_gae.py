import sys, numpy as np
sys.path.insert(0, '.')
from mobrob_amd.engine import PPOEngine
from tests.util import synthetic_rollout
from oracle import ppo_oracle as O
T, N = 1000, 4096
e = PPOEngine(obs_dim=4, act_dim=2, n_envs=N, n_steps=T, batch_size=4096, n_epochs=1, rollout_graph=False)
buf, lv, dones = synthetic_rollout(T, N, 4, 2, seed=1, p_done=0.01)
e.load_rollout(buf, lv, dones)
e.compute_gae()
e.profile(True)
for _ in range(10):
    e.compute_gae()
pr = e.profile_read()
print("gae us", 1e3 * pr["gae"][0] / pr["gae"][1], "GB/s", 20 * T * N / (pr["gae"][0] / pr["gae"][1] * 1e-3) / 1e9)
adv, ret = O.gae(buf["rewards"], buf["values"], buf["episode_starts"], lv, dones, 0.99, 0.95)
print("bit-exact", np.array_equal(e.read("advantages"), adv), np.array_equal(e.read("returns"), ret))
