import sys, numpy as np
sys.path.insert(0, '.')
from mobrob_amd.engine import PPOEngine
from mobrob_amd.envs.vec_env import DeviceGoalVecEnv
from oracle import ppo_oracle as O
H, D, A, N, T, TL = 256, 58, 12, 200, 24, 7
p = O.init_params(D, A, (H, H), (H, H), seed=8)
res = {}
for persistent in (True, False):
    e = PPOEngine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=256, n_epochs=1, pi=(H, H), vf=(H, H), seed=21, rollout_persistent=persistent)
    e.set_params(p)
    DeviceGoalVecEnv(N, D, A, 2, time_limit=TL).collect(e)
    e.synchronize()
    res[persistent] = {k: e.read(k) for k in ("obs", "actions", "rewards", "episode_starts")}
    e.close()
a, b = res[True], res[False]
for k in a:
    d = np.argwhere(a[k] != b[k])
    print(k, "mismatches", len(d), "first", d[:5].tolist())
    if len(d):
        i = tuple(d[0])
        print("   ", a[k][i], b[k][i])
d = np.argwhere(a["obs"] != b["obs"])
ts = np.unique(d[:, 0]); print("obs mismatch steps", ts[:10], "features", np.unique(d[:, 2])[:20])
print("es at mismatching (t,n):", [(int(t), int(n), b["episode_starts"][min(t, T-1), n]) for t, n, f in d[:8]])
