import sys, numpy as np
sys.path.insert(0, '.')
from mobrob_amd.engine import PPOEngine
from mobrob_amd.envs.vec_env import DeviceGoalVecEnv
from oracle import ppo_oracle as O
H, D, A, N, T, TL = 256, 58, 12, 200, 8, 7
p = O.init_params(D, A, (H, H), (H, H), seed=8)
res = {}
for persistent in (True, False):
    e = PPOEngine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=256, n_epochs=1, pi=(H, H), vf=(H, H), seed=21, rollout_persistent=persistent)
    e.set_params(p)
    DeviceGoalVecEnv(N, D, A, 2, time_limit=TL).collect(e)
    e.synchronize()
    res[persistent] = {k: e.read(k) for k in ("obs", "env_state", "actions")}
    e.close()
a, b = res[True], res[False]
for t in range(T + 1):
    print(t, "obs mism", int((a["obs"][t] != b["obs"][t]).sum()), "act mism", int((a["actions"][min(t, T-1)] != b["actions"][min(t,T-1)]).sum()))
d = np.argwhere(a["env_state"] != b["env_state"]); print("state mism", len(d), d[:10].tolist())
n = 2
print(a["env_state"][n], b["env_state"][n]); print(a["obs"][7][n][:8], b["obs"][7][n][:8])
st = b["env_state"][n].astype(np.float32)
f = np.float32
for n in (2, 15, 17):
    pos = b["obs"][7][n][4:6].astype(f); goal = b["env_state"][n][6:8].astype(f)
    rel = (goal - pos).astype(f)
    s = f(np.float64(rel[0]) * np.float64(rel[0]))
    s = f(np.float64(rel[1]) * np.float64(rel[1]) + np.float64(s))
    d = f(np.sqrt(np.float64(s))); d1 = f(np.float64(d) + np.float64(f(1e-6)))
    exp = [f(np.float64(r) / np.float64(d1)) for r in rel]
    print(n, "expected", exp, "persistent", a["obs"][7][n][:2], "per-step", b["obs"][7][n][:2], "goal same", np.array_equal(a["env_state"][n][6:8], goal))
