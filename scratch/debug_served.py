import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ppo_oracle as O
from mobrob_amd.engine import PPOEngine
from mobrob_amd.envs.native_env import NativeGoalVecEnv
from mobrob_amd.envs.wrapper import ROBOT_DIMS
robot, N, parts, T = sys.argv[1] if len(sys.argv) > 1 else "doggo", 192, 2, 37
D, A, _ = ROBOT_DIMS[robot]
p = O.init_params(D, A, (256, 256), (256, 256), seed=6)
keys = ("obs", "actions", "rewards", "episode_starts", "values", "log_probs", "advantages", "returns", "last_values")
out = {}
for mode in ("0", "2"):
    e = PPOEngine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=64, n_epochs=1, seed=11, pi=(256, 256), vf=(256, 256))
    e.set_params(p)
    env = NativeGoalVecEnv.for_robot(robot, N, time_limit=5, seed=7)
    b = dict(obs=e.pinned((N, D)), clip=e.pinned((N, A)), rew=e.pinned((N,)), done=e.pinned((N,), np.uint8),
             trunc=e.pinned((N,), np.uint8), term=e.pinned((N, D)))
    env.use_buffers(obs=b["obs"], rewards=b["rew"], dones=b["done"], truncated=b["trunc"], terminal_obs=b["term"])
    env.reset()
    os.environ["MOBROB_COLLECT_SERVER"] = mode
    os.environ["MOBROB_SERVER_TIMEOUT_S"] = "5"
    e.rollout_begin()
    e.part_pipeline(parts, b["obs"], b["clip"], b["rew"], b["done"], b["trunc"], b["term"]).collect(env.step_range_fn, env.handle)
    out[mode] = {k: e.read(k) for k in keys}
    env.close(); e.close()
for k in keys:
    a, c = out["0"][k], out["2"][k]
    d = np.argwhere(a != c)
    print(k, a.shape, "differing:", len(d), "first:", d[:6].tolist(), "slots:", sorted(set(d[:, 0].tolist()))[:10] if len(d) else [])
    if len(d):
        i = tuple(d[0]); print("   ", a[i], c[i])
