import sys, os, json
sys.path.insert(0, '.')
from mobrob_amd import _lib
if len(sys.argv) > 1: _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from mobrob_amd.engine import PPOEngine
from mobrob_amd.rl_control.init import orthogonal_policy_init
D, A, H, N, T, B = 14, 2, 64, 1024, 256, 65536
e = PPOEngine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=B, n_epochs=4, pi=(H, H), vf=(H, H))
e.set_params(orthogonal_policy_init(D, A, (H, H), (H, H), 0))
e.collect_synthetic(); e.train(None)
e.profile(True); e.train(None)
pr = e.profile_read()
print({k: round(v[0] / max(v[1], 1) * 1e3, 1) for k, v in pr.items() if v[1]})
