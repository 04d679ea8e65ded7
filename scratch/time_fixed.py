"""fixed vs per-tile cost of k_fused_train: time launches for 1, 2, 4, 8 tiles per workgroup"""
import sys, numpy as np
sys.path.insert(0, '.')
import os
from mobrob_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from mobrob_amd.engine import PPOEngine
from mobrob_amd.rl_control.init import orthogonal_policy_init
D, A, H, N, T = 58, 12, 256, 4096, 64
res = []
for B in (8192, 16384, 65536):
    e = PPOEngine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=B, n_epochs=4, pi=(H, H), vf=(H, H), ent_coef=0.01)
    e.set_params(orthogonal_policy_init(D, A, (H, H), (H, H), 0))
    e.collect_synthetic()
    e.train(None)
    e.profile(True)
    e.train(None)
    pr = e.profile_read()
    e.profile(False)
    ms, calls = pr["train_grad"]
    rms, rcalls = pr["grad_reduce"]
    ams, acalls = pr["apply"]
    res.append((B, ms / calls * 1e3, rms / rcalls * 1e3, ams / acalls * 1e3))
    print(f"B={B:7d} tiles/WG={B//64//128:2d} train {ms/calls*1e3:8.1f} us  reduce {rms/rcalls*1e3:6.1f} us  apply {ams/acalls*1e3:6.1f} us", flush=True)
    e.close()
x = np.array([r[0] / 64 / 128 for r in res]); y = np.array([r[1] for r in res])
k, c = np.polyfit(x, y, 1)
print(f"fit: {k:.1f} us per tile + {c:.1f} us fixed")
