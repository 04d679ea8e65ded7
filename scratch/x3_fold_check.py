"""Do the x3 packs kept current by k_adam_pack equal the packs rebuilt from the parameters?  (rollout after train, with and
without a set_params in between: the synthetic rollout's actions depend on the packs only)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ppo_oracle as O
from mobrob_amd.engine import PPOEngine
D, A, H, N, T, B = 58, 12, 256, 64, 32, 512
outs = []
for rebuild in (False, True):
    e = PPOEngine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=B, n_epochs=2, pi=(H, H), vf=(H, H), seed=3)
    e.set_params(O.init_params(D, A, (H, H), (H, H), seed=1))
    e.collect_synthetic(p_term=0.05, time_limit=9)
    e.train(None)
    if rebuild:
        e.set_flat_params(e.get_flat_params())
    e.collect_synthetic(p_term=0.05, time_limit=9)
    e.synchronize()
    outs.append({k: e.read(k) for k in ("actions", "values", "log_probs")})
    e.train(None)
    outs[-1]["params"] = e.get_flat_params()
    e.close()
for k in outs[0]:
    d = np.abs(outs[0][k].astype(np.float64) - outs[1][k]).max()
    print(k, "max |diff| fold vs rebuild:", d)
