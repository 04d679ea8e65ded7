import sys, numpy as np
sys.path.insert(0, "/root/repo")
sys.path.insert(0, ".")
from mobrob_amd.engine import PPOEngine
from oracle import ppo_oracle as O
from tests.test_engine_gpu import _consistent_rollout
bad = 0
for H in (64, 256):
    for (D, A, N, T, B, E) in [(1, 1, 1, 3, 2, 2), (64, 32, 5, 7, 16, 1), (65, 3, 4, 6, 8, 1), (3, 1, 33, 2, 64, 2), (16, 16, 64, 40, 2500, 1), (58, 12, 3, 1, 3, 1)]:
        rng = np.random.default_rng(1)
        p0 = O.init_params(D, A, (H, H), (H, H), seed=2)
        buf, lv, dones = _consistent_rollout(p0, T, N, D, A, seed=3)
        h = O.Hyper(n_epochs=E, batch_size=B, ent_coef=0.01)
        buf["advantages"], buf["returns"] = O.gae(buf["rewards"], buf["values"], buf["episode_starts"], lv, dones, h.gamma, h.gae_lambda)
        perms = np.stack([rng.permutation(T * N) for _ in range(E)])
        e = PPOEngine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=B, n_epochs=E, pi=(H, H), vf=(H, H), ent_coef=0.01)
        e.set_params(p0)
        e.load_rollout(buf, lv, dones)
        e.train(perms)
        got = e.get_params()
        p = {k: v.copy() for k, v in p0.items()}
        O.train(p, O.AdamState.zeros_like(p), buf, h, perms)
        err = max(float(np.max(np.abs(got[k] - p[k]))) for k in p)
        # device rollout on the same engine (synthetic env), twice
        e.collect_synthetic(p_term=0.1, time_limit=5); e.collect_synthetic(p_term=0.1, time_limit=5)
        fin = np.isfinite(e.read("advantages")).all() and np.isfinite(e.get_flat_params()).all()
        print(f"H={H} D={D} A={A} N={N} T={T} B={B}: max|dparam - oracle| {err:.2e} finite {fin}")
        bad += (err > 1e-4) or (not fin)
        e.close()
print("FAILED" if bad else "all edge shapes ok")
