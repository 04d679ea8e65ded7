"""Repeat the update-vs-oracle comparison of tests/test_fuzz_gpu.py for every case many times in ONE process and report the
spread of the error: a run-to-run outlier means a race or an uninitialised read, not a tolerance problem."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ppo_oracle as O
from tests.test_fuzz_gpu import CASES
from tests.util import synthetic_rollout
from mobrob_amd.engine import PPOEngine
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
worst = {}
for rep in range(reps):
    for ci, (H, D, A, N, T, B, E, normalize, ent) in enumerate(CASES):
        rng = np.random.default_rng(H * 1000 + D * 31 + A)
        p = O.init_params(D, A, (H, H), (H, H), seed=D + A)
        p["log_std"] = rng.normal(-0.2, 0.2, A).astype(np.float32)
        buf, lv, dones = synthetic_rollout(T, N, D, A, seed=N + T)
        mean, val = O.policy_outputs(p, buf["obs"].reshape(T * N, D))
        buf["log_probs"] = (O.gaussian_log_prob(mean, p["log_std"], buf["actions"].reshape(T * N, A)) + rng.normal(0, 0.05, T * N)).astype(np.float32).reshape(T, N)
        buf["values"] = val.reshape(T, N)
        h = O.Hyper(n_epochs=E, batch_size=B, ent_coef=ent, normalize_advantage=normalize)
        e = PPOEngine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=B, n_epochs=E, pi=(H, H), vf=(H, H), ent_coef=ent, normalize_advantage=normalize)
        e.set_params(p); e.load_rollout(buf, lv, dones); e.compute_gae()
        adv, ret = O.gae(buf["rewards"], buf["values"], buf["episode_starts"], lv, dones, h.gamma, h.gae_lambda)
        buf["advantages"], buf["returns"] = adv, ret
        perms = np.stack([rng.permutation(T * N) for _ in range(E)])
        e.epoch_begin(perms[0]); e.synchronize()
        ast = e.read("advstat")
        e.train(perms)
        got = e.get_flat_params()
        if rep == 0:
            q = {k: v.copy() for k, v in p.items()}
            O.train(q, O.AdamState.zeros_like(q), buf, h, perms)
            worst[ci] = dict(ref=O.flatten_params(q), first=got, ast=ast, errs=[])
        w = worst[ci]
        single = normalize and (min(B, T * N) == 1 or (T * N) % B == 1)
        err = float(np.max(np.abs(got - w["ref"]))) if not single else 0.0
        w["errs"].append(err)
        if not np.array_equal(ast, w["ast"]):
            print(f"case {ci} {CASES[ci]} rep {rep}: ADVSTAT differs run to run", np.abs(ast - w['ast']).max(), flush=True)
        e.close()
for ci, w in worst.items():
    errs = np.array(w["errs"])
    flag = "  <-- OUTLIER" if errs.max() > 3 * max(np.median(errs), 1e-6) or errs.max() > 3e-4 else ""
    print(f"case {ci:2d} {CASES[ci]}: err min {errs.min():.2e} median {np.median(errs):.2e} max {errs.max():.2e}{flag}")
