"""The oracle-vs-oracle leg of scratch/epoch_margin.py for three seeds per shape, WITHOUT a GPU: one free-running
epoch of the bench workloads (63 / 32 optimizer steps of 65 536 rows, SB3's clip range 0.2 and the clip range opened) walked by the
float32 oracle (SB3-CPU's arithmetic: float32 BLAS) and by the same oracle accumulating every contraction in float64.  The two are
both correct implementations of the same update; how far their parameters are apart after the epoch is the bound a free-running
comparison can be held to (tests/test_full_size_gpu.py::test_benchmarked_epoch_matches_oracle: 5e-4).

The rollout is produced on the CPU the way tests/test_full_size_gpu.py::_bench_like_engine leaves it on the device: synthetic
observations ~ N(0, 1), actions sampled from the policy, stored log-probs perturbed by N(0, 0.12) so that the ratios straddle the
clip range, rewards 0.03 + 0.1 z (+5 at a termination, p = 0.01), GAE by the oracle, a non-trivial Adam state.
    python scratch/epoch_margin_oracle.py > profiles/r5/epoch_margin_oracle.txt"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ppo_oracle as O  # noqa: E402

SHAPES = [dict(name="doggo-4096env-2x256", D=58, A=12, H=256, N=4096, T=1000),
          dict(name="point-1024env-2x64", D=14, A=2, H=64, N=1024, T=2048)]
GOLDEN_RATIO = 0x9E3779B97F4A7C15
F32 = np.float32


def rollout(shape, seed, rng, clip):
    D, A, H, N, T = (shape[k] for k in "DAHNT")
    h = O.Hyper(gamma=0.99, gae_lambda=0.95, ent_coef=0.01, n_epochs=1, batch_size=65536, learning_rate=3e-4, clip_range=clip)
    p = O.init_params(D, A, (H, H), (H, H), seed=seed)
    p["log_std"] = rng.normal(-0.3, 0.2, A).astype(F32)
    p["action_net.weight"] *= 30
    for k in p:
        if k.endswith("bias"):
            p[k] = rng.normal(0, 0.05, p[k].shape).astype(F32)
    st = O.AdamState(type(p)((k, rng.normal(0, 1e-3, v.shape).astype(F32)) for k, v in p.items()),
                     type(p)((k, (1e-6 * (0.5 + 1.5 * rng.random(v.shape))).astype(F32)) for k, v in p.items()), 1000)
    obs = rng.standard_normal((T + 1, N, D), dtype=F32)
    mean = np.empty((T, N, A), F32)
    val = np.empty((T + 1, N), F32)
    for t0 in range(0, T + 1, 64):                       # forward in slabs of 64 steps
        m, v = O.policy_outputs(p, obs[t0:t0 + 64].reshape(-1, D))
        n = min(64, T + 1 - t0)
        val[t0:t0 + n] = v.reshape(n, N)
        if t0 < T:
            k = min(64, T - t0)
            mean[t0:t0 + k] = m.reshape(n, N, A)[:k]
    eps = rng.standard_normal((T, N, A), dtype=F32)
    actions = (mean + np.exp(p["log_std"]) * eps).astype(F32)
    logp = O.gaussian_log_prob(mean.reshape(-1, A), p["log_std"], actions.reshape(-1, A)).reshape(T, N)
    logp = (logp + rng.normal(0, 0.12, logp.shape).astype(F32)).astype(F32)
    term = rng.random((T, N)) < 0.01
    rewards = (0.03 + 0.1 * rng.standard_normal((T, N), dtype=F32) + 5.0 * term).astype(F32)
    es = np.zeros((T, N), F32)
    es[1:] = term[:-1]
    adv, ret = O.gae(rewards, val[:T], es, val[T], term[-1], h.gamma, h.gae_lambda)
    buf = dict(obs=obs[:T], actions=actions, rewards=rewards, episode_starts=es, values=val[:T].copy(), log_probs=logp,
               advantages=adv, returns=ret)
    return p, st, buf, h


def epoch(p, st, buf, h, perm, acc):
    q = type(p)((k, v.copy()) for k, v in p.items())
    s2 = O.AdamState(type(p)((k, v.copy()) for k, v in st.exp_avg.items()), type(p)((k, v.copy()) for k, v in st.exp_avg_sq.items()), st.step)
    total = len(perm)
    near = 0
    for s in range(0, total, h.batch_size):
        idx = perm[s:s + h.batch_size]
        _, g, aux = O.loss_and_grads(q, *O.gather_minibatch(buf, idx), h, acc=acc)
        near += int(((np.abs(aux["ratio"] - (1 - h.clip_range)) < 2e-5) | (np.abs(aux["ratio"] - (1 + h.clip_range)) < 2e-5)).sum())
        g, _ = O.clip_grad_norm(g, h.max_grad_norm)
        O.adam_step(q, g, s2, h.learning_rate, h.beta1, h.beta2, h.adam_eps)
    return q, near


def main():
    only = sys.argv[1] if len(sys.argv) > 1 else None
    for shape in SHAPES:
        if only and only not in shape["name"]:
            continue
        for seed, rs in ((23, 6), (24, 7), (25, 8)):
            for clip in (0.2, 1e9):
                t0 = time.time()
                p, st, buf, h = rollout(shape, seed, np.random.default_rng(rs), clip)
                total = shape["T"] * shape["N"]
                perm = O.feistel_permutation(total, ((seed * GOLDEN_RATIO) & (2 ** 64 - 1)) ^ (1 << 48) ^ 1)
                a, near = epoch(p, st, buf, h, perm, None)
                b, _ = epoch(p, st, buf, h, perm, np.float64)
                worst = max(p, key=lambda k: float(np.max(np.abs(a[k] - b[k]))))
                d = float(np.max(np.abs(a[worst] - b[worst])))
                print(f"clip {clip:g}  {shape['name']}  seed {seed}  oracle f32 vs oracle f64-acc after {-(-total // h.batch_size)} free steps: "
                      f"worst {worst.replace('mlp_extractor.', '')} {d:.2e}  rows within 2e-5 of a clip boundary (f32 run): {near}  "
                      f"({time.time() - t0:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
