"""Margins of the full-size epoch tests (tests/test_full_size_gpu.py): one free-running epoch of the bench workloads through
`mobrob_ppo_train` on both matrix pipes against the float32 oracle, three seeds, with SB3's clip range 0.2 and with the clip
range opened (1e9: no gradient discontinuity) -- and, for the same inputs, the ORACLE AGAINST ITSELF (float32 BLAS vs float64
accumulation), which shows that the 1e-4-level deviations at clip 0.2 are a property of PPO's clipped surrogate, not of an
implementation.  Also counted: rows within 2e-5 of a clip boundary at the step that consumes them.
    python scratch/epoch_margin.py > profiles/r4/epoch_margin.txt"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ppo_oracle as O
from tests.test_full_size_gpu import SHAPES, _bench_like_engine, _device_perm_key, run_epoch_free

def oracle_vs_oracle(shape, seed, rs, clip):
    D, A, H, n_envs, T = (shape[k] for k in "DAHNT")
    B = 65536
    e, p, st, buf, h = _bench_like_engine(D, A, H, n_envs, T, B, seed, np.random.default_rng(rs), clip_range=clip)
    e.close()
    perm = O.feistel_permutation(T * n_envs, _device_perm_key(seed, 0))
    out, near_rows = [], 0
    for acc in (None, np.float64):
        q = type(p)((k, v.copy()) for k, v in p.items())
        s2 = O.AdamState(type(p)((k, v.copy()) for k, v in st.exp_avg.items()), type(p)((k, v.copy()) for k, v in st.exp_avg_sq.items()), st.step)
        for mb in range(-(-T * n_envs // B)):
            idx = perm[mb * B:(mb + 1) * B]
            _, g, aux = O.loss_and_grads(q, *O.gather_minibatch(buf, idx), h, acc=acc)
            if acc is None:
                near_rows += int(((np.abs(aux["ratio"] - (1 - h.clip_range)) < 2e-5) | (np.abs(aux["ratio"] - (1 + h.clip_range)) < 2e-5)).sum())
            g, _ = O.clip_grad_norm(g, h.max_grad_norm)
            O.adam_step(q, g, s2, h.learning_rate, h.beta1, h.beta2, h.adam_eps)
        out.append(q)
    return max(float(np.max(np.abs(out[0][k] - out[1][k]))) for k in p), near_rows

for clip in (0.2, 1e9):
    for shape in SHAPES:
        for seed, rs in ((23, 6), (24, 7), (25, 8)):
            t0 = time.time()
            errs, _, _ = run_epoch_free(shape, seed, rs, clip_range=clip)
            line = "  ".join(f"engine[{pipe}] vs oracle: worst {max(e, key=e.get).replace('mlp_extractor.', '')} {max(e.values()):.2e}" for pipe, e in errs.items())
            oo, near = oracle_vs_oracle(shape, seed, rs, clip) if seed == 23 else (float("nan"), -1)
            print(f"clip {clip:g}  {shape['name']}  seed {seed}  {line}  oracle f32 vs oracle f64-acc: {oo:.2e}  rows within 2e-5 of a boundary: {near}  ({time.time() - t0:.0f} s)", flush=True)
