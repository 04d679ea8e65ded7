"""Shared helpers for the parity tests (fixtures -> oracle structures)."""
import os
from collections import OrderedDict

import numpy as np

from oracle import ppo_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ENVS = ["point", "car", "doggo", "drone", "turtlebot3"]


def load_golden(env):
    return np.load(os.path.join(GOLDEN, f"{env}.npz"))


def golden_params(g, prefix="p/"):
    keys = O.param_keys()
    return OrderedDict((k, g[prefix + k].astype(np.float32).copy()) for k in keys)


def golden_adam(g, m="m/", v="v/", step_key="adam_step"):
    keys = O.param_keys()
    return O.AdamState(OrderedDict((k, g[m + k].copy()) for k in keys),
                       OrderedDict((k, g[v + k].copy()) for k in keys), int(g[step_key]))


def golden_hyper(g):
    return O.Hyper(gamma=float(g["hyper/gamma"]), gae_lambda=float(g["hyper/gae_lambda"]),
                   clip_range=float(g["hyper/clip_range"]), ent_coef=float(g["hyper/ent_coef"]),
                   vf_coef=float(g["hyper/vf_coef"]), max_grad_norm=float(g["hyper/max_grad_norm"]),
                   learning_rate=float(g["hyper/learning_rate"]), beta1=float(g["hyper/beta1"]),
                   beta2=float(g["hyper/beta2"]), adam_eps=float(g["hyper/adam_eps"]),
                   n_epochs=int(g["hyper/n_epochs"]), batch_size=int(g["hyper/batch_size"]))


def golden_minibatch(g):
    return (g["mb/obs"], g["mb/actions"], g["mb/old_values"], g["mb/old_log_prob"], g["mb/advantages"],
            g["mb/returns"])


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / (1e-6 + np.maximum(np.abs(a), np.abs(b))))) if a.size else 0.0


def scaled_err(a, b):
    """max |a-b| relative to the largest magnitude in the reference array (robust to cancellation)."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(1e-12, float(np.max(np.abs(b))))) if a.size else 0.0


def synthetic_rollout(T, N, D, A, seed=0, p_done=0.02):
    """A random rollout buffer with episode boundaries, for GAE / train parity."""
    rng = np.random.default_rng(seed)
    f = np.float32
    buf = dict(obs=rng.standard_normal((T, N, D)).astype(f), actions=rng.standard_normal((T, N, A)).astype(f),
               rewards=(0.03 + 0.5 * rng.standard_normal((T, N))).astype(f),
               episode_starts=(rng.random((T, N)) < p_done).astype(f),
               values=rng.standard_normal((T, N)).astype(f), log_probs=(-A + rng.standard_normal((T, N))).astype(f))
    last_values = rng.standard_normal(N).astype(f)
    dones = rng.random(N) < 0.3
    return buf, last_values, dones


def oracle_epoch_off_clip_boundaries(p, st, buf, h, perm, band=2e-5, max_moved=64, max_passes=6, log=None):
    """One epoch of PPO.train on the oracle (float32 BLAS: SB3-CPU's arithmetic), with every row whose ratio comes within
    `band` of a clip boundary 1 +- clip_range at the optimizer step that consumes it MOVED off the boundary first.

    Why: the clipped surrogate's gradient is discontinuous in the ratio at the boundaries and one row is ~1/sqrt(B) of a
    minibatch gradient, so a row within float32 rounding of a boundary is inside the clip range for one correct
    implementation and outside for another; over an epoch such a row shows as a 1e-4-level parameter difference that says
    nothing about either implementation (tests/test_full_size_gpu.py::_check_grad).  Every row is consumed by
    exactly one step of the epoch, so changing its stored log-prob (by -0.01: the ratio grows by 1 %) leaves all earlier
    steps as they were: the oracle is re-run from the earliest step that had a flagged row until no step has one.

    p, st, buf are updated IN PLACE (p / st to the end of the epoch, buf["log_probs"] with the moved rows).
    -> (per-step stats list, flat indices of the moved rows, passes)."""
    T = buf["rewards"].shape[0]
    total = perm.shape[0]
    B = h.batch_size
    nmb = -(-total // B)
    lo, hi = np.float32(1.0 - h.clip_range), np.float32(1.0 + h.clip_range)
    snaps = {}                      # step -> (params, Adam state) BEFORE that step
    stats_rows = [None] * nmb
    moved = []
    start, passes = 0, 0
    while True:
        passes += 1
        assert passes <= max_passes, f"{passes} passes and still rows within {band} of a clip boundary"
        if start in snaps:
            ps, ss = snaps[start]
            for k in p:
                p[k] = ps[k].copy()
                st.exp_avg[k], st.exp_avg_sq[k] = ss[0][k].copy(), ss[1][k].copy()
            st.step = ss[2]
        flagged = []
        for s in range(start, nmb):
            if s not in snaps:
                snaps[s] = ({k: v.copy() for k, v in p.items()},
                            ({k: v.copy() for k, v in st.exp_avg.items()}, {k: v.copy() for k, v in st.exp_avg_sq.items()}, st.step))
            idx = perm[s * B:(s + 1) * B]
            stats, grads, aux = O.loss_and_grads(p, *O.gather_minibatch(buf, idx), h)
            near = (np.abs(aux["ratio"] - lo) < band) | (np.abs(aux["ratio"] - hi) < band)
            if near.any():
                flagged.append((s, idx[near]))
            clipped, total_norm = O.clip_grad_norm(grads, h.max_grad_norm)
            stats["grad_norm"] = total_norm
            O.adam_step(p, clipped, st, h.learning_rate, h.beta1, h.beta2, h.adam_eps)
            stats_rows[s] = stats
        if not flagged:
            break
        rows = np.concatenate([r for _, r in flagged])
        moved.extend(int(r) for r in rows)
        assert len(moved) <= max_moved, f"{len(moved)} rows within {band} of a clip boundary -- not a rounding artefact"
        t, n = O.flat_to_tn(rows, T)
        buf["log_probs"][t, n] -= np.float32(0.01)
        start = flagged[0][0]
        for s in [k for k in snaps if k > start]:
            del snaps[s]
        if log:
            log(f"pass {passes}: {len(rows)} row(s) within {band:g} of a clip boundary in steps {[s for s, _ in flagged]}; "
                f"moved, re-running from step {start}")
    return stats_rows, moved, passes
