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


# ---- tests/golden/arch_cases.npz (make_arch_fixture.py): other activations and depths through torch's own modules ----
_ARCH = None


def arch_cases():
    """-> list of case names"""
    global _ARCH
    if _ARCH is None:
        _ARCH = np.load(os.path.join(GOLDEN, "arch_cases.npz"))
    return [str(c) for c in _ARCH["cases"]]


ARCH_CASES = ["act_tanh", "act_relu", "act_elu", "act_leakyrelu", "act_sigmoid", "act_softplus", "act_softsign", "act_hardtanh",
              "act_relu6", "depth4_tanh", "depth5_3_elu", "depth8_relu", "depth1_6_tanh", "act_silu", "act_gelu", "act_mish"]


def arch_case(name):
    """-> (case dict of arrays without the name prefix, activation, pi widths, vf widths, params, Hyper)"""
    arch_cases()
    pre = name + "/"
    c = {k[len(pre):]: _ARCH[k] for k in _ARCH.files if k.startswith(pre)}
    act, pi, vf = str(c["activation"]), tuple(int(w) for w in c["pi"]), tuple(int(w) for w in c["vf"])
    keys = O.param_keys(len(pi), len(vf))
    p = OrderedDict((k, c["p/" + k].astype(np.float32).copy()) for k in keys)
    lr, clip, ent, vfc, mgn, eps = (float(x) for x in _ARCH["hyper"])
    h = O.Hyper(clip_range=clip, ent_coef=ent, vf_coef=vfc, max_grad_norm=mgn, learning_rate=lr, adam_eps=eps, activation=act,
                batch_size=100, n_epochs=1)
    return c, act, pi, vf, p, h


# ---- tests/golden/sde_cases.npz (make_sde_fixture.py): PPO(use_sde=True) through torch's own ops ----
SDE_CASES = ["sde_tanh_2x2", "sde_relu_1_3", "sde_elu_4", "sde_expln", "sde_shared_std", "sde_shared_expln"]
_SDE = None


def sde_case(name):
    """-> (case dict, activation, pi, vf, params, Hyper(use_sde=True, sde_use_expln=...)); c["full_std"] / c["use_expln"]: the options"""
    global _SDE
    if _SDE is None:
        _SDE = np.load(os.path.join(GOLDEN, "sde_cases.npz"))
    assert [str(c) for c in _SDE["cases"]] == SDE_CASES
    pre = name + "/"
    c = {k[len(pre):]: _SDE[k] for k in _SDE.files if k.startswith(pre)}
    act, pi, vf = str(c["activation"]), tuple(int(w) for w in c["pi"]), tuple(int(w) for w in c["vf"])
    p = OrderedDict((k, c["p/" + k].astype(np.float32).copy()) for k in O.param_keys(len(pi), len(vf)))
    lr, clip, ent, vfc, mgn, eps = (float(x) for x in _SDE["hyper"])
    h = O.Hyper(clip_range=clip, ent_coef=ent, vf_coef=vfc, max_grad_norm=mgn, learning_rate=lr, adam_eps=eps, activation=act,
                batch_size=100, n_epochs=1, use_sde=True, sde_use_expln=bool(c["use_expln"]))
    return c, act, pi, vf, p, h
