"""Policy initialisation: orthogonal weights with SB3's gains (sqrt(2) hidden layers, 0.01 action head,
1.0 value head), zero biases, log_std = log_std_init (SURVEY.md Appendix A.2;
reached in the reference via PPO(...)._setup_model, src/mobrob/rl_control/ppo.py:50-59).

torch's CPU RNG stream cannot be reproduced outside torch, so a given `seed` yields a *different* (equally
distributed) initial policy than SB3 would; parity tests always supply weights."""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np


def _orthogonal(rng, rows, cols, gain):
    a = rng.standard_normal((max(rows, cols), min(rows, cols)))
    q, r = np.linalg.qr(a)
    q = q * np.sign(np.diag(r))
    if rows < cols:
        q = q.T
    return (gain * q[:rows, :cols]).astype(np.float32)


def default_linear_init(rng, rows, cols):
    """torch.nn.Linear.reset_parameters (what SB3 leaves in place with `ortho_init=False`): kaiming_uniform_(a=sqrt(5)) on
    the weight and U(-1/sqrt(fan_in), 1/sqrt(fan_in)) on the bias -- both U(-b, b) with b = 1/sqrt(fan_in)."""
    b = 1.0 / math.sqrt(cols)
    return (rng.uniform(-b, b, (rows, cols)).astype(np.float32), rng.uniform(-b, b, rows).astype(np.float32))


def policy_init(obs_dim, act_dim, pi=(64, 64), vf=(64, 64), seed=0, log_std_init=0.0, ortho_init=True, use_sde=False, full_std=True):
    """ActorCriticPolicy._build: `ortho_init=True` (SB3's default) -> orthogonal_policy_init; False -> torch's default
    nn.Linear initialisation of every layer.  `log_std_init` fills the state-independent log standard deviation."""
    if ortho_init:
        p = orthogonal_policy_init(obs_dim, act_dim, pi, vf, seed, log_std_init)
        if use_sde:   # StateDependentNoiseDistribution.proba_distribution_net: ones(latent_sde_dim, action_dim or 1) * log_std_init
            p["log_std"] = np.full((pi[-1], act_dim if full_std else 1), log_std_init, np.float32)
        return p
    rng = np.random.default_rng(seed)
    p = OrderedDict()
    p["log_std"] = np.full((pi[-1], act_dim if full_std else 1) if use_sde else (act_dim,), log_std_init, np.float32)
    layers = []
    for net, widths in (("policy_net", pi), ("value_net", vf)):   # one to three hidden layers per network, SB3's registration order
        prev = obs_dim
        for i, w in enumerate(widths):
            layers.append((f"mlp_extractor.{net}.{2 * i}", w, prev))
            prev = w
    layers += [("action_net", act_dim, pi[-1]), ("value_net", 1, vf[-1])]
    for name, rows, cols in layers:
        p[name + ".weight"], p[name + ".bias"] = default_linear_init(rng, rows, cols)
    return p


def orthogonal_policy_init(obs_dim, act_dim, pi=(64, 64), vf=(64, 64), seed=0, log_std_init=0.0):
    rng = np.random.default_rng(seed)
    p = OrderedDict()
    p["log_std"] = np.full((act_dim,), log_std_init, np.float32)
    g = math.sqrt(2.0)
    for net, widths in (("policy_net", pi), ("value_net", vf)):   # (two hidden layers draw from the generator in the order they always did)
        prev = obs_dim
        for i, w in enumerate(widths):
            p[f"mlp_extractor.{net}.{2 * i}.weight"] = _orthogonal(rng, w, prev, g)
            p[f"mlp_extractor.{net}.{2 * i}.bias"] = np.zeros(w, np.float32)
            prev = w
    p["action_net.weight"] = _orthogonal(rng, act_dim, pi[-1], 0.01)
    p["action_net.bias"] = np.zeros(act_dim, np.float32)
    p["value_net.weight"] = _orthogonal(rng, 1, vf[-1], 1.0)
    p["value_net.bias"] = np.zeros(1, np.float32)
    return p
