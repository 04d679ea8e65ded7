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


def policy_init(obs_dim, act_dim, pi=(64, 64), vf=(64, 64), seed=0, log_std_init=0.0, ortho_init=True):
    """ActorCriticPolicy._build: `ortho_init=True` (SB3's default) -> orthogonal_policy_init; False -> torch's default
    nn.Linear initialisation of every layer.  `log_std_init` fills the state-independent log standard deviation."""
    if ortho_init:
        return orthogonal_policy_init(obs_dim, act_dim, pi, vf, seed, log_std_init)
    rng = np.random.default_rng(seed)
    p = OrderedDict()
    p["log_std"] = np.full((act_dim,), log_std_init, np.float32)
    for name, rows, cols in (("mlp_extractor.policy_net.0", pi[0], obs_dim), ("mlp_extractor.policy_net.2", pi[1], pi[0]),
                             ("mlp_extractor.value_net.0", vf[0], obs_dim), ("mlp_extractor.value_net.2", vf[1], vf[0]),
                             ("action_net", act_dim, pi[1]), ("value_net", 1, vf[1])):
        p[name + ".weight"], p[name + ".bias"] = default_linear_init(rng, rows, cols)
    return p


def orthogonal_policy_init(obs_dim, act_dim, pi=(64, 64), vf=(64, 64), seed=0, log_std_init=0.0):
    rng = np.random.default_rng(seed)
    p = OrderedDict()
    p["log_std"] = np.full((act_dim,), log_std_init, np.float32)
    g = math.sqrt(2.0)
    p["mlp_extractor.policy_net.0.weight"] = _orthogonal(rng, pi[0], obs_dim, g)
    p["mlp_extractor.policy_net.0.bias"] = np.zeros(pi[0], np.float32)
    p["mlp_extractor.policy_net.2.weight"] = _orthogonal(rng, pi[1], pi[0], g)
    p["mlp_extractor.policy_net.2.bias"] = np.zeros(pi[1], np.float32)
    p["mlp_extractor.value_net.0.weight"] = _orthogonal(rng, vf[0], obs_dim, g)
    p["mlp_extractor.value_net.0.bias"] = np.zeros(vf[0], np.float32)
    p["mlp_extractor.value_net.2.weight"] = _orthogonal(rng, vf[1], vf[0], g)
    p["mlp_extractor.value_net.2.bias"] = np.zeros(vf[1], np.float32)
    p["action_net.weight"] = _orthogonal(rng, act_dim, pi[1], 0.01)
    p["action_net.bias"] = np.zeros(act_dim, np.float32)
    p["value_net.weight"] = _orthogonal(rng, 1, vf[1], 1.0)
    p["value_net.bias"] = np.zeros(1, np.float32)
    return p
