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
