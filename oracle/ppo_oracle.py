"""CPU oracle for the goal-conditioned PPO hot path (TEST INFRASTRUCTURE ONLY).

This file is a float32 NumPy restatement of the arithmetic the reference executes for its
PPO training path.  The reference (`/root/reference/src/mobrob/rl_control/ppo.py:50-59`,
`:73-74`) only *configures* the path; every FLOP runs inside the un-vendored third-party
dependency **stable-baselines3 == 2.0.0** (`/root/reference/requirements.txt:9`; the shipped
checkpoints record exactly `2.0.0` in `_stable_baselines3_version`) on PyTorch CPU
(`device: cpu`, `/root/reference/data/configs/doggo-ppo.yaml:24`).  SB3 is not installable
here (no network), so this oracle restates SB3 2.0.0's published algorithm (SURVEY.md
Appendix A) and is pinned by

  * the five reference checkpoints `data/policies/<env>-ppo.zip` (real trained weights, real
    Adam moments and step counters, real simulator observations `_last_obs`), and
  * golden vectors in `tests/golden/*.npz`, produced by `tests/golden/make_fixtures.py` (single steps per robot)
    and `tests/golden/make_loop_fixture.py` (a whole iteration: buffer GAE, flatten, 8 optimizer steps) with
    the *same third-party kernels SB3 calls* (torch.nn.Linear / Tanh, torch.distributions
    Normal, autograd, `clip_grad_norm_`, `torch.optim.Adam` loaded with the checkpoint's real
    optimizer state).

PARITY STATUS: the reference ships no tests for this path -> "parity
unpinned by reference tests"; it is pinned only by the artefacts and torch-op goldens above.
(Reference-HELD outputs exist for two things, neither of them this arithmetic: the `observation_space` /
`action_space` blobs inside the zips pin the checkpoint writer, tests/golden/reference_spaces.json; the counters
the five checkpoints were saved with -- num_timesteps, _n_updates, Adam's step, _current_progress_remaining -- pin the
learn loop's bookkeeping, `learn_loop_counters` below, tests/golden/reference_counters.json.)

`loss_and_grads(..., acc=np.float64)` accumulates every contraction in float64 (products of
float32 are exact there): the mode the full-size parity tests use, so that the checker's own
summation error over 65 536 rows stays ~1e-7 (tests/test_full_size_gpu.py).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module.  The product (`mobrob_amd`) never does: it fails loudly if the HIP library is missing.

Conventions (all arrays float32 unless stated):
  params: dict keyed like SB3's `policy.state_dict()` in registration order
      log_std, mlp_extractor.policy_net.{0,2}.{weight,bias}, mlp_extractor.value_net.{0,2}.{weight,bias},
      action_net.{weight,bias}, value_net.{weight,bias};  weights are row-major [out, in].
  rollout arrays are [T, N, ...]; the minibatch flatten is ENV-MAJOR (flat = n*T + t),
      SB3 `RolloutBuffer.swap_and_flatten` [SB3 common/buffers.py].
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass, field

import numpy as np

F32 = np.float32
LOG_SQRT_2PI = math.log(math.sqrt(2.0 * math.pi))  # torch.distributions.Normal.log_prob constant
HALF_LOG_2PI_PLUS_HALF = 0.5 + 0.5 * math.log(2.0 * math.pi)  # Normal.entropy constant


# --------------------------------------------------------------------------------------
# parameters
# --------------------------------------------------------------------------------------
def param_keys(n_hidden_pi: int = 2, n_hidden_vf: int = 2) -> list[str]:
    """SB3 `ActorCriticPolicy` parameter registration order (verified
    against `policy.pth` key order of every zip under /root/reference/data/policies)."""
    keys = ["log_std"]
    for i in range(n_hidden_pi):
        keys += [f"mlp_extractor.policy_net.{2 * i}.weight", f"mlp_extractor.policy_net.{2 * i}.bias"]
    for i in range(n_hidden_vf):
        keys += [f"mlp_extractor.value_net.{2 * i}.weight", f"mlp_extractor.value_net.{2 * i}.bias"]
    keys += ["action_net.weight", "action_net.bias", "value_net.weight", "value_net.bias"]
    return keys


def param_shapes(obs_dim: int, act_dim: int, pi=(64, 64), vf=(64, 64)) -> "OrderedDict[str, tuple]":
    shapes: OrderedDict[str, tuple] = OrderedDict()
    shapes["log_std"] = (act_dim,)
    last = obs_dim
    for i, h in enumerate(pi):
        shapes[f"mlp_extractor.policy_net.{2 * i}.weight"] = (h, last)
        shapes[f"mlp_extractor.policy_net.{2 * i}.bias"] = (h,)
        last = h
    last_pi = last
    last = obs_dim
    for i, h in enumerate(vf):
        shapes[f"mlp_extractor.value_net.{2 * i}.weight"] = (h, last)
        shapes[f"mlp_extractor.value_net.{2 * i}.bias"] = (h,)
        last = h
    shapes["action_net.weight"] = (act_dim, last_pi)
    shapes["action_net.bias"] = (act_dim,)
    shapes["value_net.weight"] = (1, last)
    shapes["value_net.bias"] = (1,)
    return shapes


def init_params(obs_dim, act_dim, pi=(64, 64), vf=(64, 64), seed=0, log_std_init=0.0):
    """Orthogonal init with SB3's gains (sqrt(2) hidden, 0.01 action head, 1.0 value head; zero
    biases; log_std = 0) -- Appendix A.2.  The *RNG stream* of torch's init is not reproducible
    outside torch, so parity tests always supply weights; this is for smoke/bench only."""
    rng = np.random.default_rng(seed)
    shapes = param_shapes(obs_dim, act_dim, pi, vf)
    p = OrderedDict()
    for k, shp in shapes.items():
        if k == "log_std":
            p[k] = np.full(shp, log_std_init, F32)
        elif k.endswith("bias"):
            p[k] = np.zeros(shp, F32)
        else:
            gain = math.sqrt(2.0)
            if k.startswith("action_net"):
                gain = 0.01
            elif k.startswith("value_net"):
                gain = 1.0
            rows, cols = shp
            a = rng.standard_normal((max(rows, cols), min(rows, cols)))
            q, r = np.linalg.qr(a)
            q = q * np.sign(np.diag(r))
            if rows < cols:
                q = q.T
            p[k] = (gain * q[:rows, :cols]).astype(F32)
    return p


def flatten_params(p) -> np.ndarray:
    return np.concatenate([np.asarray(v, F32).ravel() for v in p.values()])


def unflatten_params(flat, shapes) -> "OrderedDict[str, np.ndarray]":
    out, o = OrderedDict(), 0
    for k, shp in shapes.items():
        n = int(np.prod(shp))
        out[k] = np.asarray(flat[o:o + n], F32).reshape(shp).copy()
        o += n
    return out


def _net_layers(p, prefix):
    i, layers = 0, []
    while f"{prefix}.{2 * i}.weight" in p:
        layers.append((p[f"{prefix}.{2 * i}.weight"], p[f"{prefix}.{2 * i}.bias"]))
        i += 1
    return layers


# --------------------------------------------------------------------------------------
# forward pieces  [SB3 common/policies.py ActorCriticPolicy, common/torch_layers.py MlpExtractor]
# --------------------------------------------------------------------------------------
def _mm(a, b, acc=None):
    """a @ b with float32 operands and a float32 result.  acc=np.float64 accumulates the dot products in
    float64 (every f32*f32 product is exact there), which is what the full-size parity tests use so that the
    checker's own summation error over 65 536 rows stays far below the 1e-4 bar; acc=None is plain float32
    BLAS, torch-CPU's arithmetic."""
    if acc is None:
        return (a @ b).astype(F32)
    return (a.astype(acc) @ b.astype(acc)).astype(F32)


def _colsum(a, acc=None):
    return a.sum(axis=0, dtype=F32 if acc is None else acc).astype(F32)


def _linear(x, w, b, acc=None):
    # torch.nn.Linear: x @ W^T + b in float32
    return (_mm(x, w.T, acc) + b).astype(F32)


ACTIVATIONS = ("tanh", "relu", "elu", "leakyrelu", "sigmoid", "softplus", "softsign", "hardtanh", "relu6", "silu", "gelu", "mish")
NEEDS_PRE_ACTIVATION = ("silu", "gelu", "mish")   # not monotonic: the derivative is a function of z, not of f(z)


def _erf(x):
    """erf in float64 (Abramowitz-Stegun 7.1.26 is too coarse for 1e-5 parity: math.erf element-wise)."""
    return np.vectorize(math.erf, otypes=[np.float64])(np.asarray(x, np.float64))


def _activate(z, activation):
    """`activation_fn` of SB3's MlpExtractor (policy_kwargs; stable_baselines3/common/torch_layers.py `create_mlp` puts one module
    behind every Linear): nn.Tanh (default) or one of the torch.nn modules below with their default arguments
    [torch/nn/modules/activation.py: ReLU max(z, 0); ELU alpha 1: z > 0 ? z : exp(z) - 1; LeakyReLU negative_slope 0.01;
    Sigmoid; Softplus beta 1, threshold 20: z > 20 ? z : log(1 + exp(z)); Softsign z / (1 + |z|); Hardtanh clamp(z, -1, 1);
    ReLU6 clamp(z, 0, 6); SiLU z sigmoid(z); GELU approximate='none': z Phi(z) = 0.5 z (1 + erf(z / sqrt 2)); Mish z tanh(softplus(z))]."""
    z = np.asarray(z, F32)
    one = F32(1.0)
    if activation == "tanh":
        return np.tanh(z).astype(F32)
    if activation == "relu":
        return np.maximum(z, F32(0.0)).astype(F32)
    if activation == "elu":
        return np.where(z > 0, z, np.expm1(np.minimum(z, F32(0.0)))).astype(F32)
    if activation == "leakyrelu":
        return np.where(z > 0, z, F32(0.01) * z).astype(F32)
    if activation == "sigmoid":
        return (one / (one + np.exp(-z))).astype(F32)
    if activation == "softplus":
        return np.where(z > F32(20.0), z, np.log1p(np.exp(np.minimum(z, F32(20.0))))).astype(F32)
    if activation == "softsign":
        return (z / (one + np.abs(z))).astype(F32)
    if activation == "hardtanh":
        return np.clip(z, F32(-1.0), F32(1.0)).astype(F32)
    if activation == "relu6":
        return np.clip(z, F32(0.0), F32(6.0)).astype(F32)
    if activation == "silu":
        return (z / (one + np.exp(-z))).astype(F32)
    if activation == "gelu":
        return (0.5 * z.astype(np.float64) * (1.0 + _erf(z / math.sqrt(2.0)))).astype(F32)
    if activation == "mish":
        sp = np.where(z > F32(20.0), z, np.log1p(np.exp(np.minimum(z, F32(20.0))))).astype(F32)
        return (z * np.tanh(sp)).astype(F32)
    raise ValueError(f"activation {activation!r}")


def _activation_grad_pre(z, g, activation):
    """g * f'(z) for the non-monotonic activations, from the pre-activation z [torch's silu_backward / gelu_backward / mish_backward]:
    silu s (1 + z (1 - s)); gelu Phi(z) + z phi(z); mish t + z s (1 - t^2) with t = tanh(softplus(z)), s = sigmoid(z)."""
    z, g = np.asarray(z, F32), np.asarray(g, F32)
    one = F32(1.0)
    sg = (one / (one + np.exp(-z))).astype(F32)
    if activation == "silu":
        return (g * (sg * (one + z * (one - sg)))).astype(F32)
    if activation == "gelu":
        z64 = z.astype(np.float64)
        d = 0.5 * (1.0 + _erf(z64 / math.sqrt(2.0))) + z64 * np.exp(-0.5 * z64 * z64) / math.sqrt(2.0 * math.pi)
        return (g * d.astype(F32)).astype(F32)
    if activation == "mish":
        sp = np.where(z > F32(20.0), z, np.log1p(np.exp(np.minimum(z, F32(20.0))))).astype(F32)
        t = np.tanh(sp).astype(F32)
        return (g * (t + z * sg * (one - t * t))).astype(F32)
    raise ValueError(f"activation {activation!r}")


def _activation_grad(h, g, activation):
    """g * f'(z) with f'(z) written through the OUTPUT h = f(z) (all nine are monotonic) -- what torch's backward formulas give:
    tanh_backward g (1 - h^2); threshold_backward g [z > 0]; elu_backward g (z > 0 ? 1 : exp(z) = h + 1); leaky_relu_backward
    g (z > 0 ? 1 : 0.01); sigmoid_backward g h (1 - h); softplus_backward g sigmoid(z) = g (1 - exp(-h)); softsign: autograd of
    z / (1 + |z|) = g / (1 + |z|)^2 = g (1 - |h|)^2; hardtanh_backward g [-1 < z < 1]; relu6 = hardtanh(0, 6): g [0 < z < 6]."""
    h, g = np.asarray(h, F32), np.asarray(g, F32)
    one = F32(1.0)
    if activation == "tanh":
        return (g * (one - h * h)).astype(F32)
    if activation == "relu":
        return (g * (h > 0)).astype(F32)
    if activation == "elu":
        return np.where(h > 0, g, g * (h + one)).astype(F32)
    if activation == "leakyrelu":
        return np.where(h > 0, g, F32(0.01) * g).astype(F32)
    if activation == "sigmoid":
        return (g * (h * (one - h))).astype(F32)
    if activation == "softplus":
        return (g * (one - np.exp(-h))).astype(F32)
    if activation == "softsign":
        u = one - np.abs(h)
        return (g * (u * u)).astype(F32)
    if activation == "hardtanh":
        return (g * ((h > -1) & (h < 1))).astype(F32)
    if activation == "relu6":
        return (g * ((h > 0) & (h < 6))).astype(F32)
    raise ValueError(f"activation {activation!r}")


def mlp_latents(p, obs, acc=None, activation="tanh", pre=None):
    """Returns (per-layer activations of the policy net, of the value net); index 0 is the input.
    pre: optional dict that receives the PRE-activations per network prefix (lists, index l = layer l's z) -- what the backward of
    the non-monotonic activations needs."""
    x = np.asarray(obs).astype(F32)  # FlattenExtractor + obs.float()
    acts_pi, acts_vf = [x], [x]
    for prefix, acts in (("mlp_extractor.policy_net", acts_pi), ("mlp_extractor.value_net", acts_vf)):
        zs = []
        for w, b in _net_layers(p, prefix):
            zs.append(_linear(acts[-1], w, b, acc))
            acts.append(_activate(zs[-1], activation))
        if pre is not None:
            pre[prefix] = zs
    return acts_pi, acts_vf


def policy_outputs(p, obs, activation="tanh"):
    """mean actions [N,A] and values [N]."""
    acts_pi, acts_vf = mlp_latents(p, obs, activation=activation)
    mean = _linear(acts_pi[-1], p["action_net.weight"], p["action_net.bias"])
    value = _linear(acts_vf[-1], p["value_net.weight"], p["value_net.bias"])[:, 0]
    return mean, value


def gaussian_log_prob(mean, log_std, actions):
    """torch.distributions.Normal(mean, exp(log_std)).log_prob(actions).sum(1)
    [SB3 common/distributions.py DiagGaussianDistribution.log_prob].  Note torch evaluates
    var = scale**2 and log_scale = log(scale) from scale = exp(log_std), not from log_std."""
    std = np.exp(log_std.astype(F32)).astype(F32)
    var = (std * std).astype(F32)
    log_scale = np.log(std).astype(F32)
    d = (actions.astype(F32) - mean).astype(F32)
    lp = (-(d * d) / (F32(2.0) * var) - log_scale - F32(LOG_SQRT_2PI)).astype(F32)
    return lp.sum(axis=1, dtype=F32)


def gaussian_entropy(log_std, n):
    std = np.exp(log_std.astype(F32)).astype(F32)
    ent = (F32(HALF_LOG_2PI_PLUS_HALF) + np.log(std)).astype(F32)
    return np.full((n,), ent.sum(dtype=F32), F32)


# --------------------------------------------------------------------------------------
# gSDE  [SB3 common/distributions.py StateDependentNoiseDistribution with its defaults: full_std=True, use_expln=False,
#        squash_output=False, learn_features=False, epsilon=1e-6; reached through ActorCriticPolicy(use_sde=True)]
# --------------------------------------------------------------------------------------
SDE_EPSILON = 1e-6


def sde_get_std(log_std, n_act, use_expln=False):
    """get_std: exp(log_std), or with use_expln exp below 0 and log1p(log_std + epsilon) + 1 above ("avoid NaN: zeros values that are
    below zero" -- statement by statement); without full_std ([HL, 1]) `ones(latent_sde_dim, action_dim) * std` -> [HL, A]."""
    ls = np.asarray(log_std, F32)
    if use_expln:
        below = (np.exp(ls) * (ls <= 0)).astype(F32)
        safe = (ls * (ls > 0) + F32(SDE_EPSILON)).astype(F32)
        above = ((np.log1p(safe) + F32(1.0)) * (ls > 0)).astype(F32)
        std = (below + above).astype(F32)
    else:
        std = np.exp(ls).astype(F32)
    return std if std.shape[1] == n_act else (np.ones((std.shape[0], n_act), F32) * std).astype(F32)


def sde_sigma(latent, log_std, acc=None, n_act=None, use_expln=False):
    """proba_distribution: variance = mm(latent_sde ** 2, get_std(log_std) ** 2); Normal(mean, sqrt(variance + epsilon)).
    latent [B, HL] (the policy's last hidden activations, detached), log_std [HL, A] (or [HL, 1] with n_act given) -> sigma [B, A]."""
    std = sde_get_std(log_std, np.asarray(log_std).shape[1] if n_act is None else n_act, use_expln)
    latent = np.asarray(latent, F32)
    variance = _mm((latent * latent).astype(F32), (std * std).astype(F32), acc)
    return np.sqrt(variance + F32(SDE_EPSILON)).astype(F32)


def normal_log_prob(mean, sigma, actions):
    """torch.distributions.Normal.log_prob summed over the action dimension, sigma per (row, action)."""
    var = (sigma * sigma).astype(F32)
    lp = (-((np.asarray(actions, F32) - mean) ** 2) / (F32(2.0) * var) - np.log(sigma) - F32(math.log(math.sqrt(2 * math.pi)))).astype(F32)
    return lp.sum(axis=1, dtype=F32)


def sde_exploration_matrices(log_std, z, use_expln=False):
    """sample_weights: weights_dist = Normal(0, std); exploration_matrices = rsample((n_envs,)) = z * std.  The standard normals z
    [n_envs, HL, A] are an INPUT (torch's stream cannot be reproduced elsewhere)."""
    z = np.asarray(z, F32)
    return (z * sde_get_std(log_std, z.shape[-1], use_expln)).astype(F32)


def act_sde(p, obs, theta, low=-1.0, high=1.0, activation="tanh", use_expln=False):
    """One rollout-time policy call under gSDE: actions = mean + bmm(latent, theta) [get_noise], log-probs under
    Normal(mean, sigma(latent)).  theta [n_envs, HL, A] = the environments' exploration matrices (one shared [HL, A] matrix is
    broadcast: SB3's `exploration_mat` for foreign batch sizes)."""
    acts_pi, acts_vf = mlp_latents(p, obs, activation=activation)
    latent = acts_pi[-1]
    mean = _linear(latent, p["action_net.weight"], p["action_net.bias"])
    value = _linear(acts_vf[-1], p["value_net.weight"], p["value_net.bias"])[:, 0]
    theta = np.asarray(theta, F32)
    if theta.ndim == 2:
        noise = _mm(latent, theta)
    else:
        noise = np.stack([_mm(latent[i:i + 1], theta[i])[0] for i in range(latent.shape[0])]).astype(F32)
    actions = (mean + noise).astype(F32)
    logp = normal_log_prob(mean, sde_sigma(latent, p["log_std"], n_act=mean.shape[1], use_expln=use_expln), actions)
    return actions, np.clip(actions, F32(low), F32(high)), value, logp


def act(p, obs, eps, low=-1.0, high=1.0, activation="tanh"):
    """One rollout-time policy call [SB3 OnPolicyAlgorithm.collect_rollouts -> policy.forward].
    eps ~ N(0, I) is an INPUT (torch's CPU normal_ stream cannot be reproduced elsewhere).
    Returns raw actions (stored in the buffer), clipped actions (sent to the env), values, log-probs."""
    mean, value = policy_outputs(p, obs, activation=activation)
    std = np.exp(p["log_std"].astype(F32)).astype(F32)
    actions = (mean + np.asarray(eps, F32) * std).astype(F32)  # Normal.rsample
    logp = gaussian_log_prob(mean, p["log_std"], actions)
    clipped = np.clip(actions, F32(low), F32(high))
    return actions, clipped, value, logp


def predict(p, obs, deterministic=True, eps=None, low=-1.0, high=1.0, activation="tanh"):
    """[SB3 BasePolicy.predict] deterministic -> mean; result clipped to the Box (Appendix A.10)."""
    obs = np.asarray(obs)
    single = obs.ndim == 1
    x = obs[None] if single else obs
    mean, _ = policy_outputs(p, x, activation=activation)
    if deterministic:
        a = mean
    else:
        a = (mean + np.asarray(eps, F32) * np.exp(p["log_std"]).astype(F32)).astype(F32)
    a = np.clip(a, F32(low), F32(high))
    return a[0] if single else a


def predict_values(p, obs, activation="tanh"):
    return policy_outputs(p, obs, activation=activation)[1]


# --------------------------------------------------------------------------------------
# GAE(lambda)   [SB3 common/buffers.py RolloutBuffer.compute_returns_and_advantage]
# --------------------------------------------------------------------------------------
def gae(rewards, values, episode_starts, last_values, dones, gamma, gae_lambda):
    """Exact dtype-faithful restatement.  SB3 keeps rewards/values/episode_starts as float32
    arrays but `dones` is the VecEnv's *bool* array, so `1.0 - dones` is float64 and the running
    `last_gae_lam` is float64 for the whole reverse scan, while `delta` for t < T-1 is computed
    in float32 and the python-float coefficients gamma and gamma*lambda are rounded to float32
    when they meet a float32 array.  Stores are float32.  The HIP kernel reproduces this
    bit-for-bit (no transcendental functions, fixed order)."""
    rewards = np.asarray(rewards, F32)
    values = np.asarray(values, F32)
    episode_starts = np.asarray(episode_starts, F32)
    last_values = np.asarray(last_values, F32).ravel()
    dones = np.asarray(dones).astype(bool)
    T = rewards.shape[0]
    adv = np.zeros_like(rewards)
    last_gae_lam = 0
    gamma = float(gamma)
    gl = float(gamma) * float(gae_lambda)
    for step in reversed(range(T)):
        if step == T - 1:
            next_non_terminal = 1.0 - dones  # float64
            next_values = last_values
        else:
            next_non_terminal = 1.0 - episode_starts[step + 1]  # float32
            next_values = values[step + 1]
        delta = rewards[step] + gamma * next_values * next_non_terminal - values[step]
        last_gae_lam = delta + gl * next_non_terminal * last_gae_lam
        adv[step] = last_gae_lam
    returns = (adv + values).astype(F32)
    return adv, returns


def bootstrap_reward(reward, gamma, terminal_value):
    """Time-limit bootstrap `rewards[idx] += gamma * terminal_value` [SB3 collect_rollouts]:
    gamma*V is a float32 torch product, the sum is taken in float64 (SubprocVecEnv rewards are
    float64) and the buffer stores float32."""
    gv = F32(F32(gamma) * F32(terminal_value))
    return F32(np.float64(reward) + np.float64(gv))


# --------------------------------------------------------------------------------------
# minibatching  [SB3 RolloutBuffer.get / swap_and_flatten]
# --------------------------------------------------------------------------------------
def flat_to_tn(flat_idx, T):
    """env-major flat index (n*T + t) -> (t, n)."""
    flat_idx = np.asarray(flat_idx)
    return flat_idx % T, flat_idx // T


def feistel_permutation(n: int, key: int, rounds: int = 6) -> np.ndarray:
    """Counter-based pseudo-random permutation of range(n) used by the engine when no permutation
    is supplied (replaces SB3's `np.random.permutation`, whose MT19937 stream is not reproducible
    on device).  Balanced Feistel network over ceil(log2 n) bits (rounded up to even) with a
    murmur3-style round function, cycle-walked into [0, n).  Pure 32/64-bit integer arithmetic ->
    the HIP implementation is bit-exact against this."""
    bits = max(2, int(n - 1).bit_length())
    if bits & 1:
        bits += 1
    half = bits // 2
    mask = np.uint64((1 << half) - 1)
    k0 = np.uint64(key & 0xFFFFFFFF)
    k1 = np.uint64((key >> 32) & 0xFFFFFFFF)

    def rf(r, rnd):
        x = (r + np.uint64(0x9E3779B9) * np.uint64(rnd + 1) + k0) & np.uint64(0xFFFFFFFF)
        x ^= x >> np.uint64(16)
        x = (x * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF)
        x ^= k1
        x ^= x >> np.uint64(13)
        x = (x * np.uint64(0xC2B2AE35)) & np.uint64(0xFFFFFFFF)
        x ^= x >> np.uint64(16)
        return x & mask

    def enc(v):
        l = (v >> np.uint64(half)) & mask
        r = v & mask
        for rnd in range(rounds):
            l, r = r, l ^ rf(r, rnd)
        return (l << np.uint64(half)) | r

    v = enc(np.arange(n, dtype=np.uint64))
    while True:
        bad = v >= np.uint64(n)
        if not bad.any():
            break
        v[bad] = enc(v[bad])
    return v.astype(np.int64)


# --------------------------------------------------------------------------------------
# loss + hand-derived backward  [SB3 ppo/ppo.py PPO.train, common/policies.py evaluate_actions]
# --------------------------------------------------------------------------------------
@dataclass
class Hyper:
    gamma: float = 0.99
    gae_lambda: float = 0.95
    clip_range: float = 0.2
    ent_coef: float = 0.0
    vf_coef: float = 0.5
    max_grad_norm: float = 0.5
    learning_rate: float = 3e-4
    beta1: float = 0.9
    beta2: float = 0.999
    adam_eps: float = 1e-5
    normalize_advantage: bool = True
    n_epochs: int = 10
    batch_size: int = 64
    clip_range_vf: float | None = None   # SB3 default None: no value-function clipping
    target_kl: float | None = None       # SB3 default None: no early stop
    activation: str = "tanh"             # policy_kwargs activation_fn: "tanh" (SB3's MlpPolicy default) or another of ACTIVATIONS
    use_sde: bool = False                # PPO(use_sde=True): generalised state-dependent exploration, log_std is [HL, A]
    sde_sample_freq: int = -1            # PPO(sde_sample_freq): steps between reset_noise calls inside a rollout (-1: never)
    sde_use_expln: bool = False          # policy_kwargs use_expln (full_std is read off log_std's shape: [HL, A] or [HL, 1])


def normalize_advantages(adv, acc=None):
    """(adv - mean) / (std + 1e-8) with torch's UNBIASED std; skipped when len <= 1."""
    adv = np.asarray(adv, F32)
    if adv.shape[0] <= 1:
        return adv
    sdt = F32 if acc is None else acc
    mean = F32(adv.mean(dtype=sdt))
    std = F32(np.sqrt(np.sum((adv - mean).astype(F32) ** 2, dtype=sdt) / sdt(adv.shape[0] - 1)))
    return ((adv - mean) / (std + F32(1e-8))).astype(F32)


def loss_and_grads(p, obs, actions, old_values, old_log_prob, advantages, returns, h: Hyper,
                   adv_mean_std=None, denom=None, acc=None):
    """One minibatch: loss terms + all parameter gradients (same key order as `p`).

    Hand-derived backward of PPO.train's graph, following torch's sub-gradient conventions
    (SURVEY.md Appendix A.8): clamp passes gradient for lo <= x <= hi inclusive; min routes to
    the smaller operand and splits ties 0.5/0.5.

    adv_mean_std / denom support data-parallel sharding: when a rank holds only a shard of the
    global minibatch it passes the GLOBAL (mean, std) and the GLOBAL batch size so that the sum
    of per-rank gradients equals the single-process gradient.

    acc=np.float64: every contraction over the batch or a hidden width (GEMMs, column sums, loss sums) is
    accumulated in float64 and rounded once to float32 (see `_mm`); element-wise math stays float32."""
    sdt = F32 if acc is None else acc
    obs = np.asarray(obs).astype(F32)
    actions = np.asarray(actions, F32)
    B = obs.shape[0]
    Bg = F32(B if denom is None else denom)
    pre_z = {}
    acts_pi, acts_vf = mlp_latents(p, obs, acc, activation=h.activation, pre=pre_z)
    mean = _linear(acts_pi[-1], p["action_net.weight"], p["action_net.bias"], acc)
    values = _linear(acts_vf[-1], p["value_net.weight"], p["value_net.bias"], acc)[:, 0]
    log_std = p["log_std"].astype(F32)
    std = np.exp(log_std).astype(F32)
    if h.use_sde:   # sigma per (row, action) from the detached latent; entropy = sum_a 0.5 + log sqrt(2 pi) + log sigma
        sigma = sde_sigma(acts_pi[-1], log_std, acc, n_act=mean.shape[1], use_expln=h.sde_use_expln)
        var = (sigma * sigma).astype(F32)
        log_prob = normal_log_prob(mean, sigma, actions)
        entropy = (F32(0.5) + F32(0.5 * math.log(2 * math.pi)) + np.log(sigma)).astype(F32).sum(axis=1, dtype=F32)
    else:
        var = (std * std).astype(F32)
        log_prob = gaussian_log_prob(mean, log_std, actions)
        entropy = gaussian_entropy(log_std, B)

    adv = np.asarray(advantages, F32)
    if h.normalize_advantage and (denom if denom is not None else B) > 1:
        if adv_mean_std is None:
            adv = normalize_advantages(adv, acc)
        else:
            m, s = F32(adv_mean_std[0]), F32(adv_mean_std[1])
            adv = ((adv - m) / (s + F32(1e-8))).astype(F32)

    log_ratio = (log_prob - np.asarray(old_log_prob, F32)).astype(F32)
    ratio = np.exp(log_ratio).astype(F32)
    lo, hi = F32(1.0 - h.clip_range), F32(1.0 + h.clip_range)
    s1 = (adv * ratio).astype(F32)
    s2 = (adv * np.clip(ratio, lo, hi)).astype(F32)
    policy_loss = F32(-(F32(np.minimum(s1, s2).sum(dtype=sdt)) / Bg))
    clip_fraction = F32(F32((np.abs(ratio - F32(1.0)) > F32(h.clip_range)).astype(F32).sum(dtype=sdt)) / Bg)
    ret = np.asarray(returns, F32)
    # value clipping [SB3 PPO.train]: values_pred = old_values + clamp(values - old_values, -c, c); None -> values
    if h.clip_range_vf is None:
        values_pred, vf_pass = values, F32(1.0)
    else:
        c = F32(h.clip_range_vf)
        dv = (values - np.asarray(old_values, F32)).astype(F32)
        values_pred = (np.asarray(old_values, F32) + np.clip(dv, -c, c)).astype(F32)
        vf_pass = ((dv >= -c) & (dv <= c)).astype(F32)  # torch.clamp passes the gradient inside [lo, hi], inclusive
    value_loss = F32(F32(((ret - values_pred) ** 2).sum(dtype=sdt)) / Bg)
    entropy_loss = F32(-(entropy.sum(dtype=F32) / Bg))
    loss = F32(policy_loss + F32(h.ent_coef) * entropy_loss + F32(h.vf_coef) * value_loss)
    approx_kl = F32(F32(((np.exp(log_ratio) - F32(1.0)) - log_ratio).astype(F32).sum(dtype=sdt)) / Bg)

    # ---- backward ----
    in_range = ((ratio >= lo) & (ratio <= hi)).astype(F32)
    w1 = np.where(s1 < s2, F32(1.0), np.where(s1 > s2, F32(0.0), F32(0.5))).astype(F32)
    w2 = (F32(1.0) - w1).astype(F32)
    d_ratio = (-(w1 * adv + w2 * adv * in_range) / Bg).astype(F32)
    g_logp = (d_ratio * ratio).astype(F32)  # dL/dlog_prob_i
    d = (actions - mean).astype(F32)
    g_mean = (g_logp[:, None] * d / var).astype(F32)  # [B, A]
    if h.use_sde:
        # d logp / d sigma^2 = (d^2 / sigma^2 - 1) / (2 sigma^2); d(-mean entropy) / d sigma^2 = -1 / (2 sigma^2 Bg);
        # sigma^2[r, a] = sum_k latent[r, k]^2 exp(2 log_std[k, a]) + eps  ->  d sigma^2 / d log_std[k, a] = 2 latent^2 std^2
        g_var = ((g_logp[:, None] * (d * d / var - F32(1.0)) - F32(h.ent_coef) / Bg) / (F32(2.0) * var)).astype(F32)
        lat = acts_pi[-1]
        g_std2 = _mm((lat * lat).astype(F32).T, g_var, acc)                      # dL / d(std^2) per (latent unit, action)
        if log_std.shape[1] != g_std2.shape[1]:                                  # full_std=False: one std per latent unit, shared by the actions
            g_std2 = g_std2.sum(axis=1, keepdims=True, dtype=F32)
        std_p = sde_get_std(log_std, log_std.shape[1], h.sde_use_expln)          # the parameter's own shape
        if h.sde_use_expln:   # d std / d log_std = exp(ls) below 0, 1 / (1 + ls + eps) above
            dstd = np.where(log_std > 0, F32(1.0) / (F32(1.0) + log_std + F32(SDE_EPSILON)), np.exp(log_std)).astype(F32)
        else:
            dstd = std_p
        g_log_std = (g_std2 * (F32(2.0) * std_p * dstd)).astype(F32)
    else:
        g_log_std = _colsum((g_logp[:, None] * (d * d / var - F32(1.0))).astype(F32), acc)
        # entropy: d(-mean(entropy))/dlog_std_a = -(B_local / B_global)
        g_log_std = (g_log_std + F32(h.ent_coef) * F32(-float(B)) / Bg).astype(F32)
    g_value = (F32(h.vf_coef) * F32(2.0) * (values_pred - ret) * vf_pass / Bg).astype(F32)  # [B]

    grads = OrderedDict((k, None) for k in p.keys())
    grads["log_std"] = g_log_std

    def backprop(prefix, head_w_key, head_b_key, acts, g_out):
        grads[head_w_key] = _mm(g_out.T, acts[-1], acc)
        grads[head_b_key] = _colsum(g_out, acc)
        g_h = _mm(g_out, p[head_w_key], acc)
        layers = _net_layers(p, prefix)
        for li in reversed(range(len(layers))):
            w, _ = layers[li]
            if h.activation in NEEDS_PRE_ACTIVATION:
                g_z = _activation_grad_pre(pre_z[prefix][li], g_h, h.activation)
            else:
                g_z = _activation_grad(acts[li + 1], g_h, h.activation)
            grads[f"{prefix}.{2 * li}.weight"] = _mm(g_z.T, acts[li], acc)
            grads[f"{prefix}.{2 * li}.bias"] = _colsum(g_z, acc)
            g_h = _mm(g_z, w, acc)

    backprop("mlp_extractor.policy_net", "action_net.weight", "action_net.bias", acts_pi, g_mean)
    backprop("mlp_extractor.value_net", "value_net.weight", "value_net.bias", acts_vf, g_value[:, None])

    stats = dict(loss=loss, policy_loss=policy_loss, value_loss=value_loss, entropy_loss=entropy_loss,
                 approx_kl=approx_kl, clip_fraction=clip_fraction)
    aux = dict(values=values, log_prob=log_prob, entropy=entropy, ratio=ratio, adv=adv)
    return stats, grads, aux


def clip_grad_norm(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_ (torch 2.0): norm of per-tensor L2 norms;
    coef = max_norm / (total + 1e-6) clamped to <= 1; every gradient is multiplied by it."""
    norms = np.array([np.sqrt(np.sum(np.asarray(g, F32) ** 2, dtype=F32)) for g in grads.values()], F32)
    total = F32(np.sqrt(np.sum(norms ** 2, dtype=F32)))
    coef = F32(min(F32(max_norm) / (total + F32(1e-6)), F32(1.0)))
    out = OrderedDict((k, (np.asarray(g, F32) * coef).astype(F32)) for k, g in grads.items())
    return out, total


@dataclass
class AdamState:
    exp_avg: "OrderedDict[str, np.ndarray]"
    exp_avg_sq: "OrderedDict[str, np.ndarray]"
    step: int = 0

    @classmethod
    def zeros_like(cls, p):
        return cls(OrderedDict((k, np.zeros_like(v, F32)) for k, v in p.items()),
                   OrderedDict((k, np.zeros_like(v, F32)) for k, v in p.items()), 0)


def adam_step(p, grads, st: AdamState, lr, beta1=0.9, beta2=0.999, eps=1e-5):
    """torch.optim.Adam single-tensor path as of torch 2.0.1 (the version the checkpoints were
    trained with): m = m*b1 + (1-b1)*g ; v = v*b2 + (1-b2)*g*g ;
    p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps); bias corrections in python float64."""
    st.step += 1
    bc1 = 1.0 - beta1 ** st.step
    bc2 = 1.0 - beta2 ** st.step
    step_size = F32(lr / bc1)
    bc2_sqrt = F32(math.sqrt(bc2))
    for k in p.keys():
        g = np.asarray(grads[k], F32)
        m = (st.exp_avg[k] * F32(beta1) + F32(1.0 - beta1) * g).astype(F32)
        v = (st.exp_avg_sq[k] * F32(beta2) + F32(1.0 - beta2) * (g * g)).astype(F32)
        denom = (np.sqrt(v) / bc2_sqrt + F32(eps)).astype(F32)
        p[k] = (p[k] - step_size * (m / denom)).astype(F32)
        st.exp_avg[k], st.exp_avg_sq[k] = m, v
    return p


def train_minibatch(p, st, batch, h: Hyper, adv_mean_std=None, denom=None):
    """batch = (obs, actions, old_values, old_log_prob, advantages, returns) -> one optimizer step.  With
    `h.target_kl`, a minibatch whose approx_kl exceeds 1.5 x target is NOT applied: stats["early_stop"] = True and the
    caller ends train() [SB3 PPO.train: `continue_training = False; break` before optimizer.step()]."""
    stats, grads, _ = loss_and_grads(p, *batch, h, adv_mean_std=adv_mean_std, denom=denom)
    if h.target_kl is not None and float(stats["approx_kl"]) > 1.5 * h.target_kl:
        stats["early_stop"] = True
        stats["grad_norm"] = F32(np.nan)
        return stats
    grads, total = clip_grad_norm(grads, h.max_grad_norm)
    stats["grad_norm"] = total
    adam_step(p, grads, st, h.learning_rate, h.beta1, h.beta2, h.adam_eps)
    return stats


def gather_minibatch(buf, flat_idx):
    """buf: dict of [T,N,...] arrays; flat_idx env-major."""
    T = buf["rewards"].shape[0]
    t, n = flat_to_tn(flat_idx, T)
    return (buf["obs"][t, n], buf["actions"][t, n], buf["values"][t, n], buf["log_probs"][t, n],
            buf["advantages"][t, n], buf["returns"][t, n])


def train(p, st, buf, h: Hyper, perms):
    """PPO.train: `perms[e]` is the epoch-e permutation of range(T*N) (an INPUT).
    Returns per-minibatch stats list."""
    T, N = buf["rewards"].shape
    total = T * N
    out = []
    for e in range(h.n_epochs):
        perm = np.asarray(perms[e])
        for s in range(0, total, h.batch_size):
            idx = perm[s:s + h.batch_size]
            out.append(train_minibatch(p, st, gather_minibatch(buf, idx), h))
            if out[-1].get("early_stop"):
                return out
    return out


# --------------------------------------------------------------------------------------
# learn-loop bookkeeping [SB3 common/on_policy_algorithm.py OnPolicyAlgorithm.learn / collect_rollouts, ppo/ppo.py PPO.train,
# common/callbacks.py CheckpointCallback], reached through /root/reference/src/mobrob/rl_control/ppo.py:73-74 from
# /root/reference/examples/train.py:36-46.  Pinned by REFERENCE-HELD outputs: the counters the five reference checkpoints were
# saved with (tests/golden/reference_counters.json, tests/test_oracle.py::test_learn_loop_counters_equal_the_reference_checkpoints).
# --------------------------------------------------------------------------------------
def learn_loop_counters(total_timesteps, n_steps, n_envs, batch_size, n_epochs, checkpoint_at_timestep=None):
    """The counters SB3 leaves behind after `learn(total_timesteps)` -- or, with `checkpoint_at_timestep`, the ones a
    CheckpointCallback that fires at that `num_timesteps` (inside collect_rollouts, `train.py:37`: save_freq // n_envs vector
    steps) writes into its zip.
        while num_timesteps < total:   collect n_steps vector steps (num_timesteps += n_envs each; callbacks fire per step)
                                       _current_progress_remaining = 1 - num_timesteps / total
                                       train(): n_epochs x ceil(n_steps n_envs / batch_size) Adam steps, _n_updates += n_epochs
    -> dict(num_timesteps, _n_updates, adam_step, _current_progress_remaining, iterations)"""
    per_rollout = n_steps * n_envs
    n_minibatches = -(-per_rollout // batch_size)
    num_timesteps = n_updates = adam_step = iterations = 0
    progress = 1.0
    while num_timesteps < total_timesteps:
        if checkpoint_at_timestep is not None and num_timesteps < checkpoint_at_timestep <= num_timesteps + per_rollout:
            # saved from on_step during this rollout's collection: the rollout is not trained on yet, and
            # _current_progress_remaining is still the previous iteration's
            return dict(num_timesteps=checkpoint_at_timestep, _n_updates=n_updates, adam_step=adam_step,
                        _current_progress_remaining=progress, iterations=iterations)
        num_timesteps += per_rollout
        iterations += 1
        progress = 1.0 - float(num_timesteps) / float(total_timesteps)
        adam_step += n_epochs * n_minibatches
        n_updates += n_epochs
    return dict(num_timesteps=num_timesteps, _n_updates=n_updates, adam_step=adam_step, _current_progress_remaining=progress,
                iterations=iterations)


# --------------------------------------------------------------------------------------
# synthetic environment source (BASELINE.md §3 / SURVEY.md §8d) -- host restatement of the device
# generator; shapes & statistics of the `EnvWrapper` VecEnv contract without a simulator.
# --------------------------------------------------------------------------------------
def explained_variance(y_pred, y_true):
    var_y = np.var(y_true)
    return np.nan if var_y == 0 else float(1 - np.var(y_true - y_pred) / var_y)


@dataclass
class NumpySyntheticVecEnv:
    """obs ~ N(0,1); reward ~ N(0.03, 0.1^2) (+5 on termination); terminated ~ Bernoulli(p_term);
    truncation at `time_limit` steps with a terminal observation.  NumPy PCG64 stream (this is the
    CPU-baseline env source; the device generator uses Philox and is compared statistically)."""
    n_envs: int
    obs_dim: int
    act_dim: int
    p_term: float = 1.0 / 107.0
    time_limit: int = 1000
    seed: int = 0
    rng: np.random.Generator = field(init=False)
    ep_len: np.ndarray = field(init=False)

    def __post_init__(self):
        self.rng = np.random.default_rng(self.seed)
        self.ep_len = np.zeros(self.n_envs, np.int64)

    def reset(self):
        self.ep_len[:] = 0
        return self.rng.standard_normal((self.n_envs, self.obs_dim), dtype=F32)

    def step(self, actions):
        n = self.n_envs
        obs = self.rng.standard_normal((n, self.obs_dim), dtype=F32)
        term = self.rng.random(n) < self.p_term
        rew = (0.03 + 0.1 * self.rng.standard_normal(n, dtype=F32)).astype(F32) + F32(5.0) * term.astype(F32)
        self.ep_len += 1
        trunc = (self.ep_len >= self.time_limit) & ~term
        done = term | trunc
        terminal_obs = obs.copy()
        if done.any():
            obs[done] = self.rng.standard_normal((int(done.sum()), self.obs_dim), dtype=F32)
            self.ep_len[done] = 0
        return obs, rew.astype(F32), done, trunc, terminal_obs


def collect_rollout(p, env, last_obs, last_episode_starts, T, h: Hyper, eps_source):
    """OnPolicyAlgorithm.collect_rollouts restated.  eps_source(t) -> [N,A].
    With h.use_sde, eps_source(t) -> z [N, HL, A]: the standard normals of the reset_noise call SB3 makes at t = 0 and, with
    sde_sample_freq > 0, at every t % sde_sample_freq == 0 (asked for only then)."""
    N, D = last_obs.shape
    A = p["action_net.bias"].shape[0]   # (log_std is [HL, A] under gSDE)
    buf = dict(obs=np.zeros((T, N, D), F32), actions=np.zeros((T, N, A), F32), rewards=np.zeros((T, N), F32),
               episode_starts=np.zeros((T, N), F32), values=np.zeros((T, N), F32), log_probs=np.zeros((T, N), F32))
    dones = np.zeros(N, bool)
    theta = None
    for t in range(T):
        if h.use_sde:
            if t == 0 or (h.sde_sample_freq > 0 and t % h.sde_sample_freq == 0):
                theta = sde_exploration_matrices(p["log_std"], eps_source(t), h.sde_use_expln)
            actions, clipped, values, logp = act_sde(p, last_obs, theta, activation=h.activation, use_expln=h.sde_use_expln)
        else:
            actions, clipped, values, logp = act(p, last_obs, eps_source(t), activation=h.activation)
        new_obs, rewards, dones, trunc, terminal_obs = env.step(clipped)
        rewards = rewards.astype(F32).copy()
        if trunc.any():
            tv = predict_values(p, terminal_obs[trunc], activation=h.activation)
            rewards[trunc] = np.array([bootstrap_reward(r, h.gamma, v) for r, v in zip(rewards[trunc], tv)], F32)
        buf["obs"][t] = last_obs
        buf["actions"][t] = actions
        buf["rewards"][t] = rewards
        buf["episode_starts"][t] = last_episode_starts.astype(F32)
        buf["values"][t] = values
        buf["log_probs"][t] = logp
        last_obs, last_episode_starts = new_obs, dones
    last_values = predict_values(p, last_obs, activation=h.activation)
    adv, ret = gae(buf["rewards"], buf["values"], buf["episode_starts"], last_values, dones, h.gamma, h.gae_lambda)
    buf["advantages"], buf["returns"] = adv, ret
    return buf, last_obs, last_episode_starts
