"""GPU: the x3 kernels (float32 products as six bf16 x bf16 MFMAs on three-way split operands) against the
TORCH goldens made from the reference checkpoints, and against float64 on adversarial operands.

The five reference checkpoints (`/root/reference/data/policies/<env>-ppo.zip`, committed as `tests/golden/<env>.npz` by
`tests/golden/make_fixtures.py`) hold 2x64 networks, i.e. they reach only the all-f32 64-wide kernel families.  A 64-wide
network embeds EXACTLY into a 256-wide one: zero rows / columns for the 192 padded units, whose pre-activations are 0,
activations tanh(0) = 0 and gradients 0 (W2's padded columns are 0 -> dh1 of a padded unit is 0; W3's padded columns are 0 ->
dh2 is 0; h1 of a padded unit is 0 -> dW2's padded columns are 0).  The embedded policy computes the reference's numbers, so
the REAL weights (log_std 3-5, |mean| up to 185), the REAL Adam state and the REAL simulator observations drive
`k_rollout_persistent`, `k_value_batch` and `k_fused_train<.., X3>` against `fwd/*` and `step/*` of the torch goldens, with
the bounds `test_minibatch_step_matches_golden` holds the 64-wide kernels to (north_star: 1e-4 fp32)."""
import os
from collections import OrderedDict

import numpy as np
import pytest

from oracle import ppo_oracle as O
from tests.util import ENVS, golden_adam, golden_hyper, golden_minibatch, golden_params, load_golden, scaled_err

pytestmark = pytest.mark.gpu
H = 256
# the gradient kernel's x3 form needs heads <= 16 wide and observation rows padded to 16 / 32 / 64 columns (engine.hip
# fused_init): drone has 18 actions, turtlebot3 pads 43 -> 48 columns; both still run the 256-wide fused kernels, the
# gradient kernel on the f32 pipe (x3_mode bit 1 clear)
X3_TRAIN = {"point": True, "car": True, "doggo": True, "drone": False, "turtlebot3": False}


def embed(d64):
    """2x64 tensors (parameters, or Adam moments) -> the 2x256 network that computes the same function; padding = 0."""
    out = OrderedDict()
    for k, v in d64.items():
        v = np.asarray(v, np.float32)
        if k.endswith(".0.weight"):                      # [64, D] -> [256, D]
            w = np.zeros((H, v.shape[1]), np.float32); w[:v.shape[0]] = v
        elif k.endswith(".2.weight"):                    # [64, 64] -> [256, 256]
            w = np.zeros((H, H), np.float32); w[:v.shape[0], :v.shape[1]] = v
        elif k in ("action_net.weight", "value_net.weight"):   # [A, 64] -> [A, 256]
            w = np.zeros((v.shape[0], H), np.float32); w[:, :v.shape[1]] = v
        elif k.startswith("mlp_extractor") and k.endswith("bias"):
            w = np.zeros(H, np.float32); w[:v.shape[0]] = v
        else:                                            # log_std, head biases
            w = v.copy()
        out[k] = w
    return out


def split_embedded(k, big, small_shape):
    """(embedded block, padded remainder) of a 256-wide tensor."""
    if big.ndim == 2:
        blk = big[:small_shape[0], :small_shape[1]]
        mask = np.ones(big.shape, bool); mask[:small_shape[0], :small_shape[1]] = False
        return blk, big[mask]
    return big[:small_shape[0]], big[small_shape[0]:]


def _engine(g, **kw):
    from mobrob_amd.engine import PPOEngine
    h = golden_hyper(g)
    D, A = g["last_obs"].shape[1], g["p/log_std"].shape[0]
    base = dict(obs_dim=D, act_dim=A, n_envs=g["last_obs"].shape[0], n_steps=4, batch_size=100, n_epochs=1, pi=(H, H),
                vf=(H, H), gamma=h.gamma, gae_lambda=h.gae_lambda, clip_range=h.clip_range, ent_coef=h.ent_coef,
                vf_coef=h.vf_coef, max_grad_norm=h.max_grad_norm, learning_rate=h.learning_rate,
                adam_betas=(h.beta1, h.beta2), adam_eps=h.adam_eps, forward_x3=True)
    base.update(kw)
    return PPOEngine(**base)


@pytest.mark.parametrize("env", ENVS)
def test_embedded_checkpoint_forward_on_the_x3_rollout_kernels(env):
    """Real weights + real observations through `k_rollout_persistent` (policy forward, sampling, log-prob) and
    `k_value_batch`, both x3 at 256 wide, against torch's `fwd/mean`, `fwd/value`.  The device draws its own noise, so the
    mean is read off the stored action of a rollout taken with log_std = -60 (sigma = 9e-27: action == mean in float32); a
    second rollout with the checkpoint's own log_std checks the stored log-prob of the stored action."""
    g = load_golden(env)
    p = embed(golden_params(g))
    obs = g["last_obs"].astype(np.float32)
    N, D = obs.shape
    T = 4

    def first_step(params):
        e = _engine(g, n_steps=T)
        assert e.x3_mode() & 1
        e.set_params(params)
        e.collect_synthetic()                      # starts the device env; its last observation is slot T of `obs`
        e.synchronize()
        slots = e.read("obs")
        slots[T] = obs                             # the next rollout's first observation (enqueue_rollout: obs[T] -> obs[0])
        e.write("obs", slots)
        e.collect_synthetic()
        e.synchronize()
        out = {k: e.read(k) for k in ("obs", "actions", "values", "log_probs")}
        e.close()
        assert np.array_equal(out["obs"][0], obs)
        return out

    quiet = OrderedDict(p)
    quiet["log_std"] = np.full_like(p["log_std"], -60.0)
    r = first_step(quiet)
    assert scaled_err(r["actions"][0], g["fwd/mean"]) < 1e-4, scaled_err(r["actions"][0], g["fwd/mean"])
    assert scaled_err(r["values"][0], g["fwd/value"]) < 1e-4, scaled_err(r["values"][0], g["fwd/value"])
    # elementwise too: 1e-4 of each mean's own magnitude (|mean| spans 1e-1 .. 185), with an absolute floor of 1e-4 of the scale
    scale = float(np.max(np.abs(g["fwd/mean"])))
    assert np.all(np.abs(r["actions"][0] - g["fwd/mean"]) <= 1e-4 * np.abs(g["fwd/mean"]) + 1e-5 * scale)

    r = first_step(p)
    act = r["actions"][0].astype(np.float64)
    sd = np.exp(p["log_std"].astype(np.float64))
    d = act - g["fwd/mean"].astype(np.float64)
    lp = np.sum(-(d * d) / (2.0 * sd * sd) - np.log(sd) - 0.5 * np.log(2.0 * np.pi), axis=1)
    assert np.allclose(r["log_probs"][0], lp, rtol=1e-4, atol=1e-4), float(np.max(np.abs(r["log_probs"][0] - lp)))
    assert scaled_err(r["values"][0], g["fwd/value"]) < 1e-4


@pytest.mark.parametrize("env", ENVS)
def test_embedded_checkpoint_minibatch_step_on_the_x3_gradient_kernel(env):
    """`test_minibatch_step_matches_golden` for the 256-wide kernels: one optimizer step from the checkpoint's real weights and
    real Adam state on the golden minibatch.  Embedded block == torch's gradient / parameters / moments to the same bounds,
    padded block == 0 exactly."""
    g = load_golden(env)
    p64, st64 = golden_params(g), golden_adam(g)
    p = embed(p64)
    obs, act, old_v, old_lp, adv, ret = golden_minibatch(g)
    B = obs.shape[0]
    e = _engine(g, n_envs=1, n_steps=B, batch_size=B)
    assert e.x3_mode() & 3 == (3 if X3_TRAIN[env] else 1), e.x3_mode()
    e.set_params(p)
    e.set_optimizer_state(embed(st64.exp_avg), embed(st64.exp_avg_sq), st64.step)
    buf = dict(obs=obs[:, None], actions=act[:, None], rewards=np.zeros((B, 1), np.float32),
               episode_starts=np.zeros((B, 1), np.float32), values=old_v[:, None], log_probs=old_lp[:, None],
               advantages=adv[:, None], returns=ret[:, None])
    e.load_rollout(buf, np.zeros(1, np.float32), np.zeros(1, bool))
    e.epoch_begin(np.arange(B))
    e.minibatch_grad(0)
    grads = e.unflatten(e.read("grads"))
    for k, v in grads.items():
        ref = g["step/grad/" + k]
        blk, pad = split_embedded(k, v, ref.shape)
        assert np.max(np.abs(blk - ref)) < 1e-4 * max(1.0, float(np.max(np.abs(ref)))), (k, float(np.max(np.abs(blk - ref))))
        assert not pad.size or not np.any(pad), (k, "gradient of a padded unit", float(np.max(np.abs(pad))))
    e.minibatch_apply()
    stats = e.fetch_step_stats()[-1]
    for i, k in enumerate(["policy_loss", "value_loss", "entropy_loss", "loss", "approx_kl", "clip_fraction", "grad_norm"]):
        ref = float(g["step/" + k])
        assert abs(stats[i] - ref) < 1e-4 * max(1.0, abs(ref)), (k, stats[i], ref)
    newp = e.get_params()
    m, v, step = e.get_optimizer_state()
    assert step == int(g["adam_step"]) + 1
    for k in newp:
        ref = g["step/p/" + k]
        blk, pad = split_embedded(k, newp[k], ref.shape)
        assert np.max(np.abs(blk - ref)) < 1e-6 + 1e-5 * float(np.max(np.abs(ref))), k
        assert not pad.size or not np.any(pad), (k, "a padded parameter moved")
        mb, mpad = split_embedded(k, m[k], ref.shape)
        vb, vpad = split_embedded(k, v[k], ref.shape)
        assert np.allclose(mb, g["step/m/" + k], rtol=1e-3, atol=1e-6), k
        assert np.allclose(vb, g["step/v/" + k], rtol=1e-3, atol=1e-8), k
        assert not np.any(mpad) and not np.any(vpad), k
    e.close()


# ------------------------------------------------------------------------------------------------
# adversarial operands against float64: the x3 kernels are held to 1.5x the f32-pipe kernels' own error
# ------------------------------------------------------------------------------------------------
def _f64_grads(p, buf, idx, h):
    """Gradient of the minibatch with every contraction accumulated in float64 (products of float32 are exact there)."""
    return O.loss_and_grads(p, *O.gather_minibatch(buf, idx), h, acc=np.float64)


def _spread(rng, shape, lo=-4.0, hi=4.0):
    """Random signs x magnitudes log-uniform over e^lo .. e^hi."""
    return (np.exp(rng.uniform(lo, hi, shape)) * rng.choice([-1.0, 1.0], shape)).astype(np.float32)


def _adversarial_case(kind, D, A, rng):
    """Parameters + observations whose hidden-layer products stress the three-way split.  Pre-activations are kept of order
    one (row-normalised) so that tanh does not saturate and hide the matrix products' errors."""
    p = O.init_params(D, A, (H, H), (H, H), seed=5)
    T, N = 16, 64
    obs = rng.standard_normal((T * N, D)).astype(np.float32)
    if kind == "spread":
        # weights and activations spanning e^-4 .. e^+4 (eight orders of magnitude between the extremes of a dot product)
        obs = _spread(rng, (T * N, D))
        for net in ("policy_net", "value_net"):
            w1 = _spread(rng, (H, D))
            z = np.abs(obs.astype(np.float64) @ w1.T.astype(np.float64))
            p[f"mlp_extractor.{net}.0.weight"] = (w1 / np.percentile(z, 90, axis=0)[:, None]).astype(np.float32)
            w2 = _spread(rng, (H, H))
            p[f"mlp_extractor.{net}.2.weight"] = (w2 / (0.6 * np.abs(w2).sum(axis=1, keepdims=True))).astype(np.float32) * 3
    elif kind == "cancel":
        # h1 comes in equal pairs (duplicated rows of W1) and W2 weighs a pair with +c and -c(1 + 2^-12): every layer-2 dot
        # product is the small difference of terms 4000 times larger
        for net in ("policy_net", "value_net"):
            w1 = p[f"mlp_extractor.{net}.0.weight"]
            w1[1::2] = w1[0::2]
            b1 = rng.normal(0, 0.3, H).astype(np.float32); b1[1::2] = b1[0::2]
            p[f"mlp_extractor.{net}.0.bias"] = b1
            c = (rng.standard_normal((H, H // 2)) * 40).astype(np.float32)
            w2 = np.empty((H, H), np.float32)
            w2[:, 0::2] = c
            w2[:, 1::2] = -c * np.float32(1 + 2.0 ** -12)
            p[f"mlp_extractor.{net}.2.weight"] = w2
    elif kind in ("tiny", "subnormal"):
        # layer 1 works 2^-k down: W1 scaled by 2^-k against observations scaled by 2^+k -- the same numbers in exact
        # arithmetic (and on the f32 pipe: power-of-two scalings commute with rounding).  k = 60: every bf16 piece stays a
        # normal number, the x3 kernels must be as exact as at unit scale.  k = 112: the third piece of every weight (2^-16 of
        # it) is a bf16 SUB-NORMAL, which the matrix pipe flushes to zero -- such an operand carries 16 significant bits
        # instead of 24 (DESIGN.md 4.0 "limits"; a weight of 1e-34 next to observations of 1e+33 is outside any policy this
        # engine trains, so the case is held to the documented bound, not to float32's).  In both cases 2 % of the weights sit
        # another 2^-10 below: they must fade out, not poison.
        k = 60 if kind == "tiny" else 112
        s = np.float32(2.0 ** -k)
        obs = (obs * np.float32(2.0 ** k)).astype(np.float32)
        for net in ("policy_net", "value_net"):
            w1 = p[f"mlp_extractor.{net}.0.weight"] * s
            w1[rng.random(w1.shape) < 0.02] *= np.float32(2.0 ** -10)
            p[f"mlp_extractor.{net}.0.weight"] = w1.astype(np.float32)
    p["log_std"] = rng.normal(-0.3, 0.2, A).astype(np.float32)
    p["action_net.weight"] *= 30
    p["value_net.weight"] *= 5
    mean, val = O.policy_outputs(p, obs)
    acts = (mean + rng.standard_normal((T * N, A)).astype(np.float32) * np.exp(p["log_std"])).astype(np.float32)
    buf = dict(obs=obs.reshape(T, N, D), actions=acts.reshape(T, N, A),
               rewards=rng.standard_normal((T, N)).astype(np.float32), episode_starts=np.zeros((T, N), np.float32),
               values=(val + rng.normal(0, 0.1, T * N)).astype(np.float32).reshape(T, N),
               log_probs=(O.gaussian_log_prob(mean, p["log_std"], acts) + rng.normal(0, 0.1, T * N)).astype(np.float32).reshape(T, N))
    buf["advantages"] = rng.standard_normal((T, N)).astype(np.float32)
    buf["returns"] = (buf["values"] + buf["advantages"]).astype(np.float32)
    return p, buf, T, N


@pytest.mark.parametrize("D,A", [(14, 2), (26, 2), (58, 12)])       # observation rows padded to 16 / 32 / 64 columns
@pytest.mark.parametrize("kind", ["spread", "cancel", "tiny", "subnormal"])
def test_x3_gradient_kernel_on_adversarial_operands(kind, D, A):
    """Gradient tensors and loss scalars of one minibatch against the float64-accumulated oracle, x3 kernel and f32-pipe kernel
    side by side, and NumPy's float32 BLAS beside them.  Errors are scaled by the tensor's largest entry.  Bar: over all tensors
    the x3 kernel's worst error is at most 1.5x the worst of the other two float32 evaluations (+ 5e-7: all sit at the rounding
    noise of 1024-term float32 sums, whose order differs), and no single tensor is worse than 4x + 1e-6 (the bound of
    test_x3_gradient_kernel_is_float32_accurate).  The clip range is opened wide (1e9): the surrogate's gradient is
    discontinuous at 1 +- clip and the cancelling case puts 1e-4-level noise on the ratios by construction, so with clipping on
    the test would count clip-boundary flips, not matrix products (those are covered at full size with the boundary rows
    handled: tests/test_full_size_gpu.py)."""
    from mobrob_amd.engine import PPOEngine
    rng = np.random.default_rng(100 + D)
    p, buf, T, N = _adversarial_case(kind, D, A, rng)
    B = T * N
    h = O.Hyper(ent_coef=0.01, n_epochs=1, batch_size=B, clip_range=1e9)
    idx = rng.permutation(B)
    stats, og, aux = _f64_grads(p, buf, idx, h)
    # the same gradient in float32 BLAS (NumPy's sgemm; the reference's arithmetic is torch-CPU's sgemm): its error is part of the
    # yardstick.  A sequential fma chain (v_mfma_f32) profits when cancelling terms are NEIGHBOURS in k -- its partial sums stay
    # small --, a blocked / vectorised summation (BLAS, and the x3 products, whose truncation is relative to the single products)
    # does not; neither is "more float32" than the other.
    _, og32, _ = O.loss_and_grads(p, *O.gather_minibatch(buf, idx), h)
    errs, scal = {"blas": {k: scaled_err(og32[k], og[k]) for k in og}}, {}
    for x3 in (True, False):
        e = PPOEngine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=B, n_epochs=1, pi=(H, H), vf=(H, H),
                      ent_coef=h.ent_coef, clip_range=h.clip_range, forward_x3=x3)
        assert e.x3_mode() & 3 == (3 if x3 else 0)
        e.set_params(p)
        e.load_rollout(buf, np.zeros(N, np.float32), np.zeros(N, bool))
        e.epoch_begin(idx)
        e.minibatch_grad(0)
        got = e.unflatten(e.read("grads"))
        assert all(np.isfinite(v).all() for v in got.values())
        errs[x3] = {k: scaled_err(got[k], og[k]) for k in og}
        e.minibatch_apply()
        scal[x3] = e.fetch_step_stats()[-1]
        e.close()
    ref = {k: max(errs[False][k], errs["blas"][k]) for k in og}      # the better-known float32 evaluations of the same gradient
    report = {k.replace("mlp_extractor.", ""): (f"{errs[True][k]:.1e}", f"{errs[False][k]:.1e}", f"{errs['blas'][k]:.1e}") for k in og}
    worst_x3, worst_f32 = max(errs[True].values()), max(ref.values())
    if kind == "subnormal":
        # documented limit: operands below 2^-110 keep two of their three pieces (16 significant bits, 1.5e-5 relative per
        # product) -- the north_star bar still holds
        assert worst_x3 < 1e-4, (kind, report)
        return
    # 'cancel': every float32 evaluation sits at the conditioning of the data (results 4000x smaller than their terms: 3e-4
    # for BLAS and for v_mfma_f32 alike, above the 1e-4 bar by construction), and WHICH summation order is luckiest is a
    # property of the planted pairs, not of the arithmetic -- the x3 products' own truncation (dropped cross terms: 2^-26 of
    # each product) is below float32's rounding unit.  Measured 1.7x .. 2.2x of the better-known evaluations (profiles/r4/x3_adversarial_gradients.txt);
    # the bar there is 2.5x, everywhere else 1.5x.
    factor = 2.5 if kind == "cancel" else 1.5
    rec = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(rec):
        with open(os.path.join(rec, "x3_adversarial_gradients.txt"), "a") as f:
            f.write(f"{kind} D={D} A={A}: worst x3 {worst_x3:.2e}, worst of (f32 pipe, BLAS) {worst_f32:.2e}; per tensor (x3, f32 pipe, BLAS): {report}\n")
    assert worst_x3 <= factor * worst_f32 + 5e-7, (kind, worst_x3, worst_f32, report)
    for k in og:
        assert errs[True][k] <= 4.0 * ref[k] + 1e-6, (kind, k, report)
        # and to the north_star bar wherever float32 arithmetic itself meets it (the cancelling case is ill-conditioned by
        # construction: there every float32 evaluation sits at the conditioning of the data and only the comparisons above mean something)
        assert errs[True][k] < max(1e-4, factor * ref[k]), (kind, k, report)
    for i, k in enumerate(["policy_loss", "value_loss", "entropy_loss", "loss", "approx_kl"]):
        ref = float(stats[k])
        ex, ef = abs(float(scal[True][i]) - ref), abs(float(scal[False][i]) - ref)
        assert ex <= 1.5 * ef + 2e-6 * max(1.0, abs(ref)), (kind, k, float(scal[True][i]), float(scal[False][i]), ref)


@pytest.mark.parametrize("D,A", [(14, 2), (58, 12)])
@pytest.mark.parametrize("kind", ["spread", "cancel", "tiny", "subnormal"])
def test_x3_forward_kernels_on_adversarial_operands(kind, D, A):
    """The batched value pass (`k_value_batch`, x3) on planted adversarial observations against a float64 evaluation of the
    value network: the x3 kernel is held to 1.5x the f32-pipe kernel's error."""
    from mobrob_amd.engine import PPOEngine
    rng = np.random.default_rng(200 + D)
    p, buf, T, N = _adversarial_case(kind, D, A, rng)
    p64 = {k: v.astype(np.float64) for k, v in p.items()}
    x = buf["obs"].reshape(T * N, D).astype(np.float64)
    h1 = np.tanh(x @ p64["mlp_extractor.value_net.0.weight"].T + p64["mlp_extractor.value_net.0.bias"])
    h2 = np.tanh(h1 @ p64["mlp_extractor.value_net.2.weight"].T + p64["mlp_extractor.value_net.2.bias"])
    ref = (h2 @ p64["value_net.weight"].T + p64["value_net.bias"])[:, 0].reshape(T, N)
    err = {}
    for x3 in (True, False):
        e = PPOEngine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=64, n_epochs=1, pi=(H, H), vf=(H, H), forward_x3=x3)
        e.set_params(p)
        e.collect_synthetic()                      # starts the env; then plant the observations and re-value them:
        e.synchronize()
        slots = e.read("obs")
        # the value pass of a rollout runs over obs[0 .. T]; slot T is carried into the next rollout as its slot 0, the rest is
        # overwritten by the env -- so plant one slot per rollout and read values[0]
        vals = np.empty((T, N), np.float32)
        for t in range(T):
            slots[T] = buf["obs"][t]
            e.write("obs", slots)
            e.collect_synthetic()
            e.synchronize()
            vals[t] = e.read("values")[0]
            slots = e.read("obs")
        e.close()
        assert np.isfinite(vals).all()
        err[x3] = float(np.max(np.abs(vals - ref)))
    scale = max(1.0, float(np.max(np.abs(ref))))
    if kind == "subnormal":                       # documented limit (see _adversarial_case): 16 significant bits; 1e-4 still holds
        assert err[True] < 1e-4 * scale, (kind, err, scale)
        return
    assert err[True] <= 1.5 * err[False] + 2e-7 * scale, (kind, err, scale)
    assert err[True] < max(1e-4 * scale, 1.5 * err[False]), (kind, err, scale)
