"""GPU: the BENCHMARKED shapes against the oracle, and size-independent properties at the same sizes.

BASELINE.json config 3 (doggo 58/12, 2x256, 4096 envs x 1000 steps, minibatch 65536 -> 63 launches per epoch, the
last one 32768 rows) and config 2 (point 14/2, 2x64, 1024 envs x 2048 steps, minibatch 65536 -> 32 launches per
epoch), exactly as `bench.py` runs them: device rollout, device-drawn Feistel permutation, advantage normalisation
on.  Checked against `oracle/ppo_oracle.py`: a 65536-row minibatch gradient (13 tensors, six loss scalars) with the
oracle accumulating in float64, clip + Adam from a non-trivial optimizer state, the short last minibatch, and a
whole epoch of optimizer steps followed step by step.  A whole rollout is too much for the NumPy oracle, so the
rollout itself is covered by properties: permutation coverage, GAE linearity + a bit-exact column slice, gradient
additivity over a split minibatch, bit-reproducibility, conservation checks on a full device rollout."""
import numpy as np
import pytest

from oracle import ppo_oracle as O
from tests.util import scaled_err

pytestmark = pytest.mark.gpu

D, A, H, N = 58, 12, 256, 4096


def _engine(T, B, **kw):
    from mobrob_amd.engine import PPOEngine
    return PPOEngine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=B, n_epochs=1, pi=(H, H), vf=(H, H), **kw)


def test_feistel_permutation_covers_the_full_rollout():
    e = _engine(8, 4096)
    n = 4096 * 1000
    for key in (1, 0x9E3779B97F4A7C15, 2 ** 63 + 12345):
        p = e.feistel_permutation(n, key)
        assert p.min() == 0 and p.max() == n - 1
        assert np.array_equal(np.sort(p), np.arange(n, dtype=np.int64))          # a permutation ...
        assert np.array_equal(p[:4096], O.feistel_permutation(n, key)[:4096])     # ... and the oracle's
        assert abs(np.corrcoef(p[:100000], np.arange(100000))[0, 1]) < 0.02       # not the identity in disguise
    e.close()


def test_gae_is_linear_in_rewards_and_values_at_full_size():
    T = 1000
    e = _engine(T, 65536)
    rng = np.random.default_rng(0)
    es = (rng.random((T, N)) < 0.01).astype(np.float32)
    dones = (rng.random(N) < 0.01).astype(np.float32)

    def gae(r, v, lv):
        e.write("rewards", r); e.write("values", v); e.write("episode_starts", es)
        e.write("last_values", lv); e.write("last_dones", dones)
        e.compute_gae()
        return e.read("advantages"), e.read("returns")

    r1, v1, l1 = (rng.standard_normal((T, N)).astype(np.float32), rng.standard_normal((T, N)).astype(np.float32),
                  rng.standard_normal(N).astype(np.float32))
    r2, v2, l2 = (rng.standard_normal((T, N)).astype(np.float32), rng.standard_normal((T, N)).astype(np.float32),
                  rng.standard_normal(N).astype(np.float32))
    a1, ret1 = gae(r1, v1, l1)
    a2, _ = gae(r2, v2, l2)
    a12, ret12 = gae(r1 + r2, v1 + v2, l1 + l2)
    a3, _ = gae(3.0 * r1, 3.0 * v1, 3.0 * l1)
    scale = float(np.abs(a12).max())
    assert np.max(np.abs(a12 - (a1 + a2))) < 2e-5 * scale      # additivity
    assert np.max(np.abs(a3 - 3.0 * a1)) < 2e-5 * 3 * float(np.abs(a1).max())   # homogeneity
    assert np.array_equal(ret12, a12 + (v1 + v2))              # returns = advantages + values, exactly
    # a column slice agrees bit for bit with the oracle (the scan is independent per env)
    adv, _ = O.gae(r1[:, :8], v1[:, :8], es[:, :8], l1[:8], dones[:8] > 0, 0.99, 0.95)
    assert np.array_equal(a1[:, :8], adv)
    e.close()


def test_minibatch_gradient_is_additive_over_a_split_and_reproducible():
    """grad(mean loss over 65536 rows) == mean of the gradients of its two halves (advantage normalisation off),
    and repeating the launch gives the same bits (deterministic slab reduction)."""
    T, B = 32, 65536
    rng = np.random.default_rng(3)
    p = O.init_params(D, A, (H, H), (H, H), seed=2)
    p["log_std"] = rng.normal(-0.3, 0.1, A).astype(np.float32)
    big = _engine(T, B, normalize_advantage=False, seed=4)
    big.set_params(p)
    big.collect_synthetic(p_term=0.01, time_limit=500)
    big.synchronize()
    roll = {k: big.read(k) for k in ("obs", "actions", "rewards", "episode_starts", "values", "log_probs", "advantages",
                                     "returns", "last_values", "last_dones")}
    perm = np.arange(T * N, dtype=np.int64)           # env-major identity: first minibatch = flat rows [0, 65536)
    big.epoch_begin(perm)
    big.minibatch_grad(0)
    g_full = big.read("grads")
    big.minibatch_grad(0)
    assert np.array_equal(big.read("grads"), g_full)  # run-to-run identical
    big.close()
    half = _engine(T, B // 2, normalize_advantage=False, seed=4)
    half.set_params(p)
    buf = {k: roll[k] for k in ("actions", "rewards", "episode_starts", "values", "log_probs", "advantages", "returns")}
    buf["obs"] = roll["obs"][:T]
    half.load_rollout(buf, roll["last_values"], roll["last_dones"] > 0)
    half.epoch_begin(perm)
    half.minibatch_grad(0); g0 = half.read("grads")
    half.minibatch_grad(1); g1 = half.read("grads")
    half.close()
    mean = 0.5 * (g0.astype(np.float64) + g1.astype(np.float64))
    scale = float(np.abs(g_full).max())
    assert np.max(np.abs(mean - g_full)) < 2e-5 * scale, (float(np.max(np.abs(mean - g_full))), scale)


def test_full_size_device_rollout_conservation():
    """4096 envs x 1000 steps on the device goal environment: every stored quantity is finite, episode boundaries
    are consistent with the Monitor counters, and advantages + values == returns."""
    from mobrob_amd.envs.vec_env import DeviceGoalVecEnv
    T = 1000
    e = _engine(T, 65536, seed=7)
    e.set_params(O.init_params(D, A, (H, H), (H, H), seed=1))
    DeviceGoalVecEnv.for_robot("doggo", N, time_limit=200).collect(e)
    e.synchronize()
    es, adv, val, ret = e.read("episode_starts"), e.read("advantages"), e.read("values"), e.read("returns")
    for k in ("rewards", "log_probs", "actions", "obs"):
        assert np.isfinite(e.read(k)).all(), k
    assert np.isfinite(adv).all() and np.array_equal(ret, adv + val)
    st = e.episode_stats()
    assert st["episodes"] == int(es[1:].sum() + e.read("last_dones").sum())
    assert 4096 * 1000 // 200 <= st["episodes"] and st["ep_len_mean"] <= 200
    e.close()


# ------------------------------------------------------------------------------------------------
# the benchmarked shapes against the oracle
# ------------------------------------------------------------------------------------------------
GOLDEN_RATIO = 0x9E3779B97F4A7C15
STAT_KEYS = ("policy_loss", "value_loss", "entropy_loss", "loss", "approx_kl", "clip_fraction")


def _device_perm_key(seed, draw, rank=0):
    """The key mobrob_ppo_epoch_begin(NULL) derives for its `draw`-th (0-based) device permutation."""
    return ((seed * GOLDEN_RATIO) & (2 ** 64 - 1)) ^ ((rank + 1) << 48) ^ (draw + 1)


def _bench_like_engine(D, A, H, n_envs, T, B, seed, rng, clip_range=0.2):
    """Engine + oracle state set up the way bench.py leaves them after a rollout, with two changes that make the
    comparison bite: stored log-probs are perturbed so the ratios straddle the clip range (an on-policy first
    minibatch has ratio == 1 in every row), and the optimizer starts from a non-zero Adam state."""
    from mobrob_amd.engine import PPOEngine
    h = O.Hyper(gamma=0.99, gae_lambda=0.95, ent_coef=0.01, n_epochs=1, batch_size=B, learning_rate=3e-4, clip_range=clip_range)
    e = PPOEngine(obs_dim=D, act_dim=A, n_envs=n_envs, n_steps=T, batch_size=B, n_epochs=1, pi=(H, H), vf=(H, H),
                  gamma=h.gamma, gae_lambda=h.gae_lambda, ent_coef=h.ent_coef, learning_rate=h.learning_rate,
                  clip_range=clip_range, seed=seed)
    p = O.init_params(D, A, (H, H), (H, H), seed=seed)
    p["log_std"] = rng.normal(-0.3, 0.2, A).astype(np.float32)
    p["action_net.weight"] *= 30
    for k in p:
        if k.endswith("bias"):
            p[k] = rng.normal(0, 0.05, p[k].shape).astype(np.float32)
    # Adam moments of a run in progress: |m| / sqrt(v) of order one, so a step moves a parameter by about lr.  (Moments
    # with v << m^2 make single steps of 1e-2 and push the policy so far from the stored actions that log-ratios reach
    # several units; the clip-range membership of many rows then hinges on the last bits of log-prob and the test
    # measures the conditioning of its own inputs instead of the kernel.)
    st = O.AdamState(type(p)((k, rng.normal(0, 1e-3, v.shape).astype(np.float32)) for k, v in p.items()),
                     type(p)((k, (1e-6 * (0.5 + 1.5 * rng.random(v.shape))).astype(np.float32)) for k, v in p.items()), 1000)
    e.set_params(p)
    e.set_optimizer_state(st.exp_avg, st.exp_avg_sq, st.step)
    e.collect_synthetic(p_term=0.01, time_limit=500)
    lp = e.read("log_probs")
    e.write("log_probs", lp + rng.normal(0, 0.12, lp.shape).astype(np.float32))
    buf = {k: e.read(k) for k in ("actions", "rewards", "episode_starts", "values", "log_probs", "advantages", "returns")}
    buf["obs"] = e.read("obs")[:T]
    return e, p, st, buf, h


def _check_grad(e, p, buf, idx, h, mb, tag):
    """The clipped surrogate's gradient is DISCONTINUOUS in the ratio at 1 +- clip, and the gradient of a minibatch is a
    sum of B random-signed row terms, so one row is ~1/sqrt(B) of it (0.5 % at 32768 rows).  A row whose ratio sits
    within float32 rounding of a clip boundary is in range for one correct implementation and out of range for another
    (measured: the fused kernel and the generic GEMM chain agree to 1.5e-6 with each other and both differ from the
    oracle by 4.5e-3 on such a minibatch; about one minibatch in five of this size contains such a row).  Those rows are
    moved off the boundary (stored log-prob changed by 0.01, on the device and in the oracle's buffer) before comparing."""
    stats, og, aux = O.loss_and_grads(p, *O.gather_minibatch(buf, idx), h, acc=np.float64)
    lo, hi = 1.0 - h.clip_range, 1.0 + h.clip_range
    near = (np.abs(aux["ratio"] - lo) < 2e-5) | (np.abs(aux["ratio"] - hi) < 2e-5)
    if near.any():
        # the test adapts its own input here: keep that to the handful of rows the argument above predicts (a 4e-5-wide
        # band around two boundaries holds ~1e-4 of the rows of a minibatch whose ratios spread over ~0.5)
        assert int(near.sum()) <= 16, f"{tag}: {int(near.sum())} rows within 2e-5 of a clip boundary -- not a rounding artefact"
        t, n = O.flat_to_tn(idx[near], buf["rewards"].shape[0])
        buf["log_probs"][t, n] -= np.float32(0.01)
        e.write("log_probs", buf["log_probs"])
        stats, og, aux = O.loss_and_grads(p, *O.gather_minibatch(buf, idx), h, acc=np.float64)
        print(f"{tag}: {int(near.sum())} row(s) moved off the clip boundary")
    e.minibatch_grad(mb)
    got = e.unflatten(e.read("grads"))
    frac_clipped = float(np.mean((aux["ratio"] < 0.8) | (aux["ratio"] > 1.2)))
    errs = {k: scaled_err(got[k], og[k]) for k in og}
    assert max(errs.values()) < 1e-4, (tag, {k: f"{v:.2e}" for k, v in errs.items()}, got["log_std"], og["log_std"])
    return stats, og, frac_clipped


def _check_apply(e, p, st, og, stats, h, tag):
    e.minibatch_apply()
    row = e.fetch_step_stats(1)[0]
    clipped, total = O.clip_grad_norm(og, h.max_grad_norm)
    O.adam_step(p, clipped, st, h.learning_rate, h.beta1, h.beta2, h.adam_eps)
    for i, k in enumerate(STAT_KEYS):
        ref = float(stats[k])
        assert abs(float(row[i]) - ref) < 1e-4 * max(1.0, abs(ref)), (tag, k, float(row[i]), ref)
    assert abs(float(row[6]) - float(total)) < 1e-4 * max(1.0, float(total)), (tag, "grad_norm")
    newp = e.get_params()
    m, v, step = e.get_optimizer_state()
    assert step == st.step
    for k in p:
        assert np.max(np.abs(newp[k] - p[k])) < 1e-5, (tag, k, float(np.max(np.abs(newp[k] - p[k]))))
        assert scaled_err(m[k], st.exp_avg[k]) < 1e-4 and scaled_err(v[k], st.exp_avg_sq[k]) < 1e-4, (tag, k)


@pytest.mark.parametrize("shape", [dict(name="doggo-4096env-2x256", D=58, A=12, H=256, N=4096, T=1000),
                                   dict(name="point-1024env-2x64", D=14, A=2, H=64, N=1024, T=2048)])
def test_benchmarked_minibatch_matches_oracle(shape):
    """One full-size optimizer step of the bench workload (grad + loss scalars, then clip + Adam), and the gradient
    of the LAST minibatch of the epoch (32768 rows at the headline shape: the short-launch path)."""
    D, A, H, n_envs, T = (shape[k] for k in "DAHNT")
    B, seed = 65536, 11
    rng = np.random.default_rng(5)
    e, p, st, buf, h = _bench_like_engine(D, A, H, n_envs, T, B, seed, rng)
    total = T * n_envs
    nmb = -(-total // B)
    assert e.n_minibatches == nmb
    perm = O.feistel_permutation(total, _device_perm_key(seed, 0))
    e.epoch_begin(None)                                   # device-drawn permutation, as in bench.py
    stats, og, frac = _check_grad(e, p, buf, perm[:B], h, 0, shape["name"] + "/mb0")
    assert 0.02 < frac < 0.6, frac                        # both clip branches populated
    _check_apply(e, p, st, og, stats, h, shape["name"] + "/mb0")
    last = perm[(nmb - 1) * B:]
    assert len(last) == total - (nmb - 1) * B
    _check_grad(e, p, buf, last, h, nmb - 1, shape["name"] + "/last")   # at the post-step parameters
    e.close()


def _second_engine(shape, h, seed, B, buf, last_values, last_dones, **kw):
    """An engine on the same buffers as the one that rolled out (same seed -> the same device-drawn permutations)."""
    from mobrob_amd.engine import PPOEngine
    D, A, H, n_envs, T = (shape[k] for k in "DAHNT")
    e = PPOEngine(obs_dim=D, act_dim=A, n_envs=n_envs, n_steps=T, batch_size=B, n_epochs=1, pi=(H, H), vf=(H, H),
                  gamma=h.gamma, gae_lambda=h.gae_lambda, ent_coef=h.ent_coef, learning_rate=h.learning_rate,
                  clip_range=h.clip_range, seed=seed, **kw)
    e.load_rollout(buf, last_values, last_dones)
    return e


def _copy(d):
    return type(d)((k, v.copy()) for k, v in d.items())


def run_epoch_free(shape, seed, rng_seed, clip_range):
    """One whole epoch of the bench workload through the single C call (`mobrob_ppo_train`: 63 / 32 launches incl. the short
    last one), free running, on BOTH matrix pipes (x3 kernels and `forward_x3 = 0`; the 64-wide nets have one), against the
    oracle following every optimizer step in float32 BLAS (SB3-CPU's arithmetic).
    -> ({pipe: {tensor: max |parameter - oracle|}}, the oracle's per-step stats, {pipe: TrainStats})."""
    D, A, H, n_envs, T = (shape[k] for k in "DAHNT")
    B = 65536
    rng = np.random.default_rng(rng_seed)
    e, p, st, buf, h = _bench_like_engine(D, A, H, n_envs, T, B, seed, rng, clip_range=clip_range)
    total = T * n_envs
    nmb = -(-total // B)
    perm = O.feistel_permutation(total, _device_perm_key(seed, 0))
    p0, m0, v0, step0 = _copy(p), _copy(st.exp_avg), _copy(st.exp_avg_sq), st.step
    last_values, last_dones = e.read("last_values"), e.read("last_dones") > 0
    ostats = O.train(p, st, buf, h, perm[None])
    assert st.step == step0 + nmb
    out, stats = _free_run(shape, h, seed, B, buf, last_values, last_dones, p0, m0, v0, step0, p, st.step, first_engine=e)
    return out, ostats, stats


def _free_run(shape, h, seed, B, buf, last_values, last_dones, p0, m0, v0, step0, p_ref, step_ref, first_engine=None):
    """`mobrob_ppo_train` (one epoch, device-drawn permutation as in bench.py) from (p0, m0, v0, step0) on every matrix pipe the
    shape has -> ({pipe: {tensor: max |parameter - p_ref|}}, {pipe: TrainStats})"""
    H = shape["H"]
    nmb = -(-shape["T"] * shape["N"] // B)
    out, stats = {}, {}
    pipes = ("x3", "f32") if H == 256 else ("f32",)
    for pipe in pipes:
        if pipe == pipes[0] and first_engine is not None:
            eng = first_engine                                # the engine that rolled out
        else:                                                 # an engine on the same buffers, same seed (-> the same device-drawn permutations)
            eng = _second_engine(shape, h, seed, B, buf, last_values, last_dones, forward_x3=(pipe == "x3"))
            eng.set_params(p0)
            eng.set_optimizer_state(m0, v0, step0)
        assert eng.x3_mode() & 3 == (3 if pipe == "x3" else 0)
        stats[pipe] = eng.train(None)
        assert stats[pipe]["n_minibatches"] == nmb
        newp = eng.get_params()
        _, _, step = eng.get_optimizer_state()
        assert step == step_ref
        out[pipe] = {k: float(np.max(np.abs(newp[k] - p_ref[k]))) for k in p_ref}
        eng.close()
    return out, stats


SHAPES = [dict(name="doggo-4096env-2x256", D=58, A=12, H=256, N=4096, T=1000),
          dict(name="point-1024env-2x64", D=14, A=2, H=64, N=1024, T=2048)]

# ------------------------------------------------------------------------------------------------
# ONE float64-accumulating oracle epoch per shape and SESSION (the three epoch tests below each used to walk
# the oracle through the same 63 steps -- 494 s of GPU-box time, most of it NumPy).  The teacher trajectory -- the oracle's state
# in front of every optimizer step, its gradient and statistics, the rows it had to move off a clip boundary -- is computed
# on first use and shared by the step-by-step test (which replays it against both engines) and the clip-0.2 free-running test
# (which compares the engines' free epoch with the trajectory's end).
# ------------------------------------------------------------------------------------------------
_TEACHER = {}


def _clip_boundary_rows(p, buf, idx, h):
    """rows of the minibatch whose ratio lies within 2e-5 of 1 +- clip at parameters `p` (policy forward only, float32)"""
    obs, actions, _, old_lp, _, _ = O.gather_minibatch(buf, idx)
    x = np.asarray(obs, np.float32)
    for w, b in O._net_layers(p, "mlp_extractor.policy_net"):      # the policy network only
        x = np.tanh(x @ w.T + b, dtype=np.float32)
    mean = x @ p["action_net.weight"].T + p["action_net.bias"]
    ratio = np.exp(O.gaussian_log_prob(mean, p["log_std"], actions) - old_lp)
    lo, hi = 1.0 - h.clip_range, 1.0 + h.clip_range
    return (np.abs(ratio - lo) < 2e-5) | (np.abs(ratio - hi) < 2e-5)


def _teacher_epoch(shape):
    key = shape["name"]
    if key in _TEACHER:
        return _TEACHER[key]
    D, A, H, n_envs, T = (shape[k] for k in "DAHNT")
    B, seed = 65536, 23
    rng = np.random.default_rng(6)
    e, p, st, buf, h = _bench_like_engine(D, A, H, n_envs, T, B, seed, rng)
    last_values, last_dones = e.read("last_values"), e.read("last_dones") > 0
    e.close()
    total = T * n_envs
    nmb = -(-total // B)
    perm = O.feistel_permutation(total, _device_perm_key(seed, 0))
    tr = dict(h=h, seed=seed, B=B, nmb=nmb, perm=perm, last_values=last_values, last_dones=last_dones, lp0=buf["log_probs"].copy(),
              p0=_copy(p), m0=_copy(st.exp_avg), v0=_copy(st.exp_avg_sq), step0=st.step, steps=[], moved_total=0)
    for mb in range(nmb):
        idx = perm[mb * B:(mb + 1) * B]
        # rows on a clip boundary AT the oracle's state are moved off it before the step's gradient is formed (see the test below);
        # a row belongs to one minibatch of the epoch, so the moves of later steps never touch an earlier step's inputs
        near = _clip_boundary_rows(p, buf, idx, h)
        moved = None
        if near.any():
            assert int(near.sum()) <= 16, f"step {mb}: {int(near.sum())} rows within 2e-5 of a clip boundary -- not a rounding artefact"
            tr["moved_total"] += int(near.sum())
            t, n = O.flat_to_tn(idx[near], T)
            buf["log_probs"][t, n] -= np.float32(0.01)
            moved = (t, n)
        stats, og, aux = O.loss_and_grads(p, *O.gather_minibatch(buf, idx), h, acc=np.float64)
        clipped, total_norm = O.clip_grad_norm(og, h.max_grad_norm)
        rec = dict(p=_copy(p), m=_copy(st.exp_avg), v=_copy(st.exp_avg_sq), step=st.step, og=og, stats=stats,
                   total_norm=float(total_norm), moved=moved)
        O.adam_step(p, clipped, st, h.learning_rate, h.beta1, h.beta2, h.adam_eps)
        rec.update(p_after=_copy(p), m_after=_copy(st.exp_avg), v_after=_copy(st.exp_avg_sq), step_after=st.step)
        tr["steps"].append(rec)
    tr["buf"] = buf                       # log-probs as the epoch left them: every moved row moved
    tr["p_end"], tr["step_end"] = _copy(p), st.step
    _TEACHER[key] = tr
    return tr


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: s["name"])
def test_benchmarked_epoch_step_by_step_matches_oracle(shape):
    """EVERY optimizer step of an epoch of the bench workload, on both matrix pipes, against the oracle at the north_star
    bounds: the gradient of the step's 65536-row minibatch (13 tensors <= 1e-4 of scale, six loss scalars), then clip + Adam
    (parameters <= 1e-5, moments <= 1e-4) -- all 63 (32) steps incl. the short last one, with the clip range at SB3's 0.2.

    The engines are put on the oracle's state before every step (set_params / set_optimizer_state) instead of running free.
    Why: PPO's clipped surrogate has a gradient that is DISCONTINUOUS in the ratio at 1 +- clip, one row is ~1/sqrt(B) of a
    minibatch gradient, and an epoch consumes ~300 rows whose ratio lies within 2e-5 of a boundary (measured:
    profiles/r4/epoch_margin.txt) -- so two correct float32 implementations, e.g. this oracle with float32 and with float64
    accumulation, leave the same boundary row on different sides sooner or later, after which the trajectories differ by
    ~1e-4 in the parameters and every later step sees different ratios (the effect compounds: moving the flagged rows off the
    boundaries moves the later steps' ratios by 1e-6 .. 1e-5 and flags as many new rows).  What CAN be held to 1e-4 / 1e-5 is
    every step from a common state, with the rows that sit on a boundary AT that state moved off it (<= 16 per step, printed):
    that is what this test does, 63 times.  The free-running epoch is covered twice below: with the clip range opened (no
    discontinuity: 1e-4 holds after 63 free steps on both pipes) and at 0.2 with the bound the discontinuity allows."""
    H, T, n_envs = shape["H"], shape["T"], shape["N"]
    tr = _teacher_epoch(shape)
    h, seed, B, nmb = tr["h"], tr["seed"], tr["B"], tr["nmb"]
    # the engines start on the rollout as it was BEFORE any row was moved; the rows the teacher moved in front of step mb are
    # moved here in front of step mb too (a row belongs to one minibatch of the epoch)
    buf0 = dict(tr["buf"])
    lp = tr["lp0"].copy()
    buf0["log_probs"] = lp
    engines = {}
    for pipe in (("x3", "f32") if H == 256 else ("f32",)):
        eng = _second_engine(shape, h, seed, B, buf0, tr["last_values"], tr["last_dones"], forward_x3=(pipe == "x3"))
        assert eng.x3_mode() & 3 == (3 if pipe == "x3" else 0)
        eng.epoch_begin(None)                             # device-drawn permutation, as in bench.py
        engines[pipe] = eng
    worst_g, worst_p = 0.0, 0.0
    for mb, rec in enumerate(tr["steps"]):
        for pipe, eng in engines.items():
            tag = f"{shape['name']}/{pipe}/step{mb}"
            eng.set_params(rec["p"])                      # common state: the oracle's
            eng.set_optimizer_state(rec["m"], rec["v"], rec["step"])
            if rec["moved"] is not None:
                if eng is next(iter(engines.values())):
                    lp[rec["moved"]] -= np.float32(0.01)
                eng.write("log_probs", lp)
            eng.minibatch_grad(mb)
            got = eng.unflatten(eng.read("grads"))
            errs = {k: scaled_err(got[k], rec["og"][k]) for k in rec["og"]}
            worst_g = max(worst_g, max(errs.values()))
            assert max(errs.values()) < 1e-4, (tag, {k: f"{v:.2e}" for k, v in errs.items()})
            eng.minibatch_apply()
            row = eng.fetch_step_stats(1)[0]
            for i, k in enumerate(STAT_KEYS):
                ref = float(rec["stats"][k])
                assert abs(float(row[i]) - ref) < 1e-4 * max(1.0, abs(ref)), (tag, k, float(row[i]), ref)
            assert abs(float(row[6]) - rec["total_norm"]) < 1e-4 * max(1.0, rec["total_norm"]), (tag, "grad_norm")
            newp = eng.get_params()
            m, v, step = eng.get_optimizer_state()
            assert step == rec["step_after"]
            for k in newp:
                d = float(np.max(np.abs(newp[k] - rec["p_after"][k])))
                worst_p = max(worst_p, d)
                assert d < 1e-5, (tag, k, d)
                assert scaled_err(m[k], rec["m_after"][k]) < 1e-4 and scaled_err(v[k], rec["v_after"][k]) < 1e-4, (tag, k)
    print(f"{shape['name']}: {nmb} steps x {list(engines)}: worst gradient error {worst_g:.2e} of scale, worst parameter "
          f"difference after a step {worst_p:.2e}; {tr['moved_total']} row(s) of {T * n_envs} moved off a clip boundary")
    for eng in engines.values():
        eng.close()


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: s["name"])
def test_benchmarked_epoch_free_running_without_the_clip_discontinuity(shape):
    """63 (32) FREE-RUNNING optimizer steps through `mobrob_ppo_train` with the clip range opened to 1e9 (the surrogate is then
    smooth: -mean(adv * ratio)): rounding is all that separates the trajectories, and after the whole epoch every parameter is
    within 1e-4 (north_star) of the oracle's on the x3 kernels AND with every product on `v_mfma_f32`."""
    errs, ostats, stats = run_epoch_free(shape, 23, 6, clip_range=1e9)
    print(shape["name"], "; ".join(f"{pipe}: worst {max(e.values()):.2e}" for pipe, e in errs.items()))
    for pipe, e in errs.items():
        for k, v in e.items():
            assert v < 1e-4, (pipe, k, v)
        for k in STAT_KEYS + ("grad_norm",):
            ref = float(np.mean([float(s[k]) for s in ostats]))
            assert abs(stats[pipe][k] - ref) < 2e-4 * max(1.0, abs(ref)), (pipe, k, stats[pipe][k], ref)


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: s["name"])
def test_benchmarked_epoch_matches_oracle(shape):
    """The same epoch FREE-RUNNING at SB3's clip range 0.2 (one `mobrob_ppo_train` call per pipe from the teacher epoch's start
    state, on the rollout as the teacher left it: its boundary rows moved) against the END of the teacher trajectory -- the
    float64-accumulating oracle's own free run.  The largest parameter deviation is bimodal in the DATA: 2e-7 .. 2e-5 when no
    row's ratio falls within rounding of a clip boundary in any step, ~2e-4 when one does -- for this engine on either pipe and
    for the oracle against itself (float32 vs float64 accumulation): profiles/r4/epoch_margin.txt, profiles/r5/epoch_margin_oracle.txt.
    5e-4 is the bound the discontinuity allows; the 1e-4 / 1e-5 bounds are enforced step by step above."""
    tr = _teacher_epoch(shape)
    errs, stats = _free_run(shape, tr["h"], tr["seed"], tr["B"], tr["buf"], tr["last_values"], tr["last_dones"],
                            tr["p0"], tr["m0"], tr["v0"], tr["step0"], tr["p_end"], tr["step_end"])
    print(shape["name"], "; ".join(f"{pipe}: worst {max(e.values()):.2e}" for pipe, e in errs.items()))
    ostats = [r["stats"] for r in tr["steps"]]
    norms = [r["total_norm"] for r in tr["steps"]]
    for pipe, e in errs.items():
        for k, v in e.items():
            assert v < 5e-4, (pipe, k, v)
        for k in STAT_KEYS:
            ref = float(np.mean([float(s[k]) for s in ostats]))
            assert abs(stats[pipe][k] - ref) < 2e-4 * max(1.0, abs(ref)), (pipe, k, stats[pipe][k], ref)
        ref = float(np.mean(norms))
        assert abs(stats[pipe]["grad_norm"] - ref) < 2e-4 * max(1.0, abs(ref)), (pipe, "grad_norm", stats[pipe]["grad_norm"], ref)


def test_full_size_point_persistent_rollout_conservation():
    """Config 2's rollout (1024 envs x 2048 steps, 2x64 nets: the H=64 persistent rollout kernel) on the device
    goal environment: finite everywhere, Monitor counters consistent with the stored episode starts,
    returns == advantages + values, stored values and log-probs reproduce from the stored observations/actions
    (oracle forward on a slice), GAE bit-exact on a column slice."""
    from mobrob_amd.engine import PPOEngine
    from mobrob_amd.envs.vec_env import DeviceGoalVecEnv
    Dp, Ap, Hp, n_envs, T = 14, 2, 64, 1024, 2048
    e = PPOEngine(obs_dim=Dp, act_dim=Ap, n_envs=n_envs, n_steps=T, batch_size=65536, n_epochs=1, pi=(Hp, Hp), vf=(Hp, Hp),
                  gae_lambda=0.5, seed=3)
    p = O.init_params(Dp, Ap, (Hp, Hp), (Hp, Hp), seed=4)
    e.set_params(p)
    DeviceGoalVecEnv.for_robot("point", n_envs, time_limit=300).collect(e)
    e.synchronize()
    es, adv, val, ret = e.read("episode_starts"), e.read("advantages"), e.read("values"), e.read("returns")
    rew, lp, act, obs = e.read("rewards"), e.read("log_probs"), e.read("actions"), e.read("obs")
    for k, a in dict(rewards=rew, log_probs=lp, actions=act, obs=obs, adv=adv).items():
        assert np.isfinite(a).all(), k
    assert np.array_equal(ret, adv + val)
    st = e.episode_stats()
    assert st["episodes"] == int(es[1:].sum() + e.read("last_dones").sum())
    assert st["episodes"] > 0 and st["ep_len_mean"] <= 300 and st["episodes"] * 300 >= n_envs * (T - 300)
    cols = slice(100, 116)
    mean, v = O.policy_outputs(p, obs[:T, cols].reshape(-1, Dp))
    assert scaled_err(val[:, cols].reshape(-1), v) < 1e-4
    olp = O.gaussian_log_prob(mean, p["log_std"], act[:, cols].reshape(-1, Ap))
    assert np.allclose(lp[:, cols].reshape(-1), olp, rtol=1e-4, atol=1e-3)
    oadv, _ = O.gae(rew[:, cols], val[:, cols], es[:, cols], e.read("last_values")[cols], e.read("last_dones")[cols] > 0, 0.99, 0.5)
    assert np.array_equal(adv[:, cols], oadv)
    e.close()
