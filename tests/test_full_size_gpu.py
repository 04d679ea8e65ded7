"""GPU: size-independent properties at BASELINE.json's full sizes (doggo 58/12, 2x256, 4096 envs, 1000 steps,
minibatch 65536), where the NumPy oracle would take minutes: permutation coverage, GAE linearity, gradient
additivity over a split minibatch, bit-reproducibility, and conservation checks on a full device rollout."""
import numpy as np
import pytest

from oracle import ppo_oracle as O

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
