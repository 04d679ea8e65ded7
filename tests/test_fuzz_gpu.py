"""GPU: randomised shapes through rollout storage, GAE and a full update on every kernel family vs the oracle."""
import numpy as np
import pytest

from oracle import ppo_oracle as O
from tests.util import synthetic_rollout

pytestmark = pytest.mark.gpu

_rng = np.random.default_rng(20240607)
CASES = []
for i in range(36):
    H = int(_rng.choice([256, 64, 64, 256, 24, 128]))
    D = int(_rng.integers(1, 65))
    A = int(_rng.integers(1, 33))
    N = int(_rng.integers(1, 90))
    T = int(_rng.integers(1, 14))
    B = int(_rng.integers(1, 2 * N * T + 2))
    CASES.append((H, D, A, N, T, B, int(_rng.integers(1, 4)), bool(_rng.integers(0, 2)), float(_rng.choice([0.0, 0.01]))))


@pytest.mark.parametrize("H,D,A,N,T,B,E,normalize,ent", CASES)
def test_random_shape_update_matches_oracle(H, D, A, N, T, B, E, normalize, ent):
    from mobrob_amd.engine import PPOEngine
    rng = np.random.default_rng(H * 1000 + D * 31 + A)
    p = O.init_params(D, A, (H, H), (H, H), seed=D + A)
    p["log_std"] = rng.normal(-0.2, 0.2, A).astype(np.float32)
    buf, lv, dones = synthetic_rollout(T, N, D, A, seed=N + T)
    mean, val = O.policy_outputs(p, buf["obs"].reshape(T * N, D))
    buf["log_probs"] = (O.gaussian_log_prob(mean, p["log_std"], buf["actions"].reshape(T * N, A))
                        + rng.normal(0, 0.05, T * N)).astype(np.float32).reshape(T, N)   # off-policy ratio != 1: clipping active
    buf["values"] = val.reshape(T, N)
    h = O.Hyper(n_epochs=E, batch_size=B, ent_coef=ent, normalize_advantage=normalize)
    e = PPOEngine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=B, n_epochs=E, pi=(H, H), vf=(H, H),
                  ent_coef=ent, normalize_advantage=normalize)
    e.set_params(p)
    e.load_rollout(buf, lv, dones)
    e.compute_gae()
    adv, ret = O.gae(buf["rewards"], buf["values"], buf["episode_starts"], lv, dones, h.gamma, h.gae_lambda)
    assert np.array_equal(e.read("advantages"), adv) and np.array_equal(e.read("returns"), ret)
    buf["advantages"], buf["returns"] = adv, ret
    perms = np.stack([rng.permutation(T * N) for _ in range(E)])
    e.train(perms)
    q = {k: v.copy() for k, v in p.items()}
    O.train(q, O.AdamState.zeros_like(q), buf, h, perms)
    got = e.get_params()
    single = normalize and (min(B, T * N) == 1 or (T * N) % B == 1)  # a 1-sample minibatch has no std: NaN in SB3 as well
    if not single:
        for k in q:
            err = float(np.max(np.abs(got[k] - q[k])))
            assert err < 3e-4, (k, err)
    e.close()


_rng2 = np.random.default_rng(20261004)
EPOCH_CASES = []
for i in range(24):
    D = int(_rng2.integers(1, 65))
    A = int(_rng2.integers(1, 33))
    N = int(_rng2.integers(1, 40))
    T = int(_rng2.integers(2, 60))
    B = int(_rng2.integers(2, min(2048, N * T) + 1))          # at most 64 tiles per minibatch: the epoch kernel's range
    EPOCH_CASES.append((D, A, N, T, B, int(_rng2.integers(1, 4)), bool(_rng2.integers(0, 2)), float(_rng2.choice([0.0, 0.01]))))


@pytest.mark.parametrize("D,A,N,T,B,E,normalize,ent", EPOCH_CASES)
def test_random_shape_epoch_kernel_equals_three_launches(D, A, N, T, B, E, normalize, ent):
    """k_epoch64 (one co-operative launch per epoch, csrc/kernels_epoch64.h) against the three launches per optimizer step on random
    64-wide shapes -- observation widths 1 .. 64 (all four padded widths), 1 .. 32 actions (every loss-stage variant), ragged last
    minibatches, one-row tiles: parameters, both Adam moments and the logged statistics must be the same BITS."""
    from mobrob_amd.engine import PPOEngine
    H = 64
    rng = np.random.default_rng(D * 131 + A * 17 + N)
    p = O.init_params(D, A, (H, H), (H, H), seed=D + A)
    p["log_std"] = rng.normal(-0.2, 0.2, A).astype(np.float32)
    buf, lv, dones = synthetic_rollout(T, N, D, A, seed=N + T)
    mean, val = O.policy_outputs(p, buf["obs"].reshape(T * N, D))
    buf["log_probs"] = (O.gaussian_log_prob(mean, p["log_std"], buf["actions"].reshape(T * N, A))
                        + rng.normal(0, 0.05, T * N)).astype(np.float32).reshape(T, N)
    buf["values"] = val.reshape(T, N)
    perms = np.stack([[rng.permutation(T * N) for _ in range(E)] for _ in range(2)])
    out = {}
    for epoch in (1, 0):
        e = PPOEngine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=B, n_epochs=E, pi=(H, H), vf=(H, H),
                      ent_coef=ent, normalize_advantage=normalize)
        e.set_hyper(epoch_kernel=epoch)
        e.set_params(p)
        e.load_rollout(buf, lv, dones)
        e.compute_gae()
        stats = [e.train(perms[0]), e.train(perms[1])]
        assert e.update_mode() == epoch
        m, v, step = e.get_optimizer_state()
        out[epoch] = (e.get_flat_params(), m, v, step, stats)
        e.close()
    (pa, ma, va, sa, sta), (pb, mb, vb, sb, stb) = out[1], out[0]
    assert sa == sb and np.array_equal(pa, pb, equal_nan=True)
    for k in ma:
        assert np.array_equal(ma[k], mb[k], equal_nan=True) and np.array_equal(va[k], vb[k], equal_nan=True), k
    for x, y in zip(sta, stb):
        assert all((x[k] == y[k]) or (np.isnan(x[k]) and np.isnan(y[k])) for k in x), (x, y)
