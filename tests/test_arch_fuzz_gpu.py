"""Seeded random corners of the generic chain's keyword surface -- observation widths 3 .. 150, 1 .. 40 actions, depths 1 .. 8 per network
with widths 8 .. 200, any of the twelve activations, gSDE with or without full_std / use_expln, minibatches that are no multiple of
anything -- each against the oracle: act (values, log-probs), every gradient tensor of a minibatch, the statistics of the step."""
import numpy as np
import pytest

from oracle import ppo_oracle as O
from tests.test_arch_gpu import _engine, _rollout_for
from tests.util import scaled_err

pytestmark = pytest.mark.gpu
SMOOTH = ["tanh", "elu", "sigmoid", "softplus", "softsign", "silu", "gelu", "mish"]   # (kinked ones: their own small-shape cases)


def _case(seed):
    r = np.random.default_rng(1000 + seed)
    D, A = int(r.integers(3, 151)), int(r.integers(1, 41))
    widths = lambda: tuple(int(8 * r.integers(1, 26)) for _ in range(int(r.integers(1, 9))))
    sde = bool(r.integers(0, 2))
    return dict(D=D, A=A, pi=widths(), vf=widths(), act=SMOOTH[int(r.integers(0, len(SMOOTH)))], sde=sde,
                full=bool(r.integers(0, 2)) if sde else True, expln=bool(r.integers(0, 2)) if sde else False,
                T=int(r.integers(3, 12)), N=int(r.integers(2, 40)))


@pytest.mark.parametrize("seed", range(16))
def test_random_configuration_matches_the_oracle(seed):
    c = _case(seed)
    D, A, T, N, pi, vf, act, sde = c["D"], c["A"], c["T"], c["N"], c["pi"], c["vf"], c["act"], c["sde"]
    B = T * N
    rng = np.random.default_rng(seed)
    p = O.init_params(D, A, pi, vf, seed=seed)
    p["log_std"] = (rng.normal(0.0 if c["expln"] else -1.0, 0.4, (pi[-1], A if c["full"] else 1)) if sde else rng.normal(-0.3, 0.2, A)).astype(np.float32)
    p["action_net.weight"] *= 10
    h = O.Hyper(ent_coef=0.01, n_epochs=1, batch_size=B, activation=act, use_sde=sde, sde_use_expln=c["expln"])
    # rollout data: _rollout_for knows the default gSDE options only -> build the gSDE log-probs here for the others
    buf, lv, dones = _rollout_for(p, act, D, A, T, N, rng, False) if not sde else (None, None, None)
    if sde:
        from tests.util import synthetic_rollout
        buf, lv, dones = synthetic_rollout(T, N, D, A, seed=5)
        flat = buf["obs"].reshape(B, D)
        mean, val = O.policy_outputs(p, flat, activation=act)
        sigma = O.sde_sigma(O.mlp_latents(p, flat, activation=act)[0][-1], p["log_std"], n_act=A, use_expln=c["expln"])
        acts = (mean + rng.standard_normal((B, A)).astype(np.float32) * sigma).astype(np.float32)
        lp = O.normal_log_prob(mean, sigma, acts)
        old = (lp + rng.normal(0, 0.1, B)).astype(np.float32)
        ratio = np.exp(lp.astype(np.float64) - old)
        old[np.minimum(np.abs(ratio - 1.2), np.abs(ratio - 0.8)) < 1e-4] += np.float32(0.01)
        buf["actions"], buf["log_probs"] = acts.reshape(T, N, A), old.reshape(T, N)
        buf["values"] = (val + rng.normal(0, 0.1, B)).astype(np.float32).reshape(T, N)
    buf["advantages"], buf["returns"] = O.gae(buf["rewards"], buf["values"], buf["episode_starts"], lv, dones, h.gamma, h.gae_lambda)
    e = _engine(D, A, N, T, pi, vf, batch_size=B, n_epochs=1, ent_coef=h.ent_coef, activation=act, use_sde=sde, sde_full_std=c["full"],
                sde_use_expln=c["expln"])
    e.set_params(p)
    obs0 = buf["obs"][0]
    if sde:
        z = rng.standard_normal((N, pi[-1], A)).astype(np.float32)
        e.sde_set_noise(z)
        a_raw, _, val0, lp0 = e.act(obs0)
        o_raw, _, o_val, o_lp = O.act_sde(p, obs0, O.sde_exploration_matrices(p["log_std"], z, c["expln"]), activation=act, use_expln=c["expln"])
    else:
        eps = rng.standard_normal((N, A)).astype(np.float32)
        a_raw, _, val0, lp0 = e.act(obs0, eps)
        o_raw, _, o_val, o_lp = O.act(p, obs0, eps, activation=act)
    assert scaled_err(a_raw, o_raw) < 1e-4 and scaled_err(val0, o_val) < 1e-4 and np.allclose(lp0, o_lp, rtol=1e-4, atol=1e-3), c
    e.rollout_begin()
    e.load_rollout(buf, lv, dones)
    perm = rng.permutation(B)
    e.epoch_begin(perm)
    e.minibatch_grad(0)
    got = e.unflatten(e.read("grads"))
    ostats, og, _ = O.loss_and_grads(p, *O.gather_minibatch(buf, perm), h)
    for k in og:
        assert got[k].shape == og[k].shape and scaled_err(got[k], og[k]) < 1e-4, (k, scaled_err(got[k], og[k]), c)
    e.minibatch_apply()
    stats = e.fetch_step_stats()[-1]
    for i, k in enumerate(["policy_loss", "value_loss", "entropy_loss", "loss", "approx_kl", "clip_fraction"]):
        assert abs(stats[i] - float(ostats[k])) < 2e-4 * max(1.0, abs(float(ostats[k]))), (k, stats[i], float(ostats[k]), c)
    e.close()
