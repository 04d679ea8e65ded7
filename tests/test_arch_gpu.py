"""`policy_kwargs` beyond the reference YAMLs' two tanh layers -- the other `activation_fn` modules and `net_arch` depths up to
eight (SB3 accepts any; the reference splats `ppo_kwargs` into PPO verbatim, /root/reference/src/mobrob/rl_control/ppo.py:58) --
through the generic GEMM chain, against (i) torch's own modules / autograd / clip_grad_norm_ / optim.Adam
(tests/golden/arch_cases.npz, made by tests/golden/make_arch_fixture.py) and (ii) the oracle, on every stage of the path:
act, one optimizer step, the rollout's time-limit bootstrap (per-row value evaluator), a whole train(), PPO(...) + the SB3 zip."""
import numpy as np
import pytest

from oracle import ppo_oracle as O
from tests.util import ARCH_CASES, arch_case, scaled_err, synthetic_rollout

pytestmark = pytest.mark.gpu


def _engine(D, A, N, T, pi, vf, **kw):
    from mobrob_amd.engine import PPOEngine
    return PPOEngine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, pi=pi, vf=vf, **kw)


@pytest.mark.parametrize("name", ARCH_CASES)
def test_act_and_one_step_match_torch_golden(name):
    c, act, pi, vf, p, h = arch_case(name)
    obs, eps = c["fwd/obs"], c["fwd/eps"]
    N, D, A = obs.shape[0], obs.shape[1], eps.shape[1]
    e = _engine(D, A, N, 4, pi, vf, batch_size=N, n_epochs=1, activation=act)
    assert list(e.shapes.keys()) == O.param_keys(len(pi), len(vf))
    assert e.x3_mode() == 0 or (act == "tanh" and len(pi) == 2 and len(vf) == 2)
    e.set_params(p)
    a_raw, a_clip, val, lp = e.act(obs, eps)
    o_raw, _, o_val, o_lp = O.act(p, obs, eps, activation=act)
    assert scaled_err(a_raw, c["fwd/actions"]) < 1e-4 and scaled_err(a_raw, o_raw) < 1e-4
    assert scaled_err(val, c["fwd/value"]) < 1e-4 and scaled_err(val, o_val) < 1e-4
    assert np.allclose(lp, c["fwd/log_prob"], rtol=1e-4, atol=1e-4) and np.allclose(lp, o_lp, rtol=1e-4, atol=1e-4)
    assert np.array_equal(a_clip, np.clip(a_raw, -1, 1))
    assert np.allclose(e.predict(obs, deterministic=True), np.clip(c["fwd/mean"], -1, 1), atol=1e-4)
    e.close()
    # one optimizer step on the golden minibatch (injected as a T = B, N = 1 rollout, identity permutation)
    mb_obs, mb_act = c["mb/obs"], c["mb/actions"]
    B = mb_obs.shape[0]
    e = _engine(D, A, 1, B, pi, vf, batch_size=B, n_epochs=1, activation=act, clip_range=h.clip_range, ent_coef=h.ent_coef,
                vf_coef=h.vf_coef, max_grad_norm=h.max_grad_norm, learning_rate=h.learning_rate, adam_eps=h.adam_eps)
    e.set_params(p)
    buf = dict(obs=mb_obs[:, None], actions=mb_act[:, None], rewards=np.zeros((B, 1), np.float32),
               episode_starts=np.zeros((B, 1), np.float32), values=c["mb/old_values"][:, None],
               log_probs=c["mb/old_log_prob"][:, None], advantages=c["mb/advantages"][:, None], returns=c["mb/returns"][:, None])
    e.load_rollout(buf, np.zeros(1, np.float32), np.zeros(1, bool))
    e.epoch_begin(np.arange(B))
    e.minibatch_grad(0)
    grads = e.unflatten(e.read("grads"))
    for k, v in grads.items():
        ref = c["step/grad/" + k]
        assert np.max(np.abs(v - ref)) < 1e-4 * max(1.0, float(np.max(np.abs(ref)))), (k, float(np.max(np.abs(v - ref))))
    e.minibatch_apply()
    stats = e.fetch_step_stats()[-1]
    for i, k in enumerate(["policy_loss", "value_loss", "entropy_loss", "loss", "approx_kl", "clip_fraction", "grad_norm"]):
        ref = float(c["step/" + k])
        assert abs(stats[i] - ref) < 1e-4 * max(1.0, abs(ref)), (k, stats[i], ref)
    newp = e.get_params()
    m, v, step = e.get_optimizer_state()
    assert step == 1
    for k in newp:
        assert np.max(np.abs(newp[k] - c["step/p/" + k])) < 1e-6 + 1e-5 * float(np.max(np.abs(c["step/p/" + k]))), k
        assert np.allclose(m[k], c["step/m/" + k], rtol=1e-3, atol=1e-6) and np.allclose(v[k], c["step/v/" + k], rtol=1e-3, atol=1e-8), k
    e.close()


TRAIN_CASES = [("elu", (64, 48), (32, 64)), ("leakyrelu", (32, 32, 32), (64,)), ("sigmoid", (64, 64), (64, 64)),
               ("softplus", (48,), (48, 24)), ("softsign", (64, 64), (64, 64)), ("hardtanh", (32, 64), (64, 32)),
               ("relu6", (64, 64), (64, 64)), ("tanh", (32, 32, 24, 24), (32, 24, 32, 24)), ("relu", (24,) * 6, (32, 16, 32, 16, 8)),
               ("elu", (16,) * 8, (16,) * 8), ("silu", (64, 48), (32, 64)), ("gelu", (32, 32, 32), (64,)), ("mish", (64, 64), (64, 64))]


@pytest.mark.parametrize("act,pi,vf", TRAIN_CASES)
def test_train_matches_oracle(act, pi, vf):
    """PPO.train over two epochs with supplied permutations (short last minibatch): every gradient tensor of the first minibatch,
    the logged statistics and the parameters after all optimizer steps."""
    D, A, T, N, B, E = 14, 2, 30, 7, 64, 2
    rng = np.random.default_rng(11)
    p = O.init_params(D, A, pi, vf, seed=2)
    p["log_std"] = rng.normal(-0.3, 0.2, A).astype(np.float32)
    p["action_net.weight"] *= 30
    buf, lv, dones = synthetic_rollout(T, N, D, A, seed=5)
    if act == "relu6":
        buf["obs"] *= 4
    mean, val = O.policy_outputs(p, buf["obs"].reshape(T * N, D), activation=act)
    acts = (mean + rng.standard_normal((T * N, A)).astype(np.float32) * np.exp(p["log_std"])).astype(np.float32)
    buf["actions"] = acts.reshape(T, N, A)
    buf["log_probs"] = (O.gaussian_log_prob(mean, p["log_std"], acts) + rng.normal(0, 0.1, T * N)).astype(np.float32).reshape(T, N)
    buf["values"] = (val + rng.normal(0, 0.1, T * N)).astype(np.float32).reshape(T, N)
    h = O.Hyper(gamma=0.99, gae_lambda=0.95, ent_coef=0.01, n_epochs=E, batch_size=B, learning_rate=3e-4, activation=act)
    buf["advantages"], buf["returns"] = O.gae(buf["rewards"], buf["values"], buf["episode_starts"], lv, dones, h.gamma, h.gae_lambda)
    perms = np.stack([rng.permutation(T * N) for _ in range(E)])
    e = _engine(D, A, N, T, pi, vf, batch_size=B, n_epochs=E, gamma=h.gamma, gae_lambda=h.gae_lambda, ent_coef=h.ent_coef,
                learning_rate=h.learning_rate, activation=act)
    e.set_params(p)
    e.load_rollout(buf, lv, dones)
    e.compute_gae()
    assert np.array_equal(e.read("advantages"), buf["advantages"])
    e.epoch_begin(perms[0])
    e.minibatch_grad(0)
    got = e.unflatten(e.read("grads"))
    _, og, _ = O.loss_and_grads(p, *O.gather_minibatch(buf, perms[0][:B]), h)
    for k in og:
        assert scaled_err(got[k], og[k]) < 1e-4, (k, scaled_err(got[k], og[k]))
    stats = e.train(perms)
    ostats = O.train(p, O.AdamState.zeros_like(p), buf, h, perms)
    nmb = -(-T * N // B)
    assert stats["n_minibatches"] == E * nmb == len(ostats)
    last = ostats[-nmb:]
    for k in ["policy_loss", "value_loss", "loss", "approx_kl", "clip_fraction", "grad_norm"]:
        ref = float(np.mean([float(s[k]) for s in last]))
        assert abs(stats[k] - ref) < 2e-4 * max(1.0, abs(ref)), (k, stats[k], ref)
    newp = e.get_params()
    for k in p:
        assert np.max(np.abs(newp[k] - p[k])) < 1e-4, (k, float(np.max(np.abs(newp[k] - p[k]))))
    e.close()


@pytest.mark.parametrize("act,pi,vf", [("elu", (32,), (32, 24, 16, 24)), ("softplus", (32, 32), (24,) * 5), ("tanh", (16,) * 4, (16,) * 8),
                                       ("gelu", (32, 32), (32, 16, 8))])
def test_host_rollout_with_bootstrap_matches_oracle(act, pi, vf):
    """act / store / finish_rollout == oracle collect_rollout on the same env stream; truncated rows' rewards take
    gamma * V(terminal_obs) from the per-row value evaluator (value_net_row: any depth, any activation)."""
    D, A, N, T = 14, 2, 6, 12
    p = O.init_params(D, A, pi, vf, seed=4)
    p["value_net.bias"] = np.array([3.0], np.float32)
    rng = np.random.default_rng(0)
    eps = rng.standard_normal((T, N, A)).astype(np.float32)
    h = O.Hyper(gamma=0.99, gae_lambda=0.9, activation=act)
    env_a = O.NumpySyntheticVecEnv(N, D, A, p_term=0.1, time_limit=5, seed=3)
    obuf, _, _ = O.collect_rollout({k: v.copy() for k, v in p.items()}, env_a, env_a.reset(), np.ones(N, bool), T, h, lambda t: eps[t])
    e = _engine(D, A, N, T, pi, vf, batch_size=8, n_epochs=1, gamma=h.gamma, gae_lambda=h.gae_lambda, activation=act)
    e.set_params(p)
    env_b = O.NumpySyntheticVecEnv(N, D, A, p_term=0.1, time_limit=5, seed=3)
    obs = env_b.reset()
    e.rollout_begin()
    saw_trunc = False
    for t in range(T):
        _, a_clip, _, _ = e.act(obs, eps[t])
        obs, rew, done, trunc, term_obs = env_b.step(a_clip)
        saw_trunc |= bool(trunc.any())
        e.store(rew, done, trunc, term_obs)
    e.finish_rollout(obs, done)
    assert saw_trunc
    for k in ["actions", "rewards", "values", "log_probs", "advantages", "returns"]:
        assert scaled_err(e.read(k), obuf[k]) < 1e-4, k
    e.close()


@pytest.mark.parametrize("act,pi,vf", [("leakyrelu", (32, 24, 16, 8), (32, 24, 16, 8)), ("sigmoid", (32,), (32, 32))])
@pytest.mark.parametrize("kind", ["synthetic", "goal"])
def test_device_rollout_bootstrap(act, pi, vf, kind):
    """Device env sources (per-step kernels of the generic path): truncated rows' rewards carry gamma * V(terminal_obs)."""
    D, A, N, T, TL = 26, 2, 96, 20, 10
    p = O.init_params(D, A, pi, vf, seed=2)
    p["value_net.bias"] = np.array([7.0], np.float32)
    e = _engine(D, A, N, T, pi, vf, batch_size=480, n_epochs=1, seed=9, activation=act)
    e.set_params(p)
    if kind == "synthetic":
        e.collect_synthetic(p_term=0.0, time_limit=TL)
    else:
        e.collect_goal_env(pos_dim=2, mix=np.eye(2, A, dtype=np.float32), time_limit=TL, terminate_on_goal=False)
    e.synchronize()
    tr = e.read("truncated").astype(bool)
    assert tr.any()
    tobs = e.read("terminal_obs")[:, :D]
    _, v = O.policy_outputs(p, tobs, activation=act)
    assert scaled_err(e.read("terminal_values")[tr], v[tr]) < 1e-4
    obs, acts = e.read("obs")[:T].reshape(T * N, -1)[:, :D], e.read("actions").reshape(T * N, A)
    mean, val = O.policy_outputs(p, obs, activation=act)
    assert scaled_err(e.read("values").reshape(-1), val) < 1e-4
    assert np.allclose(e.read("log_probs").reshape(-1), O.gaussian_log_prob(mean, p["log_std"], acts), rtol=1e-4, atol=1e-3)
    e.close()


@pytest.mark.parametrize("kwargs", [dict(net_arch=[32, 32, 32, 32], activation_fn="ELU"),
                                    dict(net_arch=dict(pi=[32], vf=[32, 24, 16, 8, 8]), activation_fn="LeakyReLU", share_features_extractor=False),
                                    dict(net_arch=[64, 64], activation_fn="Softsign", normalize_images=True),
                                    dict(net_arch=[32, 32], activation_fn="SiLU")])
def test_ppo_learns_saves_and_loads(kwargs, tmp_path):
    """PPO(...) with these policy_kwargs (activation classes are given by torch.nn class where torch is importable): learns on the
    device goal env, writes an SB3-layout zip (state-dict keys in nn.Sequential numbering, policy_kwargs as SB3's one-blob form with
    the class by reference), loads back bit-identical with the same activation."""
    import torch
    from mobrob_amd import checkpoint as ck
    from mobrob_amd.rl_control.ppo import PPOCtrl, PPO
    pk = dict(kwargs, activation_fn=getattr(torch.nn, kwargs["activation_fn"]))
    cfg = {"ppo_kwargs": {"policy": "MlpPolicy", "n_steps": 32, "batch_size": 256, "n_epochs": 2, "policy_kwargs": pk},
           "env_name": "point", "time_limit": 50, "n_envs": 64, "vec_env_type": "device_goal", "enable_gui": False, "seed": 0}
    ctrl = PPOCtrl.from_config(cfg)
    ctrl.ppo.learn(total_timesteps=3 * 32 * 64)
    path = str(tmp_path / "m.zip")
    ctrl.ppo.save(path)
    z = ck.load_zip(path)
    arch = kwargs["net_arch"]
    pi = arch["pi"] if isinstance(arch, dict) else arch
    vf = arch["vf"] if isinstance(arch, dict) else arch
    assert list(z["params"].keys()) == ck.policy_keys(len(pi), len(vf)) == O.param_keys(len(pi), len(vf))
    assert z["data"]["policy_kwargs"]["activation_fn"] == kwargs["activation_fn"]
    for k in ("share_features_extractor", "normalize_images"):
        if k in kwargs:
            assert z["data"]["policy_kwargs"][k] == kwargs[k]
    back = PPO.load(path)
    assert back.activation == ctrl.ppo.activation == kwargs["activation_fn"].lower() and back.engine.cfg.activation == ctrl.ppo.engine.cfg.activation
    a, b = ctrl.ppo.engine.get_flat_params(), back.engine.get_flat_params()
    assert np.array_equal(a, b) and np.isfinite(a).all()
    assert back.net_arch == (tuple(pi), tuple(vf))
    obs = np.random.default_rng(0).standard_normal((5, ctrl.ppo.obs_dim)).astype(np.float32)
    assert np.array_equal(ctrl.ppo.predict(obs, deterministic=True)[0], back.predict(obs, deterministic=True)[0])
    m, _ = O.policy_outputs(back.engine.get_params(), obs, activation=back.activation)
    assert np.allclose(back.predict(obs, deterministic=True)[0], np.clip(m, -1, 1), atol=1e-4)


def test_unsupported_policy_kwargs_are_refused_by_name():
    from mobrob_amd.rl_control.ppo import PPO
    for pk, word in [(dict(activation_fn="PReLU"), "PReLU"), (dict(activation_fn="Softmax"), "Softmax"), (dict(net_arch=[8] * 9), "hidden"),
                     (dict(features_extractor_class="NatureCNN"), "features_extractor_class")]:
        with pytest.raises((NotImplementedError, ValueError), match=word):
            PPO("MlpPolicy", None, policy_kwargs=pk, _dims=(4, 6, 2))


def _rollout_for(p, act, D, A, T, N, rng, sde):
    buf, lv, dones = synthetic_rollout(T, N, D, A, seed=5)
    flat = buf["obs"].reshape(T * N, D)
    mean, val = O.policy_outputs(p, flat, activation=act)
    if sde:
        sigma = O.sde_sigma(O.mlp_latents(p, flat, activation=act)[0][-1], p["log_std"])
        acts = (mean + rng.standard_normal((T * N, A)).astype(np.float32) * sigma).astype(np.float32)
        lp = O.normal_log_prob(mean, sigma, acts)
    else:
        acts = (mean + rng.standard_normal((T * N, A)).astype(np.float32) * np.exp(p["log_std"])).astype(np.float32)
        lp = O.gaussian_log_prob(mean, p["log_std"], acts)
    buf["actions"] = acts.reshape(T, N, A)
    old = (lp + rng.normal(0, 0.1, T * N)).astype(np.float32)
    # The clipped surrogate is discontinuous in the ratio at 1 +- clip_range: a row whose ratio sits within float32 rounding of a
    # boundary gets its whole gradient from one implementation and none from the other (measured: one such row in 65 536 moved
    # every policy tensor by 3e-4 of its scale).  Such rows are moved off the boundary -- parity is about arithmetic, not ties.
    ratio = np.exp(lp.astype(np.float64) - old)
    near = np.minimum(np.abs(ratio - 1.2), np.abs(ratio - 0.8)) < 1e-4
    old[near] += np.float32(0.01)
    buf["log_probs"] = old.reshape(T, N)
    buf["values"] = (val + rng.normal(0, 0.1, T * N)).astype(np.float32).reshape(T, N)
    return buf, lv, dones


@pytest.mark.parametrize("tiles", ["1", "21", "22"])
@pytest.mark.parametrize("act,pi,vf,sde", [("tanh", (72, 40), (136,), False), ("elu", (40, 104, 72), (64, 64), False),
                                           ("gelu", (96,), (40, 72), False), ("relu", (96,), (40, 72), True),
                                           ("tanh", (128, 256), (256,), False)])   # (column tiles in fours: the waves-along-N orientation)
def test_every_gemm_tiling_matches_the_oracle(tiles, act, pi, vf, sde, monkeypatch):
    """The generic chain's GEMM picks 1x1, 2x1 or 2x2 tiles of 32x32 per wave by launch size (engine.hip launch_gemm), which small
    test shapes never reach: MOBROB_GEMM_TILES forces each form through shapes that are no multiple of its tile (rows 190 and 20,
    widths 40 / 72 / 104 / 136, 27 observation columns, 5 actions) -- first minibatch's gradients, statistics and parameters after
    two epochs against the oracle; the three forms agree with each other to rounding."""
    monkeypatch.setenv("MOBROB_GEMM_TILES", tiles)
    D, A, T, N, B, E = 27, 5, 30, 7, 190, 2
    rng = np.random.default_rng(11)
    p = O.init_params(D, A, pi, vf, seed=2)
    p["log_std"] = (rng.normal(-1.5, 0.3, (pi[-1], A)) if sde else rng.normal(-0.3, 0.2, A)).astype(np.float32)
    p["action_net.weight"] *= 30
    buf, lv, dones = _rollout_for(p, act, D, A, T, N, rng, sde)
    h = O.Hyper(gamma=0.99, gae_lambda=0.95, ent_coef=0.01, n_epochs=E, batch_size=B, learning_rate=3e-4, activation=act, use_sde=sde)
    buf["advantages"], buf["returns"] = O.gae(buf["rewards"], buf["values"], buf["episode_starts"], lv, dones, h.gamma, h.gae_lambda)
    perms = np.stack([rng.permutation(T * N) for _ in range(E)])
    e = _engine(D, A, N, T, pi, vf, batch_size=B, n_epochs=E, gamma=h.gamma, gae_lambda=h.gae_lambda, ent_coef=h.ent_coef,
                learning_rate=h.learning_rate, activation=act, use_sde=sde)
    e.set_params(p)
    e.load_rollout(buf, lv, dones)
    e.compute_gae()
    e.epoch_begin(perms[0])
    e.minibatch_grad(0)
    got = e.unflatten(e.read("grads"))
    _, og, _ = O.loss_and_grads(p, *O.gather_minibatch(buf, perms[0][:B]), h)
    for k in og:
        assert scaled_err(got[k], og[k]) < 1e-4, (k, scaled_err(got[k], og[k]))
    stats = e.train(perms)
    ostats = O.train(p, O.AdamState.zeros_like(p), buf, h, perms)
    nmb = -(-T * N // B)
    for k in ["policy_loss", "value_loss", "entropy_loss", "loss", "approx_kl", "clip_fraction", "grad_norm"]:
        ref = float(np.mean([float(s[k]) for s in ostats[-nmb:]]))
        assert abs(stats[k] - ref) < 2e-4 * max(1.0, abs(ref)), (k, stats[k], ref)
    newp = e.get_params()
    for k in p:
        assert np.max(np.abs(newp[k] - p[k])) < 1e-4, (k, float(np.max(np.abs(newp[k] - p[k]))))
    # rollout-time forward (NT launches over N rows) and the value pass through the same forced tiling
    obs = rng.standard_normal((N, D)).astype(np.float32)
    assert np.allclose(e.predict(obs, deterministic=True), np.clip(O.policy_outputs(p, obs, activation=act)[0], -1, 1), atol=1e-4)
    e.close()


@pytest.mark.parametrize("act,sde,pi,vf", [("tanh", False, (128, 128), (128, 96)), ("silu", False, (128, 128), (128, 96)),
                                           ("tanh", True, (128, 128), (128, 96)), ("elu", False, (128, 128, 64), (96,)),
                                           ("elu", False, (256, 256), (256, 256))])
def test_full_size_minibatch_on_the_generic_chain(act, sde, pi, vf):
    """65 536 rows through the launch-size tile selection as shipped (2x2 tiles for the hidden layers, 2x1 for the heads, batch
    split with float atomics for the weight gradients; the two networks' GEMMs paired per launch, with unequal depths the tails
    and a head opposite a hidden layer alone) at widths no fused family covers: every gradient tensor of one minibatch against the
    float64-accumulating oracle.  (Smooth activations only: with 8 M pre-activations a handful sit within rounding of ReLU's kink, and
    float32 against float64 accumulation then flips their derivative -- 1e-3 of a first-layer gradient, measured.)"""
    D, A, T, N = 26, 3, 64, 1024
    B = T * N
    rng = np.random.default_rng(3)
    p = O.init_params(D, A, pi, vf, seed=7)
    p["log_std"] = (rng.normal(-1.5, 0.3, (pi[-1], A)) if sde else rng.normal(-0.3, 0.2, A)).astype(np.float32)
    p["action_net.weight"] *= 30
    buf, lv, dones = _rollout_for(p, act, D, A, T, N, rng, sde)
    h = O.Hyper(ent_coef=0.01, n_epochs=1, batch_size=B, activation=act, use_sde=sde)
    buf["advantages"], buf["returns"] = O.gae(buf["rewards"], buf["values"], buf["episode_starts"], lv, dones, h.gamma, h.gae_lambda)
    e = _engine(D, A, N, T, pi, vf, batch_size=B, n_epochs=1, ent_coef=h.ent_coef, activation=act, use_sde=sde)
    assert e.x3_mode() == 0
    e.set_params(p)
    e.load_rollout(buf, lv, dones)
    perm = rng.permutation(B)
    e.epoch_begin(perm)
    e.minibatch_grad(0)
    got = e.unflatten(e.read("grads"))
    ostats, og, _ = O.loss_and_grads(p, *O.gather_minibatch(buf, perm), h, acc=np.float64)
    for k in og:
        assert scaled_err(got[k], og[k]) < 1e-4, (k, scaled_err(got[k], og[k]))
    e.minibatch_apply()
    stats = e.fetch_step_stats()[-1]
    for i, k in enumerate(["policy_loss", "value_loss", "entropy_loss", "loss", "approx_kl", "clip_fraction"]):
        assert abs(stats[i] - float(ostats[k])) < 1e-4 * max(1.0, abs(float(ostats[k]))), (k, stats[i], float(ostats[k]))
    e.close()
