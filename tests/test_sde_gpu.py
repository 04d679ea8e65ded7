"""PPO(use_sde=True) -- generalised state-dependent exploration (SB3 StateDependentNoiseDistribution with its defaults; the reference
splats `ppo_kwargs` into PPO verbatim, /root/reference/src/mobrob/rl_control/ppo.py:58, and its README points at SB3's PPO for "all
supported parameters") -- through the generic GEMM chain, against torch's own ops (tests/golden/sde_cases.npz) and the oracle:
sampling with supplied exploration matrices, one optimizer step (the [HL][A] log_std gradient through the variance), a whole train(),
a host rollout with sde_sample_freq, the engine's own draws, PPO(...) + the SB3 zip."""
import numpy as np
import pytest

from oracle import ppo_oracle as O
from tests.util import SDE_CASES, scaled_err, sde_case, synthetic_rollout

pytestmark = pytest.mark.gpu


def _engine(D, A, N, T, pi, vf, **kw):
    from mobrob_amd.engine import PPOEngine
    return PPOEngine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, pi=pi, vf=vf, use_sde=True, **kw)


@pytest.mark.parametrize("name", SDE_CASES)
def test_act_and_one_step_match_torch_golden(name):
    c, act, pi, vf, p, h = sde_case(name)
    obs, z = c["fwd/obs"], c["fwd/z"]
    N, D, A = obs.shape[0], obs.shape[1], z.shape[2]
    opts = dict(sde_full_std=bool(c["full_std"]), sde_use_expln=bool(c["use_expln"]))
    e = _engine(D, A, N, 4, pi, vf, batch_size=N, n_epochs=1, activation=act, **opts)
    assert e.shapes["log_std"] == (pi[-1], A if opts["sde_full_std"] else 1) and list(e.shapes.keys()) == O.param_keys(len(pi), len(vf))
    assert e.x3_mode() == 0
    e.set_params(p)
    e.sde_set_noise(z)
    theta = O.sde_exploration_matrices(p["log_std"], z, h.sde_use_expln)
    assert scaled_err(e.read("sde_noise"), theta) < 1e-6
    a_raw, a_clip, val, lp = e.act(obs)
    o_raw, _, o_val, o_lp = O.act_sde(p, obs, theta, activation=act, use_expln=h.sde_use_expln)
    assert scaled_err(a_raw, c["fwd/actions"]) < 1e-4 and scaled_err(a_raw, o_raw) < 1e-4
    assert scaled_err(val, c["fwd/value"]) < 1e-4 and scaled_err(val, o_val) < 1e-4
    assert np.allclose(lp, c["fwd/log_prob"], rtol=1e-4, atol=1e-4) and np.allclose(lp, o_lp, rtol=1e-4, atol=1e-4)
    assert np.array_equal(a_clip, np.clip(a_raw, -1, 1))
    assert np.allclose(e.predict(obs, deterministic=True), np.clip(c["fwd/mean"], -1, 1), atol=1e-4)
    # predict(deterministic=False): a batch of n_envs rows takes the environments' matrices, any other batch the single one (= env 0's here)
    assert scaled_err(e.predict(obs, deterministic=False), np.clip(c["fwd/actions"], -1, 1)) < 1e-4
    assert scaled_err(e.predict(obs[:7], deterministic=False), np.clip(c["fwd/single"][:7], -1, 1)) < 1e-4
    e.close()
    mb_obs, mb_act = c["mb/obs"], c["mb/actions"]
    B = mb_obs.shape[0]
    e = _engine(D, A, 1, B, pi, vf, batch_size=B, n_epochs=1, activation=act, clip_range=h.clip_range, ent_coef=h.ent_coef,
                vf_coef=h.vf_coef, max_grad_norm=h.max_grad_norm, learning_rate=h.learning_rate, adam_eps=h.adam_eps, **opts)
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
        assert v.shape == ref.shape and np.max(np.abs(v - ref)) < 1e-4 * max(1.0, float(np.max(np.abs(ref)))), (k, float(np.max(np.abs(v - ref))))
    assert scaled_err(grads["log_std"], c["step/grad/log_std"]) < 1e-4    # relative to the matrix's own scale (its entries are ~1e-2)
    e.minibatch_apply()
    stats = e.fetch_step_stats()[-1]
    for i, k in enumerate(["policy_loss", "value_loss", "entropy_loss", "loss", "approx_kl", "clip_fraction", "grad_norm"]):
        ref = float(c["step/" + k])
        assert abs(stats[i] - ref) < 1e-4 * max(1.0, abs(ref)), (k, stats[i], ref)
    newp = e.get_params()
    for k in newp:
        assert np.max(np.abs(newp[k] - c["step/p/" + k])) < 1e-6 + 1e-5 * float(np.max(np.abs(c["step/p/" + k]))), k
    e.close()


@pytest.mark.parametrize("act,pi,vf,full,expln", [("tanh", (64, 64), (64, 64), True, False), ("relu", (32,), (48, 24), True, False),
                                                  ("softsign", (32, 24, 40), (32,), True, False), ("tanh", (32, 32), (32,), False, False),
                                                  ("elu", (48,), (32, 32), True, True), ("tanh", (24, 24), (24,), False, True)])
def test_train_matches_oracle(act, pi, vf, full, expln):
    D, A, T, N, B, E = 14, 2, 30, 7, 64, 2
    rng = np.random.default_rng(11)
    p = O.init_params(D, A, pi, vf, seed=2)
    p["log_std"] = rng.normal(0.0 if expln else -1.5, 0.5 if expln else 0.3, (pi[-1], A if full else 1)).astype(np.float32)
    p["action_net.weight"] *= 30
    buf, lv, dones = synthetic_rollout(T, N, D, A, seed=5)
    h = O.Hyper(gamma=0.99, gae_lambda=0.95, ent_coef=0.01, n_epochs=E, batch_size=B, learning_rate=3e-4, activation=act, use_sde=True,
                sde_use_expln=expln)
    acts_pi, _ = O.mlp_latents(p, buf["obs"].reshape(T * N, D), activation=act)
    mean, val = O.policy_outputs(p, buf["obs"].reshape(T * N, D), activation=act)
    sigma = O.sde_sigma(acts_pi[-1], p["log_std"], n_act=A, use_expln=expln)
    acts = (mean + rng.standard_normal((T * N, A)).astype(np.float32) * sigma).astype(np.float32)
    buf["actions"] = acts.reshape(T, N, A)
    buf["log_probs"] = (O.normal_log_prob(mean, sigma, acts) + rng.normal(0, 0.1, T * N)).astype(np.float32).reshape(T, N)
    buf["values"] = (val + rng.normal(0, 0.1, T * N)).astype(np.float32).reshape(T, N)
    buf["advantages"], buf["returns"] = O.gae(buf["rewards"], buf["values"], buf["episode_starts"], lv, dones, h.gamma, h.gae_lambda)
    perms = np.stack([rng.permutation(T * N) for _ in range(E)])
    e = _engine(D, A, N, T, pi, vf, batch_size=B, n_epochs=E, gamma=h.gamma, gae_lambda=h.gae_lambda, ent_coef=h.ent_coef,
                learning_rate=h.learning_rate, activation=act, sde_full_std=full, sde_use_expln=expln)
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
    last = ostats[-nmb:]
    for k in ["policy_loss", "value_loss", "entropy_loss", "loss", "approx_kl", "clip_fraction", "grad_norm"]:
        ref = float(np.mean([float(s[k]) for s in last]))
        assert abs(stats[k] - ref) < 2e-4 * max(1.0, abs(ref)), (k, stats[k], ref)
    newp = e.get_params()
    for k in p:
        assert np.max(np.abs(newp[k] - p[k])) < 1e-4, (k, float(np.max(np.abs(newp[k] - p[k]))))
    e.close()


@pytest.mark.parametrize("freq", [-1, 4])
def test_host_rollout_with_supplied_noise_matches_oracle(freq):
    """act / store / finish_rollout == oracle collect_rollout with the SAME exploration matrices, redrawn every sde_sample_freq
    steps (supplied at exactly the steps SB3 calls reset_noise); time-limit truncations take the value bootstrap."""
    D, A, N, T, pi, vf = 14, 2, 6, 12, (32, 24), (24,)
    p = O.init_params(D, A, pi, vf, seed=4)
    p["log_std"] = np.full((pi[-1], A), -1.0, np.float32)
    p["value_net.bias"] = np.array([3.0], np.float32)
    rng = np.random.default_rng(0)
    zs = rng.standard_normal((T, N, pi[-1], A)).astype(np.float32)
    h = O.Hyper(gamma=0.99, gae_lambda=0.9, use_sde=True, sde_sample_freq=freq)
    env_a = O.NumpySyntheticVecEnv(N, D, A, p_term=0.1, time_limit=5, seed=3)
    obuf, _, _ = O.collect_rollout({k: v.copy() for k, v in p.items()}, env_a, env_a.reset(), np.ones(N, bool), T, h, lambda t: zs[t])
    e = _engine(D, A, N, T, pi, vf, batch_size=8, n_epochs=1, gamma=h.gamma, gae_lambda=h.gae_lambda, sde_sample_freq=freq)
    e.set_params(p)
    env_b = O.NumpySyntheticVecEnv(N, D, A, p_term=0.1, time_limit=5, seed=3)
    obs = env_b.reset()
    e.rollout_begin()
    for t in range(T):
        if t == 0 or (freq > 0 and t % freq == 0):
            e.sde_set_noise(zs[t])
        _, a_clip, _, _ = e.act(obs)
        obs, rew, done, trunc, term_obs = env_b.step(a_clip)
        e.store(rew, done, trunc, term_obs)
    e.finish_rollout(obs, done)
    for k in ["actions", "rewards", "values", "log_probs", "advantages", "returns"]:
        assert scaled_err(e.read(k), obuf[k]) < 1e-4, k
    e.close()


@pytest.mark.parametrize("freq,kind", [(-1, "host"), (5, "host"), (-1, "synthetic"), (10, "goal")])
def test_own_draws_follow_the_schedule(freq, kind):
    """The engine's own exploration matrices: N(0, exp(log_std)^2) entries, constant between reset_noise points and new at them
    (start of a rollout, every sde_sample_freq steps); stored actions = mean + latent . theta, stored log-probs = the distribution's."""
    D, A, N, T, pi, vf = 14, 2, 64, 20, (32, 8), (32, 32)   # (HL = 8 < the stretches' lengths: one theta per stretch is a real constraint)
    p = O.init_params(D, A, pi, vf, seed=1)
    p["log_std"] = np.random.default_rng(2).normal(-1.0, 0.4, (pi[-1], A)).astype(np.float32)
    e = _engine(D, A, N, T, pi, vf, batch_size=N * T, n_epochs=1, sde_sample_freq=freq, seed=5)
    e.set_params(p)
    thetas = []
    if kind == "host":
        env = O.NumpySyntheticVecEnv(N, D, A, p_term=0.05, time_limit=9, seed=3)
        obs = env.reset()
        e.rollout_begin()
        for t in range(T):
            _, a_clip, _, _ = e.act(obs)
            thetas.append(e.read("sde_noise"))
            obs, rew, done, trunc, term_obs = env.step(a_clip)
            e.store(rew, done, trunc, term_obs)
        e.finish_rollout(obs, done)
    elif kind == "synthetic":
        e.collect_synthetic(p_term=0.02, time_limit=9)
    else:
        e.collect_goal_env(pos_dim=2, mix=np.eye(2, A, dtype=np.float32), time_limit=9, terminate_on_goal=False)
    e.synchronize()
    obs_all, acts_all = e.read("obs")[:T], e.read("actions")
    lat = [O.mlp_latents(p, obs_all[t])[0][-1] for t in range(T)]
    mean = [O.policy_outputs(p, obs_all[t])[0] for t in range(T)]
    for t in range(T):   # log-probs are the state-dependent distribution's at the stored actions
        lp = O.normal_log_prob(mean[t], O.sde_sigma(lat[t], p["log_std"]), acts_all[t])
        assert np.allclose(e.read("log_probs")[t], lp, rtol=1e-4, atol=1e-3), t
    if kind == "host":
        std = np.exp(p["log_std"])
        for t in range(T):
            fresh = t == 0 or (freq > 0 and t % freq == 0)
            assert fresh != np.array_equal(thetas[t], thetas[t - 1]) if t > 0 else True, t
            noise = np.einsum("nk,nka->na", lat[t], thetas[t])
            assert scaled_err(acts_all[t], mean[t] + noise) < 1e-4, t
        zz = thetas[0] / std
        assert abs(float(zz.mean())) < 0.12 and abs(float(zz.std()) - 1.0) < 0.1      # 1024 draws: four standard errors
        zf = zz.reshape(N, -1)                                             # environments draw independently
        assert len(np.unique(zf.round(6), axis=0)) == N and abs(float(np.mean(np.sum(zf[:-1] * zf[1:], axis=1)) / zf.shape[1])) < 0.15
    else:
        # device rollouts: between reset points the noise of an env is ONE linear map of its latent -- a single theta_n (least
        # squares over the stretch, more steps than HL) explains the whole stretch; a window across a reset point has no such theta
        def residual(n, t0, t1):
            Lm = np.stack([lat[t][n] for t in range(t0, t1)]).astype(np.float64)                    # [steps, HL]
            Y = np.stack([acts_all[t][n] - mean[t][n] for t in range(t0, t1)]).astype(np.float64)   # [steps, A]
            theta, *_ = np.linalg.lstsq(Lm, Y, rcond=None)
            return float(np.max(np.abs(Lm @ theta - Y))) / max(1e-6, float(np.max(np.abs(Y))))
        seg = T if freq <= 0 else freq
        assert seg > pi[-1]
        for n in range(0, N, 8):
            assert residual(n, 0, seg) < 1e-3, (n, residual(n, 0, seg))
            if freq > 0:
                assert residual(n, seg, 2 * seg) < 1e-3 and residual(n, seg // 2, seg // 2 + seg) > 1e-2, n
    e.close()


@pytest.mark.parametrize("vec_env_type", ["device_goal", "native", "dummy"])
def test_ppo_with_sde_learns_saves_and_loads(vec_env_type, tmp_path):
    """PPO(use_sde=True) through PPOCtrl on the device env, the native C host env (pipelined row ranges: each range redraws its own
    environments' matrices) and Python envs stepped in process."""
    from mobrob_amd import checkpoint as ck
    from mobrob_amd.rl_control.ppo import PPOCtrl, PPO
    cfg = {"ppo_kwargs": {"policy": "MlpPolicy", "n_steps": 32, "batch_size": 256, "n_epochs": 2, "use_sde": True, "sde_sample_freq": 4,
                          "policy_kwargs": {"net_arch": [32, 32], "log_std_init": -2.0}},
           "env_name": "point", "time_limit": 50, "n_envs": 64 if vec_env_type != "dummy" else 8, "vec_env_type": vec_env_type,
           "enable_gui": False, "seed": 0}
    ctrl = PPOCtrl.from_config(cfg)
    ppo = ctrl.ppo
    assert ppo.use_sde and ppo.sde_sample_freq == 4 and ppo.engine.get_params()["log_std"].shape == (32, ppo.act_dim)
    assert np.all(ppo.engine.get_params()["log_std"] == -2.0)
    before = ppo.engine.get_flat_params()
    ppo.learn(total_timesteps=3 * 32 * ppo.n_envs)
    after = ppo.engine.get_flat_params()
    assert np.isfinite(after).all() and not np.array_equal(before, after)
    assert not np.array_equal(ppo.engine.get_params()["log_std"], np.full((32, ppo.act_dim), -2.0, np.float32))   # the matrix is trained
    # the rollout the last update was computed on: stored log-probs are the state-dependent distribution's at the stored actions
    # (under the parameters BEFORE that update, so only finiteness and the clipped actions' range are checked here)
    assert np.isfinite(ppo.engine.read("log_probs")).all() and np.isfinite(ppo.engine.read("actions")).all()
    path = str(tmp_path / "sde.zip")
    ppo.save(path)
    z = ck.load_zip(path)
    assert z["data"]["use_sde"] is True and z["data"]["sde_sample_freq"] == 4 and z["params"]["log_std"].shape == (32, ppo.act_dim)
    back = PPO.load(path)
    assert back.use_sde and back.sde_sample_freq == 4 and np.array_equal(back.engine.get_flat_params(), after)
    obs = np.random.default_rng(0).standard_normal((5, ppo.obs_dim)).astype(np.float32)
    assert np.array_equal(ppo.predict(obs, deterministic=True)[0], back.predict(obs, deterministic=True)[0])
    noisy = back.predict(obs, deterministic=False)[0]
    assert noisy.shape == (5, ppo.act_dim) and not np.array_equal(noisy, back.predict(obs, deterministic=True)[0])
    with pytest.raises(NotImplementedError):
        PPO("MlpPolicy", None, use_sde=True, policy_kwargs=dict(squash_output=True), _dims=(4, 6, 2))


def test_ppo_passes_full_std_and_use_expln_through(tmp_path):
    from mobrob_amd.rl_control.ppo import PPO
    from mobrob_amd.envs.vec_env import DeviceGoalVecEnv
    env = DeviceGoalVecEnv.for_robot("point", 32, time_limit=50)
    ppo = PPO("MlpPolicy", env, n_steps=16, batch_size=128, n_epochs=1, use_sde=True, seed=1,
              policy_kwargs=dict(net_arch=[32, 16], full_std=False, use_expln=True, log_std_init=0.5))
    assert ppo.engine.get_params()["log_std"].shape == (16, 1) and ppo.engine.cfg.sde_full_std == 0 and ppo.engine.cfg.sde_use_expln == 1
    ppo.learn(total_timesteps=2 * 16 * 32)
    path = str(tmp_path / "opts.zip")
    ppo.save(path)
    back = PPO.load(path)
    assert back.use_sde and not back.sde_full_std and back.sde_use_expln and np.array_equal(back.engine.get_flat_params(), ppo.engine.get_flat_params())


def test_supplied_noise_holds_through_a_device_rollout_captured_before_it():
    """A device rollout is replayed from a captured graph with the resampling launches baked in; supplying the matrices afterwards
    must take effect (the graph is captured again): every step of the next rollout uses exactly the supplied theta, and releasing
    the hold brings the engine's own draws back."""
    D, A, N, T, pi, vf = 14, 2, 32, 8, (32, 8), (32,)
    p = O.init_params(D, A, pi, vf, seed=1)
    p["log_std"] = np.full((pi[-1], A), -1.0, np.float32)
    e = _engine(D, A, N, T, pi, vf, batch_size=N * T, n_epochs=1, sde_sample_freq=2, seed=5)
    e.set_params(p)
    e.collect_synthetic(p_term=0.02, time_limit=9)          # captures the graph, own draws every 2 steps
    e.synchronize()
    z = np.random.default_rng(3).standard_normal((N, pi[-1], A)).astype(np.float32)
    theta = O.sde_exploration_matrices(p["log_std"], z)
    e.sde_set_noise(z)
    e.collect_synthetic(p_term=0.02, time_limit=9)
    e.synchronize()
    obs_all, acts_all = e.read("obs")[:T], e.read("actions")
    for t in range(T):
        lat, mean = O.mlp_latents(p, obs_all[t])[0][-1], O.policy_outputs(p, obs_all[t])[0]
        assert scaled_err(acts_all[t], mean + np.einsum("nk,nka->na", lat, theta)) < 1e-4, t
    assert np.array_equal(e.read("sde_noise"), theta.astype(np.float32)) or scaled_err(e.read("sde_noise"), theta) < 1e-6
    e.sde_set_noise(None)
    e.collect_synthetic(p_term=0.02, time_limit=9)
    e.synchronize()
    assert scaled_err(e.read("sde_noise"), theta) > 0.1
    e.close()
