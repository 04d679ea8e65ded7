"""`net_arch` depths other than the reference's two hidden layers (SB3 accepts any list: the reference splats `ppo_kwargs` into PPO
verbatim, /root/reference/src/mobrob/rl_control/ppo.py:58): one and three hidden layers per network, also mixed, through the
generic GEMM chain -- against the oracle (whose MLPs are depth-generic, oracle/ppo_oracle.py mlp_latents) on every stage of the path:
act, the host rollout incl. the time-limit bootstrap, the device rollout's bootstrap, a whole train() and the SB3 zip."""
import numpy as np
import pytest

from oracle import ppo_oracle as O
from tests.util import scaled_err, synthetic_rollout

pytestmark = pytest.mark.gpu

ARCHS = [((64,), (64,)), ((64, 48, 32), (64, 48, 32)), ((40,), (64, 32, 16)), ((32, 64, 24), (48,)), ((256,), (256, 256, 256))]


def _engine(D, A, N, T, pi, vf, **kw):
    from mobrob_amd.engine import PPOEngine
    return PPOEngine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, pi=pi, vf=vf, **kw)


@pytest.mark.parametrize("pi,vf", ARCHS)
def test_parameter_layout_and_act(pi, vf):
    D, A, N = 26, 3, 50
    p = O.init_params(D, A, pi, vf, seed=4)
    p["log_std"] = np.random.default_rng(1).normal(-0.3, 0.2, A).astype(np.float32)
    e = _engine(D, A, N, 4, pi, vf, batch_size=50, n_epochs=1)
    assert list(e.shapes.keys()) == O.param_keys(len(pi), len(vf)) == list(p.keys())
    assert e.P == sum(v.size for v in p.values())
    assert e.x3_mode() == 0                                    # other depths than two: the generic GEMM chain
    e.set_params(p)
    back = e.get_params()
    assert all(np.array_equal(back[k], p[k]) for k in p)
    rng = np.random.default_rng(0)
    obs, eps = rng.standard_normal((N, D)).astype(np.float32), rng.standard_normal((N, A)).astype(np.float32)
    a_raw, a_clip, val, lp = e.act(obs, eps)
    o_raw, o_clip, o_val, o_lp = O.act(p, obs, eps)
    assert scaled_err(a_raw, o_raw) < 1e-4 and scaled_err(val, o_val) < 1e-4
    assert np.allclose(lp, o_lp, rtol=1e-4, atol=1e-4)
    det = e.predict(obs, deterministic=True)
    assert np.allclose(det, np.clip(O.policy_outputs(p, obs)[0], -1, 1), atol=1e-4)
    e.close()


@pytest.mark.parametrize("pi,vf", ARCHS)
def test_train_matches_oracle(pi, vf):
    """PPO.train over two epochs with supplied permutations (short last minibatch), every gradient tensor of the first minibatch
    and the parameters after all optimizer steps."""
    D, A, T, N, B, E = 14, 2, 30, 7, 64, 2
    rng = np.random.default_rng(11)
    p = O.init_params(D, A, pi, vf, seed=2)
    p["log_std"] = rng.normal(-0.3, 0.2, A).astype(np.float32)
    p["action_net.weight"] *= 30
    buf, lv, dones = synthetic_rollout(T, N, D, A, seed=5)
    mean, val = O.policy_outputs(p, buf["obs"].reshape(T * N, D))
    acts = (mean + rng.standard_normal((T * N, A)).astype(np.float32) * np.exp(p["log_std"])).astype(np.float32)
    buf["actions"] = acts.reshape(T, N, A)
    buf["log_probs"] = (O.gaussian_log_prob(mean, p["log_std"], acts) + rng.normal(0, 0.1, T * N)).astype(np.float32).reshape(T, N)
    buf["values"] = (val + rng.normal(0, 0.1, T * N)).astype(np.float32).reshape(T, N)
    h = O.Hyper(gamma=0.99, gae_lambda=0.95, ent_coef=0.01, n_epochs=E, batch_size=B, learning_rate=3e-4)
    buf["advantages"], buf["returns"] = O.gae(buf["rewards"], buf["values"], buf["episode_starts"], lv, dones, h.gamma, h.gae_lambda)
    perms = np.stack([rng.permutation(T * N) for _ in range(E)])
    e = _engine(D, A, N, T, pi, vf, batch_size=B, n_epochs=E, gamma=h.gamma, gae_lambda=h.gae_lambda, ent_coef=h.ent_coef,
                learning_rate=h.learning_rate)
    e.set_params(p)
    e.load_rollout(buf, lv, dones)
    e.compute_gae()
    assert np.array_equal(e.read("advantages"), buf["advantages"])
    # first minibatch: every gradient tensor
    e.epoch_begin(perms[0])
    e.minibatch_grad(0)
    got = e.unflatten(e.read("grads"))
    _, og, _ = O.loss_and_grads(p, *O.gather_minibatch(buf, perms[0][:B]), h)
    for k in og:
        assert scaled_err(got[k], og[k]) < 1e-4, (k, scaled_err(got[k], og[k]))
    # the whole update
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
    assert e.get_optimizer_state()[2] == E * nmb
    e.close()


@pytest.mark.parametrize("pi,vf", [((64,), (64,)), ((64, 48, 32), (64, 48, 32)), ((40,), (64, 32, 16))])
def test_host_rollout_with_bootstrap_matches_oracle(pi, vf):
    """act / store / finish_rollout == oracle collect_rollout on the same env stream, including time-limit truncations whose
    rewards take gamma * V(terminal_obs) from the per-row value evaluator (one to three hidden layers)."""
    D, A, N, T = 14, 2, 6, 12
    p = O.init_params(D, A, pi, vf, seed=4)
    p["value_net.bias"] = np.array([3.0], np.float32)
    rng = np.random.default_rng(0)
    eps = rng.standard_normal((T, N, A)).astype(np.float32)
    h = O.Hyper(gamma=0.99, gae_lambda=0.9)
    env_a = O.NumpySyntheticVecEnv(N, D, A, p_term=0.1, time_limit=5, seed=3)
    obuf, _, _ = O.collect_rollout({k: v.copy() for k, v in p.items()}, env_a, env_a.reset(), np.ones(N, bool), T, h, lambda t: eps[t])
    e = _engine(D, A, N, T, pi, vf, batch_size=8, n_epochs=1, gamma=h.gamma, gae_lambda=h.gae_lambda)
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


@pytest.mark.parametrize("pi,vf", [((32,), (32,)), ((32, 24, 16), (32, 24, 16))])
@pytest.mark.parametrize("kind", ["synthetic", "goal"])
def test_device_rollout_bootstrap(pi, vf, kind):
    """Device env sources (per-step kernels of the generic path): truncated rows' rewards carry gamma * V(terminal_obs)."""
    D, A, N, T, TL = 26, 2, 96, 20, 10
    p = O.init_params(D, A, pi, vf, seed=2)
    p["value_net.bias"] = np.array([7.0], np.float32)
    e = _engine(D, A, N, T, pi, vf, batch_size=480, n_epochs=1, seed=9)
    e.set_params(p)
    if kind == "synthetic":
        e.collect_synthetic(p_term=0.0, time_limit=TL)
    else:
        e.collect_goal_env(pos_dim=2, mix=np.eye(2, A, dtype=np.float32), time_limit=TL, terminate_on_goal=False)
    e.synchronize()
    tr = e.read("truncated").astype(bool)
    assert tr.any()
    tobs = e.read("terminal_obs")[:, :D]
    _, v = O.policy_outputs(p, tobs)
    tv = e.read("terminal_values")
    assert scaled_err(tv[tr], v[tr]) < 1e-4
    # stored values / log-probs of the rollout are the policy's on the stored observations and actions
    obs, acts = e.read("obs")[:T].reshape(T * N, -1)[:, :D], e.read("actions").reshape(T * N, A)
    mean, val = O.policy_outputs(p, obs)
    assert scaled_err(e.read("values").reshape(-1), val) < 1e-4
    assert np.allclose(e.read("log_probs").reshape(-1), O.gaussian_log_prob(mean, p["log_std"], acts), rtol=1e-4, atol=1e-3)
    e.close()


@pytest.mark.parametrize("arch", [[64], [64, 48, 32], dict(pi=[40], vf=[64, 32, 16])])
def test_ppo_learns_saves_and_loads(arch, tmp_path):
    """PPO(...) with policy_kwargs net_arch of depth 1 / 3: learns on the device goal env, writes an SB3-layout zip whose
    state-dict keys follow nn.Sequential's numbering (0, 2, 4), and loads back bit-identical."""
    from mobrob_amd import checkpoint as ck
    from mobrob_amd.rl_control.ppo import PPOCtrl, PPO
    cfg = {"ppo_kwargs": {"policy": "MlpPolicy", "n_steps": 32, "batch_size": 256, "n_epochs": 2, "policy_kwargs": {"net_arch": arch}},
           "env_name": "point", "time_limit": 50, "n_envs": 64, "vec_env_type": "device_goal", "enable_gui": False, "seed": 0}
    ctrl = PPOCtrl.from_config(cfg)
    ctrl.ppo.learn(total_timesteps=3 * 32 * 64)
    path = str(tmp_path / "m.zip")
    ctrl.ppo.save(path)
    z = ck.load_zip(path)
    pi = arch["pi"] if isinstance(arch, dict) else arch
    vf = arch["vf"] if isinstance(arch, dict) else arch
    assert list(z["params"].keys()) == ck.policy_keys(len(pi), len(vf)) == O.param_keys(len(pi), len(vf))
    assert z["data"]["policy_kwargs"]["net_arch"] in (arch, dict(pi=list(pi), vf=list(vf)))
    back = PPO.load(path)
    a, b = ctrl.ppo.engine.get_flat_params(), back.engine.get_flat_params()
    assert np.array_equal(a, b) and np.isfinite(a).all()
    assert back.net_arch == (tuple(pi), tuple(vf))
    obs = np.random.default_rng(0).standard_normal((5, ctrl.ppo.obs_dim)).astype(np.float32)
    assert np.array_equal(ctrl.ppo.predict(obs, deterministic=True)[0], back.predict(obs, deterministic=True)[0])
