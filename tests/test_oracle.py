"""CPU: the oracle against the golden vectors (torch-op outputs on the reference's real weights)."""
import numpy as np
import pytest

from oracle import ppo_oracle as O
from tests.util import ENVS, scaled_err, golden_adam, golden_hyper, golden_minibatch, golden_params, load_golden, rel_err


@pytest.mark.parametrize("env", ENVS)
def test_forward_matches_golden(env):
    g = load_golden(env)
    p = golden_params(g)
    actions, clipped, value, logp = O.act(p, g["last_obs"], g["fwd/eps"])
    mean, _ = O.policy_outputs(p, g["last_obs"])
    # saturated policies: |mean| up to ~185, sigma up to ~150 -> compare relatively
    assert scaled_err(mean, g["fwd/mean"]) < 1e-5
    assert scaled_err(value, g["fwd/value"]) < 1e-5
    assert scaled_err(actions, g["fwd/actions"]) < 1e-5
    assert np.array_equal(clipped, np.clip(actions, -1, 1))
    assert np.allclose(logp, g["fwd/log_prob"], rtol=1e-5, atol=1e-4)
    assert np.allclose(O.gaussian_entropy(p["log_std"], len(value)), g["fwd/entropy"], rtol=1e-6, atol=1e-5)
    det = O.predict(p, g["last_obs"], deterministic=True)
    assert np.allclose(det, np.clip(g["fwd/mean"], -1, 1), atol=1e-4)


@pytest.mark.parametrize("env", ENVS)
def test_minibatch_step_matches_golden(env):
    g = load_golden(env)
    p, st, h = golden_params(g), golden_adam(g), golden_hyper(g)
    stats, grads, aux = O.loss_and_grads(p, *golden_minibatch(g), h)
    assert np.allclose(aux["adv"], g["step/adv_norm"], rtol=1e-5, atol=1e-6)
    assert np.allclose(aux["ratio"], g["step/ratio"], rtol=2e-4)  # exp of O(100)-magnitude log-probs
    for k in ["loss", "policy_loss", "value_loss", "entropy_loss", "approx_kl", "clip_fraction"]:
        assert abs(float(stats[k]) - float(g["step/" + k])) < 1e-4 * max(1.0, abs(float(g["step/" + k]))), k
    gn = np.sqrt(sum(float(np.sum(np.asarray(v, np.float64) ** 2)) for v in grads.values()))
    for k, v in grads.items():
        ref = g["step/grad/" + k]
        assert v.shape == ref.shape
        assert np.max(np.abs(v - ref)) < 1e-4 * max(1.0, float(np.max(np.abs(ref)))), k
    clipped, total = O.clip_grad_norm(grads, h.max_grad_norm)
    assert abs(total - float(g["step/grad_norm"])) < 1e-4 * float(g["step/grad_norm"])
    assert abs(gn - float(total)) < 1e-3 * gn
    O.adam_step(p, clipped, st, h.learning_rate, h.beta1, h.beta2, h.adam_eps)
    assert st.step == int(g["adam_step"]) + 1
    for k in p:
        assert np.max(np.abs(p[k] - g["step/p/" + k])) < 1e-6 + 1e-5 * float(np.max(np.abs(g["step/p/" + k]))), k
        assert np.allclose(st.exp_avg[k], g["step/m/" + k], rtol=1e-4, atol=1e-7), k
        assert np.allclose(st.exp_avg_sq[k], g["step/v/" + k], rtol=1e-4, atol=1e-9), k


@pytest.mark.parametrize("clip_range_vf", [None, 0.3])
def test_grads_match_torch_autograd_random_net(clip_range_vf):
    """Hand-derived backward vs torch autograd on a random 2x32 net incl. clip edge cases; with and without SB3's
    value-function clipping (`clip_range_vf`)."""
    import torch
    rng = np.random.default_rng(3)
    D, A, B = 7, 3, 64
    p = O.init_params(D, A, (32, 32), (32, 32), seed=5)
    p["log_std"] = rng.normal(0, 0.3, A).astype(np.float32)
    obs = rng.standard_normal((B, D)).astype(np.float32)
    mean, _ = O.policy_outputs(p, obs)
    act = (mean + rng.standard_normal((B, A)).astype(np.float32) * np.exp(p["log_std"])).astype(np.float32)
    lp = O.gaussian_log_prob(mean, p["log_std"], act)
    old_lp = (lp + rng.normal(0, 0.2, B)).astype(np.float32)
    old_lp[:4] = lp[:4]  # ratio == 1 exactly -> inside the clip range, tie between the two surrogates
    adv = rng.standard_normal(B).astype(np.float32)
    adv[5] = 0.0
    ret = rng.standard_normal(B).astype(np.float32)
    h = O.Hyper(ent_coef=0.01, normalize_advantage=False, clip_range_vf=clip_range_vf)
    _, val = O.policy_outputs(p, obs)
    old_v = (val + rng.normal(0, 0.35, B)).astype(np.float32)   # some rows inside, some outside the vf clip range
    stats, grads, _ = O.loss_and_grads(p, obs, act, old_v, old_lp, adv, ret, h)

    tp = {k: torch.tensor(v, requires_grad=True) for k, v in p.items()}
    x = torch.tensor(obs)
    hpi = torch.tanh(torch.tanh(x @ tp["mlp_extractor.policy_net.0.weight"].T + tp["mlp_extractor.policy_net.0.bias"])
                     @ tp["mlp_extractor.policy_net.2.weight"].T + tp["mlp_extractor.policy_net.2.bias"])
    hvf = torch.tanh(torch.tanh(x @ tp["mlp_extractor.value_net.0.weight"].T + tp["mlp_extractor.value_net.0.bias"])
                     @ tp["mlp_extractor.value_net.2.weight"].T + tp["mlp_extractor.value_net.2.bias"])
    mu = hpi @ tp["action_net.weight"].T + tp["action_net.bias"]
    v = (hvf @ tp["value_net.weight"].T + tp["value_net.bias"]).flatten()
    dist = torch.distributions.Normal(mu, torch.ones_like(mu) * tp["log_std"].exp())
    logp = dist.log_prob(torch.tensor(act)).sum(1)
    ratio = torch.exp(logp - torch.tensor(old_lp))
    a = torch.tensor(adv)
    pl = -torch.min(a * ratio, a * torch.clamp(ratio, 0.8, 1.2)).mean()
    if clip_range_vf is not None:
        ov = torch.tensor(old_v)
        frac_clipped = float(((v - ov).abs() > clip_range_vf).float().mean())
        assert 0.1 < frac_clipped < 0.9
        v = ov + torch.clamp(v - ov, -clip_range_vf, clip_range_vf)
    vl = torch.nn.functional.mse_loss(torch.tensor(ret), v)
    el = -dist.entropy().sum(1).mean()
    loss = pl + 0.01 * el + 0.5 * vl
    loss.backward()
    assert abs(loss.item() - float(stats["loss"])) < 1e-5
    for k in p:
        assert np.allclose(grads[k], tp[k].grad.numpy(), rtol=2e-4, atol=2e-6), k


def test_gae_properties():
    rng = np.random.default_rng(0)
    T, N = 50, 6
    r = rng.standard_normal((T, N)).astype(np.float32)
    v = rng.standard_normal((T, N)).astype(np.float32)
    es = np.zeros((T, N), np.float32)
    lv = rng.standard_normal(N).astype(np.float32)
    dones = np.zeros(N, bool)
    # lambda = 0 -> advantage == one-step TD error
    adv, ret = O.gae(r, v, es, lv, dones, 0.99, 0.0)
    nv = np.concatenate([v[1:], lv[None]], 0)
    assert np.allclose(adv, r + np.float32(0.99) * nv - v, atol=1e-6)
    assert np.allclose(ret, adv + v)
    # no episode boundaries -> closed form in float64
    adv, _ = O.gae(r, v, es, lv, dones, 0.97, 0.9)
    delta = r.astype(np.float64) + 0.97 * nv - v
    ref = np.zeros((T, N))
    acc = np.zeros(N)
    for t in reversed(range(T)):
        acc = delta[t] + 0.97 * 0.9 * acc
        ref[t] = acc
    assert np.allclose(adv, ref, atol=1e-5)
    # episode start at t+1 cuts both the bootstrap and the trace
    es2 = es.copy()
    es2[10, 2] = 1.0
    adv2, _ = O.gae(r, v, es2, lv, dones, 0.97, 0.9)
    assert abs(adv2[9, 2] - (r[9, 2] - v[9, 2])) < 1e-6
    assert np.allclose(adv2[10:, 2], adv[10:, 2])
    # dones at the last step cut the bootstrap from last_values
    d2 = dones.copy()
    d2[1] = True
    adv3, _ = O.gae(r, v, es, lv, d2, 0.97, 0.9)
    assert abs(adv3[T - 1, 1] - (r[T - 1, 1] - v[T - 1, 1])) < 1e-6
    assert adv3.dtype == np.float32


def test_feistel_is_permutation():
    for n in [1, 2, 7, 100, 4096, 16000, 65537]:
        perm = O.feistel_permutation(n, key=0xDEADBEEF12345 + n)
        assert np.array_equal(np.sort(perm), np.arange(n))
    a = O.feistel_permutation(16000, 1)
    b = O.feistel_permutation(16000, 2)
    assert (a != b).mean() > 0.99
    assert abs(np.corrcoef(a, np.arange(16000))[0, 1]) < 0.05


def test_flat_index_is_env_major():
    T, N = 5, 3
    x = np.arange(T * N).reshape(T, N)
    flat = x.swapaxes(0, 1).reshape(T * N)  # SB3 swap_and_flatten
    t, n = O.flat_to_tn(np.arange(T * N), T)
    assert np.array_equal(x[t, n], flat)


def test_checkpoint_sanity_anchors():
    """Anchors recorded during the survey."""
    g = load_golden("point")
    p = golden_params(g)
    mean, v = O.policy_outputs(p, g["last_obs"])
    assert np.allclose(p["log_std"], [3.001, 3.634], atol=2e-3)
    assert np.allclose(mean[0], [-24.102, -0.758], atol=2e-3)
    assert np.allclose(v, [1.507, 1.516], atol=2e-3)
    g = load_golden("doggo")
    _, v = O.policy_outputs(golden_params(g), g["last_obs"])
    assert np.allclose(v[:4], [3.464, 3.274, 3.692, 3.804], atol=2e-3)
    assert int(g["adam_step"]) == 1499200 == int(g["hyper/_n_updates"]) * 160


def test_whole_iteration_matches_torch_golden():
    """tests/golden/train_loop.npz (make_loop_fixture.py): RolloutBuffer GAE, env-major flatten, 2 epochs of 4
    minibatches (the last one short) computed with NumPy + torch from the real doggo checkpoint.  The oracle's gae /
    gather / train loop must land on the same advantages (bit for bit), losses and parameters."""
    import os
    from collections import OrderedDict
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "train_loop.npz"))
    T, N, B, E, D, A = (int(x) for x in g["shape"])
    gamma, lam, clip, ent_coef, vf_coef, max_norm, lr, eps = (float(x) for x in g["hyper"])
    keys = O.param_keys()
    p = OrderedDict((k, g[f"p0/{k}"].copy()) for k in keys)
    st = O.AdamState(OrderedDict((k, g[f"m0/{k}"].copy()) for k in keys), OrderedDict((k, g[f"v0/{k}"].copy()) for k in keys),
                     int(g["adam_step"]))
    adv, ret = O.gae(g["rewards"], g["values"], g["episode_starts"], g["last_values"], g["dones"], gamma, lam)
    assert np.array_equal(adv, g["advantages"]) and np.array_equal(ret, g["returns"])
    buf = {k: g[k] for k in ("obs", "actions", "rewards", "episode_starts", "values", "log_probs", "advantages", "returns")}
    h = O.Hyper(gamma=gamma, gae_lambda=lam, clip_range=clip, ent_coef=ent_coef, vf_coef=vf_coef, max_grad_norm=max_norm,
                learning_rate=lr, adam_eps=eps, n_epochs=E, batch_size=B)
    stats = O.train(p, st, buf, h, g["perms"])
    ref = g["stats"]
    assert len(stats) == len(ref) == E * 4 and int(ref[3][7]) == T * N - 3 * B   # last minibatch of an epoch is short
    for s, r in zip(stats, ref):
        for k, j in (("policy_loss", 0), ("value_loss", 1), ("entropy_loss", 2), ("loss", 3), ("approx_kl", 4),
                     ("clip_fraction", 5), ("grad_norm", 6)):
            assert abs(float(s[k]) - r[j]) < 2e-5 * max(1.0, abs(r[j])), (k, float(s[k]), r[j])
    for k in keys:
        assert np.max(np.abs(p[k] - g[f"p1/{k}"])) < 2e-6, k
        assert scaled_err(st.exp_avg[k], g[f"m1/{k}"]) < 1e-4, k
    assert st.step == int(g["adam_step"]) + E * 4


def test_target_kl_stops_before_the_offending_optimizer_step():
    """SB3 PPO.train with target_kl: the first minibatch whose approx_kl exceeds 1.5 x target is evaluated (its
    statistics are logged) but not applied, and nothing after it runs."""
    from tests.util import synthetic_rollout
    D, A, T, N, B = 6, 2, 8, 8, 16
    p = O.init_params(D, A, (16, 16), (16, 16), seed=1)
    buf, lv, dones = synthetic_rollout(T, N, D, A, seed=2)
    buf["advantages"], buf["returns"] = O.gae(buf["rewards"], buf["values"], buf["episode_starts"], lv, dones, 0.99, 0.95)
    perms = [np.random.default_rng(e).permutation(T * N) for e in range(3)]
    free = O.train({k: v.copy() for k, v in p.items()}, O.AdamState.zeros_like(p), buf, O.Hyper(n_epochs=3, batch_size=B), perms)
    kls = [float(s["approx_kl"]) for s in free]
    assert len(kls) == 12
    target = sorted(kls)[6] / 1.5                       # a threshold some minibatch in the middle crosses
    first = next(i for i, k in enumerate(kls) if k > 1.5 * target)
    p2, st2 = {k: v.copy() for k, v in p.items()}, O.AdamState.zeros_like(p)
    out = O.train(p2, st2, buf, O.Hyper(n_epochs=3, batch_size=B, target_kl=target), perms)
    assert len(out) == first + 1 and out[-1].get("early_stop") and st2.step == first
    assert all(abs(float(a["approx_kl"]) - float(b["approx_kl"])) < 1e-7 for a, b in zip(out, free))


def test_relu_networks_match_torch_autograd():
    """`policy_kwargs=dict(activation_fn=nn.ReLU)` (an SB3 keyword the reference passes through, ppo.py:58): the oracle's
    forward and hand-written backward with ReLU hidden layers against torch autograd on the same loss."""
    import torch
    rng = np.random.default_rng(0)
    D, A, H, B = 7, 3, 16, 40
    p = O.init_params(D, A, (H, H), (H, H), seed=1)
    obs = rng.standard_normal((B, D)).astype(np.float32)
    act = rng.standard_normal((B, A)).astype(np.float32)
    oldv, adv, ret = (rng.standard_normal(B).astype(np.float32) for _ in range(3))
    oldlp = (-3 + 0.1 * rng.standard_normal(B)).astype(np.float32)
    h = O.Hyper(activation="relu", ent_coef=0.01)
    stats, g, _ = O.loss_and_grads(p, obs, act, oldv, oldlp, adv, ret, h)
    tp = {k: torch.tensor(v, requires_grad=True) for k, v in p.items()}
    x = torch.tensor(obs)

    def net(pre):
        h1 = torch.relu(x @ tp[pre + ".0.weight"].T + tp[pre + ".0.bias"])
        return torch.relu(h1 @ tp[pre + ".2.weight"].T + tp[pre + ".2.bias"])
    mean = net("mlp_extractor.policy_net") @ tp["action_net.weight"].T + tp["action_net.bias"]
    val = (net("mlp_extractor.value_net") @ tp["value_net.weight"].T + tp["value_net.bias"])[:, 0]
    dist = torch.distributions.Normal(mean, torch.ones_like(mean) * tp["log_std"].exp())
    lp, ent = dist.log_prob(torch.tensor(act)).sum(1), dist.entropy().sum(1)
    a = torch.tensor(adv)
    a = (a - a.mean()) / (a.std() + 1e-8)
    ratio = torch.exp(lp - torch.tensor(oldlp))
    loss = (-torch.min(a * ratio, a * torch.clamp(ratio, 0.8, 1.2)).mean() + 0.01 * (-ent.mean())
            + 0.5 * torch.nn.functional.mse_loss(torch.tensor(ret), val))
    loss.backward()
    assert abs(float(loss.detach()) - float(stats["loss"])) < 1e-6
    for k in g:
        assert np.max(np.abs(tp[k].grad.numpy() - g[k])) < 1e-6, k
    m_t, v_t = O.policy_outputs(p, obs, activation="tanh")
    m_r, v_r = O.policy_outputs(p, obs, activation="relu")
    assert np.max(np.abs(m_r - mean.detach().numpy())) < 1e-6 and np.max(np.abs(m_t - m_r)) > 1e-3


def test_torch_baseline_matches_the_oracle():
    """oracle/torch_baseline.py (the torch-CPU leg of bench.py's cpu_baseline: nn.Linear / Normal / autograd / clip_grad_norm_ /
    optim.Adam in SB3's order) runs the same iteration as the NumPy oracle: same rollout on the same synthetic env source and
    noise, same parameters after two epochs of minibatch steps (incl. a short last minibatch)."""
    import torch
    from oracle import torch_baseline as TB
    torch.set_num_threads(1)
    D, A, H, N, T = 9, 3, 32, 6, 17
    rng = np.random.default_rng(4)
    p = O.init_params(D, A, (H, H), (H, H), seed=1)
    p["log_std"] = rng.normal(-0.2, 0.1, A).astype(np.float32)
    h = O.Hyper(n_epochs=2, batch_size=40, ent_coef=0.01)
    eps = rng.standard_normal((T, N, A)).astype(np.float32)
    perms = [rng.permutation(T * N) for _ in range(2)]
    env_a = O.NumpySyntheticVecEnv(N, D, A, p_term=0.05, time_limit=9, seed=3)
    env_b = O.NumpySyntheticVecEnv(N, D, A, p_term=0.05, time_limit=9, seed=3)
    pol = TB.TorchPolicy(p)
    buf_t, _, _ = TB.collect_rollout(pol, env_b, env_b.reset(), np.ones(N, bool), T, h, lambda t: eps[t])
    buf_o, _, _ = O.collect_rollout(p, env_a, env_a.reset(), np.ones(N, bool), T, h, lambda t: eps[t])
    for k in ("obs", "rewards", "episode_starts"):
        assert np.allclose(buf_t[k], buf_o[k], atol=1e-5), k
    for k in ("actions", "values", "log_probs", "advantages", "returns"):
        assert np.allclose(buf_t[k], buf_o[k], rtol=1e-4, atol=1e-4), k
    TB.train(pol, TB.make_optimizer(pol, h), buf_o, h, perms)
    q = {k: v.copy() for k, v in p.items()}
    O.train(q, O.AdamState.zeros_like(q), buf_o, h, perms)
    got = pol.state()
    for k in q:
        assert np.max(np.abs(got[k] - q[k])) < 2e-5, (k, float(np.max(np.abs(got[k] - q[k]))))


def test_learn_loop_counters_equal_the_reference_checkpoints():
    """The counters SB3's learn loop left in the five reference zips (reference-HELD outputs: tests/golden/reference_counters.json,
    made by tests/golden/make_counter_fixture.py) against the oracle's restatement of the loop's bookkeeping: what a timestep
    is, when the loop stops (drone overshoots its 1 000 000 by a whole rollout, progress -0.008), that `_n_updates` counts epochs
    and Adam counts minibatches, and what a CheckpointCallback sees mid-collection (doggo: saved at 30 M of 50 M, after 1874
    trained rollouts, `_current_progress_remaining` = 1 - 1874 * 16000 / 5e7)."""
    import json
    import os
    ref = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_counters.json")))
    assert set(ref) == {"point", "car", "doggo", "drone", "turtlebot3"}
    for env, r in ref.items():
        finished = r["num_timesteps"] >= r["_total_timesteps"]
        got = O.learn_loop_counters(r["_total_timesteps"], r["n_steps"], r["n_envs"], r["batch_size"], r["n_epochs"],
                                    checkpoint_at_timestep=None if finished else r["num_timesteps"])
        for k in ("num_timesteps", "_n_updates", "adam_step"):
            assert got[k] == r[k], (env, k, got[k], r[k])
        assert abs(got["_current_progress_remaining"] - r["_current_progress_remaining"]) < 1e-12, (env, got, r)


def test_arch_fixture_lists_its_cases():
    from tests.util import ARCH_CASES, arch_cases
    assert arch_cases() == ARCH_CASES


@pytest.mark.parametrize("name", __import__("tests.util", fromlist=["ARCH_CASES"]).ARCH_CASES)
def test_other_activations_and_depths_match_torch_golden(name):
    """`policy_kwargs` beyond the YAMLs' two tanh layers (activation_fn, net_arch depths 1 .. 8; the reference splats ppo_kwargs
    into PPO, ppo.py:58): the oracle's forward, hand-written backward, clip and Adam against torch's own modules / autograd /
    clip_grad_norm_ / optim.Adam (tests/golden/make_arch_fixture.py)."""
    from tests.util import arch_case
    c, act, pi, vf, p, h = arch_case(name)
    mean, value = O.policy_outputs(p, c["fwd/obs"], activation=act)
    assert scaled_err(mean, c["fwd/mean"]) < 1e-5 and scaled_err(value, c["fwd/value"]) < 1e-5
    actions, _, _, logp = O.act(p, c["fwd/obs"], c["fwd/eps"], activation=act)
    assert scaled_err(actions, c["fwd/actions"]) < 1e-5 and np.allclose(logp, c["fwd/log_prob"], rtol=1e-5, atol=1e-4)
    mb = (c["mb/obs"], c["mb/actions"], c["mb/old_values"], c["mb/old_log_prob"], c["mb/advantages"], c["mb/returns"])
    stats, grads, _ = O.loss_and_grads(p, *mb, h)
    for k in ["loss", "policy_loss", "value_loss", "entropy_loss", "approx_kl", "clip_fraction"]:
        assert abs(float(stats[k]) - float(c["step/" + k])) < 1e-5 * max(1.0, abs(float(c["step/" + k]))), k
    for k, v in grads.items():
        ref = c["step/grad/" + k]
        assert v.shape == ref.shape and np.max(np.abs(v - ref)) < 1e-5 * max(1.0, float(np.max(np.abs(ref)))), k
    clipped, total = O.clip_grad_norm(grads, h.max_grad_norm)
    assert abs(float(total) - float(c["step/grad_norm"])) < 1e-5 * float(c["step/grad_norm"])
    st = O.AdamState.zeros_like(p)
    newp = O.adam_step({k: v.copy() for k, v in p.items()}, clipped, st, h.learning_rate, h.beta1, h.beta2, h.adam_eps)
    for k in p:
        assert np.max(np.abs(newp[k] - c["step/p/" + k])) < 1e-6, k
        assert np.allclose(st.exp_avg[k], c["step/m/" + k], rtol=1e-4, atol=1e-8) and np.allclose(st.exp_avg_sq[k], c["step/v/" + k], rtol=1e-4, atol=1e-11), k


@pytest.mark.parametrize("name", __import__("tests.util", fromlist=["SDE_CASES"]).SDE_CASES)
def test_gsde_matches_torch_golden(name):
    """PPO(use_sde=True): the oracle's StateDependentNoiseDistribution restatement (sample, log-prob, entropy, the log_std matrix's
    gradient through the variance) against torch's own ops (tests/golden/make_sde_fixture.py)."""
    from tests.util import sde_case
    c, act, pi, vf, p, h = sde_case(name)
    A = c["fwd/actions"].shape[1]
    assert p["log_std"].shape == (pi[-1], A if bool(c["full_std"]) else 1)
    theta = O.sde_exploration_matrices(p["log_std"], c["fwd/z"], h.sde_use_expln)
    actions, clipped, value, logp = O.act_sde(p, c["fwd/obs"], theta, activation=act, use_expln=h.sde_use_expln)
    assert scaled_err(actions, c["fwd/actions"]) < 1e-5 and scaled_err(value, c["fwd/value"]) < 1e-5
    assert np.allclose(logp, c["fwd/log_prob"], rtol=1e-5, atol=1e-4) and np.array_equal(clipped, np.clip(actions, -1, 1))
    single, _, _, _ = O.act_sde(p, c["fwd/obs"], theta[0], activation=act, use_expln=h.sde_use_expln)
    assert scaled_err(single, c["fwd/single"]) < 1e-5
    mb = (c["mb/obs"], c["mb/actions"], c["mb/old_values"], c["mb/old_log_prob"], c["mb/advantages"], c["mb/returns"])
    stats, grads, aux = O.loss_and_grads(p, *mb, h)
    for k in ["loss", "policy_loss", "value_loss", "entropy_loss", "approx_kl", "clip_fraction"]:
        assert abs(float(stats[k]) - float(c["step/" + k])) < 1e-5 * max(1.0, abs(float(c["step/" + k]))), k
    for k, v in grads.items():
        ref = c["step/grad/" + k]
        assert v.shape == ref.shape and np.max(np.abs(v - ref)) < 1e-5 * max(1.0, float(np.max(np.abs(ref)))), k
    assert float(np.max(np.abs(c["step/grad/log_std"]))) > 1e-4      # (the comparison above is not vacuous for the matrix)
    clipped_g, total = O.clip_grad_norm(grads, h.max_grad_norm)
    assert abs(float(total) - float(c["step/grad_norm"])) < 1e-5 * float(c["step/grad_norm"])
    newp = O.adam_step({k: v.copy() for k, v in p.items()}, clipped_g, O.AdamState.zeros_like(p), h.learning_rate, h.beta1, h.beta2, h.adam_eps)
    for k in p:
        assert np.max(np.abs(newp[k] - c["step/p/" + k])) < 1e-6, k


def test_gsde_rollout_keeps_one_matrix_per_environment_between_reset_points():
    """collect_rollout under use_sde: the noise of an environment is ONE linear map of its latent between reset_noise points (t = 0 and
    every sde_sample_freq steps), and the stored log-probs are the state-dependent distribution's at the stored actions."""
    D, A, N, T, pi, vf, freq = 10, 2, 4, 12, (16, 8), (16,), 6
    p = O.init_params(D, A, pi, vf, seed=4)
    p["log_std"] = np.full((pi[-1], A), -1.0, np.float32)
    zs = np.random.default_rng(0).standard_normal((T, N, pi[-1], A)).astype(np.float32)
    asked = []
    h = O.Hyper(use_sde=True, sde_sample_freq=freq)
    env = O.NumpySyntheticVecEnv(N, D, A, p_term=0.05, time_limit=7, seed=3)
    buf, _, _ = O.collect_rollout(p, env, env.reset(), np.ones(N, bool), T, h, lambda t: (asked.append(t), zs[t])[1])
    assert asked == [0, 6]
    for t in range(T):
        lat = O.mlp_latents(p, buf["obs"][t])[0][-1]
        mean = O.policy_outputs(p, buf["obs"][t])[0]
        theta = O.sde_exploration_matrices(p["log_std"], zs[0 if t < 6 else 6])
        assert np.allclose(buf["actions"][t], mean + np.einsum("nk,nka->na", lat, theta), atol=1e-5)
        assert np.allclose(buf["log_probs"][t], O.normal_log_prob(mean, O.sde_sigma(lat, p["log_std"]), buf["actions"][t]), atol=1e-5)
