"""GPU parity tests: HIP engine (through the C ABI) vs the CPU oracle and the golden vectors."""
import os

import numpy as np
import pytest

from oracle import ppo_oracle as O
from tests.util import (ENVS, golden_adam, golden_hyper, golden_minibatch, golden_params, load_golden, scaled_err,
                        synthetic_rollout)

pytestmark = pytest.mark.gpu


def make_engine(g=None, **kw):
    from mobrob_amd.engine import PPOEngine
    if g is not None:
        h = golden_hyper(g)
        D, A = g["last_obs"].shape[1], g["p/log_std"].shape[0]
        base = dict(obs_dim=D, act_dim=A, n_envs=g["last_obs"].shape[0], n_steps=4, batch_size=100, n_epochs=1,
                    gamma=h.gamma, gae_lambda=h.gae_lambda, clip_range=h.clip_range, ent_coef=h.ent_coef,
                    vf_coef=h.vf_coef, max_grad_norm=h.max_grad_norm, learning_rate=h.learning_rate,
                    adam_betas=(h.beta1, h.beta2), adam_eps=h.adam_eps)
        base.update(kw)
        return PPOEngine(**base)
    return PPOEngine(**kw)


@pytest.mark.parametrize("fast", [True, False])  # fused 64-wide kernels / generic GEMM chain
@pytest.mark.parametrize("env", ENVS)
def test_act_matches_golden_and_oracle(env, fast):
    g = load_golden(env)
    p = golden_params(g)
    e = make_engine(g, fast_kernels=fast)
    e.set_params(p)
    a_raw, a_clip, val, lp = e.act(g["last_obs"], g["fwd/eps"])
    o_raw, o_clip, o_val, o_lp = O.act(p, g["last_obs"], g["fwd/eps"])
    # tolerance: north_star 1e-4 (fp32), relative to the magnitude of the compared tensor
    assert scaled_err(a_raw, g["fwd/actions"]) < 1e-4 and scaled_err(a_raw, o_raw) < 1e-4
    assert scaled_err(val, g["fwd/value"]) < 1e-4 and scaled_err(val, o_val) < 1e-4
    # log-probs elementwise to 1e-4 (relative for large magnitudes, absolute where |logp| is small)
    assert np.allclose(lp, g["fwd/log_prob"], rtol=1e-4, atol=1e-4), float(np.max(np.abs(lp - g["fwd/log_prob"])))
    assert np.allclose(lp, o_lp, rtol=1e-4, atol=1e-4), float(np.max(np.abs(lp - o_lp)))
    assert np.array_equal(a_clip, np.clip(a_raw, -1, 1))
    det = e.predict(g["last_obs"], deterministic=True)
    assert np.allclose(det, np.clip(g["fwd/mean"], -1, 1), atol=1e-4)
    one = e.predict(g["last_obs"][0], deterministic=True)
    assert one.shape == (p["log_std"].shape[0],) and np.allclose(one, det[0], atol=1e-6)
    e.close()


# k_gae walks time in 64-step tiles over 16-env column groups: cover both sides of every boundary
@pytest.mark.parametrize("T,N", [(1, 1), (2, 3), (37, 5), (64, 130), (257, 64), (65, 16), (66, 17), (129, 15), (130, 33),
                                 (1000, 48)])
def test_gae_bit_exact(T, N):
    buf, lv, dones = synthetic_rollout(T, N, 4, 2, seed=T * 1000 + N, p_done=0.05)
    e = make_engine(obs_dim=4, act_dim=2, n_envs=N, n_steps=T, batch_size=8, n_epochs=1, gamma=0.99, gae_lambda=0.95)
    e.load_rollout(buf, lv, dones)
    e.compute_gae()
    adv, ret = O.gae(buf["rewards"], buf["values"], buf["episode_starts"], lv, dones, 0.99, 0.95)
    assert np.array_equal(e.read("advantages"), adv)
    assert np.array_equal(e.read("returns"), ret)
    e.close()


@pytest.mark.parametrize("gamma,lam", [(0.99, 0.5), (0.999, 0.99), (1.0, 1.0), (0.9, 0.0)])
def test_gae_bit_exact_hyper(gamma, lam):
    buf, lv, dones = synthetic_rollout(100, 70, 4, 2, seed=7, p_done=0.03)
    e = make_engine(obs_dim=4, act_dim=2, n_envs=70, n_steps=100, batch_size=8, n_epochs=1, gamma=gamma, gae_lambda=lam)
    e.load_rollout(buf, lv, dones)
    e.compute_gae()
    adv, ret = O.gae(buf["rewards"], buf["values"], buf["episode_starts"], lv, dones, gamma, lam)
    assert np.array_equal(e.read("advantages"), adv) and np.array_equal(e.read("returns"), ret)
    e.close()


@pytest.mark.parametrize("fast", [True, False])
@pytest.mark.parametrize("env", ENVS)
def test_minibatch_step_matches_golden(env, fast):
    """One optimizer step from the checkpoint's real weights + real Adam state on the golden minibatch."""
    g = load_golden(env)
    p, st, h = golden_params(g), golden_adam(g), golden_hyper(g)
    obs, act, old_v, old_lp, adv, ret = golden_minibatch(g)
    B, D, A = obs.shape[0], obs.shape[1], act.shape[1]
    # inject the minibatch as a T=B, N=1 rollout; identity permutation -> the minibatch is rows 0..B-1
    e = make_engine(g, n_envs=1, n_steps=B, batch_size=B, n_epochs=1, fast_kernels=fast)
    e.set_params(p)
    e.set_optimizer_state(st.exp_avg, st.exp_avg_sq, st.step)
    buf = dict(obs=obs[:, None], actions=act[:, None], rewards=np.zeros((B, 1), np.float32),
               episode_starts=np.zeros((B, 1), np.float32), values=old_v[:, None], log_probs=old_lp[:, None],
               advantages=adv[:, None], returns=ret[:, None])
    e.load_rollout(buf, np.zeros(1, np.float32), np.zeros(1, bool))
    e.epoch_begin(np.arange(B))
    e.minibatch_grad(0)
    grads = e.unflatten(e.read("grads"))
    for k, v in grads.items():
        ref = g["step/grad/" + k]
        assert np.max(np.abs(v - ref)) < 1e-4 * max(1.0, float(np.max(np.abs(ref)))), k
    e.minibatch_apply()
    stats = e.fetch_step_stats()[-1]
    for i, k in enumerate(["policy_loss", "value_loss", "entropy_loss", "loss", "approx_kl", "clip_fraction",
                           "grad_norm"]):
        ref = float(g["step/" + k])
        assert abs(stats[i] - ref) < 1e-4 * max(1.0, abs(ref)), (k, stats[i], ref)
    newp = e.get_params()
    m, v, step = e.get_optimizer_state()
    assert step == int(g["adam_step"]) + 1
    for k in newp:
        assert np.max(np.abs(newp[k] - g["step/p/" + k])) < 1e-6 + 1e-5 * float(np.max(np.abs(g["step/p/" + k]))), k
        assert np.allclose(m[k], g["step/m/" + k], rtol=1e-3, atol=1e-6), k
        assert np.allclose(v[k], g["step/v/" + k], rtol=1e-3, atol=1e-8), k
    e.close()


@pytest.mark.parametrize("shape", [dict(D=14, A=2, H=64, T=40, N=5, B=50, E=2),
                                   dict(D=58, A=12, H=64, T=25, N=16, B=100, E=2),
                                   dict(D=58, A=12, H=256, T=16, N=24, B=128, E=1),
                                   dict(D=43, A=2, H=64, T=30, N=7, B=64, E=2),    # 210 = 3*64 + 18: short last batch
                                   dict(D=12, A=18, H=64, T=64, N=300, B=6000, E=1),  # many tiles per wave
                                   dict(D=58, A=12, H=64, T=25, N=16, B=100, E=2, fast=False)])
def test_full_train_matches_oracle(shape):
    """PPO.train over several epochs with supplied permutations (incl. a short final minibatch)."""
    D, A, H, T, N, B, E = (shape[k] for k in "DAHTNBE")
    rng = np.random.default_rng(11)
    p = O.init_params(D, A, (H, H), (H, H), seed=2)
    p["log_std"] = rng.normal(-0.3, 0.2, A).astype(np.float32)
    p["action_net.weight"] *= 30
    buf, lv, dones = synthetic_rollout(T, N, D, A, seed=5)
    # make the stored actions/log-probs consistent with the policy so that ratios are O(1)
    flat_obs = buf["obs"].reshape(T * N, D)
    mean, val = O.policy_outputs(p, flat_obs)
    acts = (mean + rng.standard_normal((T * N, A)).astype(np.float32) * np.exp(p["log_std"])).astype(np.float32)
    buf["actions"] = acts.reshape(T, N, A)
    buf["log_probs"] = (O.gaussian_log_prob(mean, p["log_std"], acts) + rng.normal(0, 0.1, T * N)).astype(np.float32).reshape(T, N)
    buf["values"] = (val + rng.normal(0, 0.1, T * N)).astype(np.float32).reshape(T, N)
    h = O.Hyper(gamma=0.99, gae_lambda=0.95, ent_coef=0.01, n_epochs=E, batch_size=B, learning_rate=3e-4)
    buf["advantages"], buf["returns"] = O.gae(buf["rewards"], buf["values"], buf["episode_starts"], lv, dones, h.gamma, h.gae_lambda)
    perms = np.stack([rng.permutation(T * N) for _ in range(E)])

    e = make_engine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=B, n_epochs=E, pi=(H, H), vf=(H, H),
                    gamma=h.gamma, gae_lambda=h.gae_lambda, ent_coef=h.ent_coef, learning_rate=h.learning_rate,
                    fast_kernels=shape.get("fast", True))
    e.set_params(p)
    e.load_rollout(buf, lv, dones)
    e.compute_gae()
    assert np.array_equal(e.read("advantages"), buf["advantages"])
    stats = e.train(perms)
    st = O.AdamState.zeros_like(p)
    ostats = O.train(p, st, buf, h, perms)
    nmb = -(-T * N // B)
    assert stats["n_minibatches"] == E * nmb == len(ostats)
    last = ostats[-nmb:]
    for k in ["policy_loss", "value_loss", "loss", "approx_kl", "clip_fraction", "grad_norm"]:
        ref = float(np.mean([float(s[k]) for s in last]))
        assert abs(stats[k] - ref) < 2e-4 * max(1.0, abs(ref)), (k, stats[k], ref)
    newp = e.get_params()
    for k in p:
        # after E*nmb Adam steps of size 3e-4 the trajectories must still agree to 1e-4 absolute
        assert np.max(np.abs(newp[k] - p[k])) < 1e-4, (k, float(np.max(np.abs(newp[k] - p[k]))))
    m, v, step = e.get_optimizer_state()
    assert step == E * nmb
    e.close()


def test_feistel_bit_exact():
    e = make_engine(obs_dim=4, act_dim=2, n_envs=2, n_steps=2, batch_size=2, n_epochs=1)
    for n, key in [(1, 5), (7, 1), (100, 2 ** 40 + 17), (16000, 0xDEADBEEF12345), (65537, 3), (1 << 20, 99)]:
        assert np.array_equal(e.feistel_permutation(n, key), O.feistel_permutation(n, key)), (n, key)
    e.close()


@pytest.mark.parametrize("T,N,B", [(64, 40, 100),        # ragged: 2560 rows, 26 minibatches, the last one 60 rows
                                   (37, 5, 64),           # 185 rows: not a multiple of four, three minibatches
                                   (250, 128, 4096),      # 32 000 rows, several workgroups per minibatch
                                   (16, 300, 2),          # 2400 minibatches: more than the LDS bins hold -> global adds
                                   (3, 3, 64)])           # one short minibatch
def test_advantage_statistics_are_streamed_exactly_and_reproducibly(T, N, B):
    """k_adv_stats_stream: one coalesced pass in storage order, every element finds its minibatch through the INVERSE
    permutation, sums in fixed-point integers -> equal to float64 sums to 1e-12 and the same bits on every run, for the
    device-drawn Feistel permutation and for a host-supplied one."""
    from tests.test_full_size_gpu import _device_perm_key
    rng = np.random.default_rng(T * N)
    adv = (rng.standard_normal((T, N)) * rng.choice([1e-3, 1.0, 40.0])).astype(np.float32)
    adv[rng.integers(0, T), rng.integers(0, N)] = 173.25                       # an outlier sets the fixed-point scale
    total, nmb = T * N, -(-T * N // B)
    flat = adv.T.reshape(-1)                                                    # SB3's env-major flat order: n * T + t

    def expected(perm):
        out = np.zeros((nmb, 4))
        for mb in range(nmb):
            x = flat[perm[mb * B:(mb + 1) * B]].astype(np.float64)
            out[mb] = [x.sum(), (x * x).sum(), len(x), 0.0]
        return out

    seed = 11
    runs = []
    for rep in range(2):
        e = make_engine(obs_dim=4, act_dim=2, n_envs=N, n_steps=T, batch_size=B, n_epochs=1, seed=seed)
        e.write("advantages", adv)
        e.mark_rollout_ready()
        got = []
        for draw in range(2):                                                   # two device-drawn permutations ...
            e.epoch_begin(None)
            e.synchronize()
            got.append(e.read("advstat"))
            want = expected(O.feistel_permutation(total, _device_perm_key(seed, draw)))
            assert np.allclose(got[-1], want, rtol=1e-11, atol=1e-9 * float(np.abs(adv).max()) ** 2), (draw, np.abs(got[-1] - want).max())
        perm = rng.permutation(total) if rep == 0 else runs[0][2]
        e.epoch_begin(perm)                                                     # ... and a host-supplied one
        e.synchronize()
        got.append(e.read("advstat"))
        assert np.allclose(got[-1], expected(perm), rtol=1e-11, atol=1e-9 * float(np.abs(adv).max()) ** 2)
        assert nmb == 1 or not np.array_equal(got[0], got[1])                   # the draws differ (one minibatch: order-free sums agree exactly)
        runs.append((got[0], got[1], perm, got[2]))
        e.close()
    for k in (0, 1, 3):
        assert np.array_equal(runs[0][k], runs[1][k])                          # bit-reproducible, whatever the atomics' order
    # all-zero advantages (a scale with nothing to scale) stay zero
    e = make_engine(obs_dim=4, act_dim=2, n_envs=N, n_steps=T, batch_size=B, n_epochs=1)
    e.write("advantages", np.zeros((T, N), np.float32))
    e.mark_rollout_ready()
    e.epoch_begin(None)
    st = e.read("advstat")
    assert np.all(st[:, :2] == 0.0) and st[:, 2].sum() == total
    e.close()


def test_host_rollout_matches_oracle():
    """act/store/finish_rollout through the host path == oracle collect_rollout on the same env stream,
    including a time-limit truncation with bootstrap."""
    D, A, N, T = 14, 2, 6, 12
    p = O.init_params(D, A, seed=4)
    rng = np.random.default_rng(0)
    eps = rng.standard_normal((T, N, A)).astype(np.float32)
    h = O.Hyper(gamma=0.99, gae_lambda=0.9)
    env_a = O.NumpySyntheticVecEnv(N, D, A, p_term=0.1, time_limit=5, seed=3)
    obs0 = env_a.reset()
    obuf, _, _ = O.collect_rollout({k: v.copy() for k, v in p.items()}, env_a, obs0, np.ones(N, bool), T, h,
                                   lambda t: eps[t])
    e = make_engine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=8, n_epochs=1, gamma=h.gamma,
                    gae_lambda=h.gae_lambda)
    e.set_params(p)
    env_b = O.NumpySyntheticVecEnv(N, D, A, p_term=0.1, time_limit=5, seed=3)
    obs = env_b.reset()
    e.rollout_begin()
    saw_trunc = False
    for t in range(T):
        a_raw, a_clip, val, lp = e.act(obs, eps[t])
        obs, rew, done, trunc, term_obs = env_b.step(a_clip)
        saw_trunc |= bool(trunc.any())
        e.store(rew, done, trunc, term_obs)
    e.finish_rollout(obs, done)
    assert saw_trunc
    got = {k: e.read(k) for k in ["actions", "rewards", "episode_starts", "values", "log_probs", "advantages", "returns"]}
    assert np.array_equal(e.read("obs")[:T], obuf["obs"])
    assert np.array_equal(got["episode_starts"], obuf["episode_starts"])
    for k in ["actions", "rewards", "values", "log_probs", "advantages", "returns"]:
        assert scaled_err(got[k], obuf[k]) < 1e-4, k
    e.close()


@pytest.mark.parametrize("H", [64, 256])
def test_three_learn_iterations_match_oracle(H):
    """OnPolicyAlgorithm.learn for three iterations -- collect (with carried last_obs / episode starts), GAE, all
    epochs of minibatch updates -- engine vs oracle on the same env stream, action noise and permutations."""
    D, A, N, T, B, E, ITERS = 26, 2, 10, 16, 40, 2, 3
    p0 = O.init_params(D, A, (H, H), (H, H), seed=12)
    h = O.Hyper(gamma=0.99, gae_lambda=0.95, ent_coef=0.01, n_epochs=E, batch_size=B)
    rng = np.random.default_rng(5)
    eps = rng.standard_normal((ITERS, T, N, A)).astype(np.float32)
    perms = np.stack([[rng.permutation(T * N) for _ in range(E)] for _ in range(ITERS)])
    # oracle
    po = {k: v.copy() for k, v in p0.items()}
    st = O.AdamState.zeros_like(po)
    env_a = O.NumpySyntheticVecEnv(N, D, A, p_term=0.08, time_limit=9, seed=1)
    obs_a, starts = env_a.reset(), np.ones(N, bool)
    for it in range(ITERS):
        buf, obs_a, starts = O.collect_rollout(po, env_a, obs_a, starts, T, h, lambda t: eps[it, t])
        O.train(po, st, buf, h, perms[it])
    # engine, host path
    e = make_engine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=B, n_epochs=E, pi=(H, H), vf=(H, H),
                    gamma=h.gamma, gae_lambda=h.gae_lambda, ent_coef=h.ent_coef)
    e.set_params(p0)
    env_b = O.NumpySyntheticVecEnv(N, D, A, p_term=0.08, time_limit=9, seed=1)
    obs = env_b.reset()
    for it in range(ITERS):
        e.rollout_begin()
        for t in range(T):
            _, a_clip, _, _ = e.act(obs, eps[it, t])
            obs, rew, done, trunc, term_obs = env_b.step(a_clip)
            e.store(rew, done, trunc, term_obs)
        e.finish_rollout(obs, done)
        e.train(perms[it])
    got = e.get_params()
    for k in po:
        assert np.max(np.abs(got[k] - po[k])) < 1e-4, (k, float(np.max(np.abs(got[k] - po[k]))))
    m, v, step = e.get_optimizer_state()
    assert step == ITERS * E * (T * N // B)
    for k in po:
        assert scaled_err(m[k], st.exp_avg[k]) < 1e-3 and scaled_err(v[k], st.exp_avg_sq[k]) < 1e-3, k
    e.close()


def test_pinned_zero_copy_path_equals_staged_path():
    """act/store with hipHostMalloc buffers (kernels read / write them in place) == the staged-copy path."""
    D, A, N, T = 43, 2, 70, 9
    p = O.init_params(D, A, seed=2)
    rng = np.random.default_rng(1)
    obs_seq = rng.standard_normal((T + 1, N, D)).astype(np.float32)
    rew_seq = rng.standard_normal((T, N)).astype(np.float32)
    done_seq = rng.random((T, N)) < 0.2
    trunc_seq = done_seq & (rng.random((T, N)) < 0.5)
    term_seq = rng.standard_normal((T, N, D)).astype(np.float32)
    out = {}
    for pinned in (False, True):
        e = make_engine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=64, n_epochs=1, seed=6)
        e.set_params(p)
        mk = (lambda shape, dt=np.float32: e.pinned(shape, dt)) if pinned else (lambda shape, dt=np.float32: np.zeros(shape, dt))
        ob, cl, rw, dn, tr, tm = mk((N, D)), mk((N, A)), mk((N,)), mk((N,), np.uint8), mk((N,), np.uint8), mk((N, D))
        e.rollout_begin()
        clips = []
        for t in range(T):
            ob[:] = obs_seq[t]
            e.act(ob, out_clipped=cl, want_all=False)
            clips.append(cl.copy())
            rw[:], dn[:], tr[:], tm[:] = rew_seq[t], done_seq[t], trunc_seq[t], term_seq[t]
            e.store(rw, dn, tr if trunc_seq[t].any() else None, tm if trunc_seq[t].any() else None)
        ob[:] = obs_seq[T]
        e.finish_rollout(ob, dn)
        out[pinned] = {k: e.read(k) for k in ("obs", "actions", "rewards", "episode_starts", "values", "log_probs", "advantages")}
        out[pinned]["clips"] = np.stack(clips)
        e.close()
    for k in out[True]:
        assert np.array_equal(out[True][k], out[False][k]), k
    assert np.array_equal(out[True]["clips"], np.clip(out[True]["actions"], -1, 1))


@pytest.mark.parametrize("robot,hidden,N,parts", [("doggo", 256, 100, 3), ("point", 64, 96, 2), ("car", 48, 70, 2),
                                                  ("doggo", 256, 7, 7)])
def test_pipelined_part_rollout_equals_whole_batch_rollout(robot, hidden, N, parts):
    """act_part / wait_part / store_part over row ranges (the env steps one range while the GPU runs the policy for
    the others) == act / store over all rows, bit for bit: same noise per env and step, same bootstrap, same GAE.
    Shapes cover the H=256 and H=64 fused kernels and the generic path, ragged ranges and one env per part."""
    from mobrob_amd.envs.native_env import NativeGoalVecEnv
    from mobrob_amd.envs.wrapper import ROBOT_DIMS
    D, A, _ = ROBOT_DIMS[robot]
    T = 23
    p = O.init_params(D, A, (hidden, hidden), (hidden, hidden), seed=4)
    keys = ("obs", "actions", "rewards", "episode_starts", "values", "log_probs", "advantages", "returns", "last_values")
    out = {}
    for mode in ("whole", "parts", "native", "native-threads"):
        e = make_engine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=64, n_epochs=1, seed=9,
                        pi=(hidden, hidden), vf=(hidden, hidden))
        e.set_params(p)
        env = NativeGoalVecEnv.for_robot(robot, N, time_limit=6, seed=3)  # short episodes: truncations every rollout
        b = dict(obs=e.pinned((N, D)), clip=e.pinned((N, A)), rew=e.pinned((N,)), done=e.pinned((N,), np.uint8),
                 trunc=e.pinned((N,), np.uint8), term=e.pinned((N, D)))
        env.use_buffers(obs=b["obs"], rewards=b["rew"], dones=b["done"], truncated=b["trunc"], terminal_obs=b["term"])
        env.reset()
        for _ in range(2):  # two rollouts: the noise counter carries over
            e.rollout_begin()
            if mode == "whole":
                for _t in range(T):
                    e.act(b["obs"], out_clipped=b["clip"], want_all=False)
                    nt = env.step_arrays(b["clip"])[5]
                    e.store(b["rew"], b["done"], b["trunc"] if nt else None, b["term"] if nt else None)
            elif mode.startswith("native"):  # the same pipeline as one C call (mobrob_ppo_collect_host), finish_rollout included;
                # "-threads": the opt-in two-thread collector (a driver thread owns the stream, the caller steps the env)
                os.environ["MOBROB_COLLECT_THREADS"] = "1" if mode.endswith("threads") else "0"
                try:
                    e.part_pipeline(parts, b["obs"], b["clip"], b["rew"], b["done"], b["trunc"], b["term"]).collect(
                        env.step_range_fn, env.handle)
                finally:
                    os.environ.pop("MOBROB_COLLECT_THREADS", None)
                continue
            else:
                pipe = e.part_pipeline(parts, b["obs"], b["clip"], b["rew"], b["done"], b["trunc"], b["term"])
                assert pipe.bounds[0][0] == 0 and pipe.bounds[-1][1] == N
                for q in range(parts):
                    pipe.act(q)
                for t in range(T):
                    for q in range(parts):
                        pipe.wait(q)
                        nt = env.step_range(*pipe.bounds[q], b["clip"])
                        pipe.store(q, nt > 0)
                        if t + 1 < T:
                            pipe.act(q)
            e.finish_rollout(b["obs"], b["done"])
        out[mode] = {k: e.read(k) for k in keys}
        out[mode]["stats"] = env.episode_stats()
        if mode == "parts":  # protocol errors are reported, not executed
            e.rollout_begin()
            with pytest.raises(Exception):
                pipe.store(0)                      # nothing acted yet
            pipe.act(0)
            with pytest.raises(Exception):
                pipe.act(0)                        # step 0 of part 0 not stored yet
            with pytest.raises(Exception):
                e.part_pipeline(parts + 1, b["obs"], b["clip"], b["rew"], b["done"], b["trunc"], b["term"]).act(0)
            with pytest.raises(ValueError):
                e.part_pipeline(parts, np.zeros((N, D), np.float32), b["clip"], b["rew"], b["done"], b["trunc"], b["term"]).act(1)  # pageable
        env.close()
        e.close()
    assert out["whole"]["stats"]["episodes"] > N  # truncations and goals happened
    for mode in ("parts", "native", "native-threads"):
        for k in keys:
            assert np.array_equal(out["whole"][k], out[mode][k]), (mode, k)
        sw, sp = out["whole"]["stats"], out[mode]["stats"]
        assert (sw["episodes"], sw["goals"]) == (sp["episodes"], sp["goals"]) and abs(sw["ep_rew_mean"] - sp["ep_rew_mean"]) < 1e-9


@pytest.mark.parametrize("H", [256, 64])
@pytest.mark.parametrize("robot,N,parts", [("doggo", 192, 2), ("doggo", 64, 1), ("point", 128, 4), ("car", 96, 3),
                                           ("turtlebot3", 64, 2), ("drone", 128, 2)])
def test_served_host_rollout_equals_the_launch_per_step_rollout(robot, N, parts, H):
    """mobrob_ppo_collect_host on a 256-wide x3 engine: the persistent rollout kernel serves the host environment (flags in pinned
    memory, no launch and no event inside the step loop; kernels_rollout.h KIND 3) against the launch-per-step collector
    (MOBROB_COLLECT_SERVER=0: act_part / store_part per row range and step).  Same Philox counters, same sampling / storage /
    time-limit bootstrap code; the policy forward is the rollout kernel's split-bf16 one where the launch-per-step path runs
    k_fused_act on the f32 pipe, so the two agree to float32 rounding (as the device rollout and its per-step form do:
    test_persistent_rollout_equals_per_step_rollout), not in bits -- and the quantities the kernel only moves are exact: the
    clipped actions handed to the host, the rewards / observations taken from it.  Short episodes: truncations in every rollout.
    Four padded observation widths, one to four row ranges, two rollouts each (noise counter and episode-start flags carry over).
    MOBROB_COLLECT_SERVER=2 makes the engine refuse to fall back, so the served path is what ran.
    H = 64 (round 6: k_rollout64_tile<.., 3>, the networks of every reference YAML): the tile kernel's forward is the one-wave
    kernel's MFMA sequence per accumulator, so there the two collectors agree BIT FOR BIT."""
    from mobrob_amd.envs.native_env import NativeGoalVecEnv
    from mobrob_amd.envs.wrapper import ROBOT_DIMS
    D, A, _ = ROBOT_DIMS[robot]
    T = 37
    p = O.init_params(D, A, (H, H), (H, H), seed=6)
    keys = ("obs", "actions", "rewards", "episode_starts", "values", "log_probs", "advantages", "returns", "last_values")
    out = {}
    for mode in ("0", "2"):
        e = make_engine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=64, n_epochs=1, seed=11, pi=(H, H), vf=(H, H))
        e.set_params(p)
        env = NativeGoalVecEnv.for_robot(robot, N, time_limit=5, seed=7)
        b = dict(obs=e.pinned((N, D)), clip=e.pinned((N, A)), rew=e.pinned((N,)), done=e.pinned((N,), np.uint8),
                 trunc=e.pinned((N,), np.uint8), term=e.pinned((N, D)))
        env.use_buffers(obs=b["obs"], rewards=b["rew"], dones=b["done"], truncated=b["trunc"], terminal_obs=b["term"])
        env.reset()
        os.environ["MOBROB_COLLECT_SERVER"] = mode
        os.environ["MOBROB_SERVER_TIMEOUT_S"] = "5"
        try:
            res = []
            for _ in range(2):
                e.rollout_begin()
                e.part_pipeline(parts, b["obs"], b["clip"], b["rew"], b["done"], b["trunc"], b["term"]).collect(env.step_range_fn, env.handle)
                r = {k: e.read(k) for k in keys}
                # what the kernel only moves is exact: the host saw clip(actions) of the last step; the last observations and the
                # rewards of rows that were not truncated in the last step are the host's
                assert np.array_equal(b["clip"], np.clip(r["actions"][-1], -1, 1))
                assert np.array_equal(r["obs"][T], b["obs"])
                keep = b["trunc"] == 0
                assert np.array_equal(r["rewards"][-1][keep], b["rew"][keep])
                res.append(r)
        finally:
            os.environ.pop("MOBROB_COLLECT_SERVER", None)
            os.environ.pop("MOBROB_SERVER_TIMEOUT_S", None)
        out[mode] = res
        out[mode + "stats"] = env.episode_stats()
        env.close()
        e.close()
    assert out["0stats"]["episodes"] > N
    assert out["0stats"]["episodes"] == out["2stats"]["episodes"] and out["0stats"]["goals"] == out["2stats"]["goals"]
    tol = dict(obs=2e-5, actions=2e-5, rewards=2e-5, values=1e-4, log_probs=2e-4, advantages=1e-3, returns=1e-3, last_values=1e-4)
    for r in range(2):
        assert np.array_equal(out["0"][r]["episode_starts"], out["2"][r]["episode_starts"])
        for k, bound in tol.items():
            if H == 64:
                assert np.array_equal(out["0"][r][k], out["2"][r][k]), (r, k)
            else:
                assert np.max(np.abs(out["0"][r][k] - out["2"][r][k])) < bound, (r, k)


class _RecordingStepRange:
    """A mobrob_env_step_range_fn that forwards to a native environment's and keeps a copy of what the host handed over per step and
    row range (rewards, done / truncated flags, terminal and next observations): what an oracle check of a host-env rollout needs."""

    def __init__(self, env, b, T, N, D):
        import ctypes
        self.FN = ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
                                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p)
        self.inner = ctypes.cast(ctypes.c_void_p(env.step_range_fn), self.FN)
        self.b, self.T, self.N, self.D = b, T, N, D
        self.cb = self.FN(self._call)
        self.address = ctypes.cast(self.cb, ctypes.c_void_p).value
        self.begin()

    def begin(self):
        T, N, D = self.T, self.N, self.D
        self.rew, self.done, self.trunc = np.zeros((T, N), np.float32), np.zeros((T, N), bool), np.zeros((T, N), bool)
        self.term, self.nxt = np.zeros((T, N, D), np.float32), np.zeros((T, N, D), np.float32)
        self.clip = np.zeros((T, N, self.b["clip"].shape[1]), np.float32)
        self.steps = {}
        self.obs0 = self.b["obs"].copy()

    def _call(self, h, i0, i1, a, o, r, d, tr, to):
        t = self.steps.get(i0, 0)
        self.clip[t, i0:i1] = self.b["clip"][i0:i1]          # the clipped actions the host is about to step with
        rc = self.inner(h, i0, i1, a, o, r, d, tr, to)
        self.rew[t, i0:i1], self.done[t, i0:i1] = self.b["rew"][i0:i1], self.b["done"][i0:i1] != 0
        self.trunc[t, i0:i1] = (self.b["trunc"][i0:i1] != 0) if rc > 0 else False
        self.term[t, i0:i1], self.nxt[t, i0:i1] = self.b["term"][i0:i1], self.b["obs"][i0:i1]
        self.steps[i0] = t + 1
        return rc


def _check_host_rollout_against_oracle(e, p, rec, first_starts, gamma=0.99, lam=0.95):
    """A host-env rollout held to the ORACLE at north_star's 1e-4: values / log-probs of the stored observations and actions
    (O.policy_outputs, O.gaussian_log_prob), what the kernel only moves exactly (observations, rewards, episode starts, clipped
    actions), the time-limit bootstrap of truncated rows (O.bootstrap_reward on V(terminal observation)), and GAE on the stored
    arrays bit for bit (O.gae).  Returns the episode-start flags the NEXT rollout must begin with."""
    T, N, D = rec.T, rec.N, rec.D
    r = {k: e.read(k) for k in ("obs", "actions", "rewards", "episode_starts", "values", "log_probs", "advantages", "returns",
                                 "last_values", "last_dones")}
    assert np.array_equal(r["obs"][0], rec.obs0) and np.array_equal(r["obs"][1:T + 1], rec.nxt)
    assert np.array_equal(rec.clip, np.clip(r["actions"], -1, 1))
    starts = np.concatenate([first_starts[None].astype(np.float32), rec.done[:-1].astype(np.float32)])
    assert np.array_equal(r["episode_starts"], starts)
    mean, val = O.policy_outputs(p, r["obs"][:T].reshape(T * N, D))
    assert np.max(np.abs(r["values"].reshape(-1) - val)) < 1e-4 * max(1.0, float(np.max(np.abs(val))))
    lp = O.gaussian_log_prob(mean, p["log_std"], r["actions"].reshape(T * N, -1))
    assert np.max(np.abs(r["log_probs"].reshape(-1) - lp)) < 1e-4 * max(1.0, float(np.max(np.abs(lp))))
    _, lastv = O.policy_outputs(p, r["obs"][T])
    assert np.max(np.abs(r["last_values"] - lastv)) < 1e-4 * max(1.0, float(np.max(np.abs(lastv))))
    keep = ~rec.trunc
    assert np.array_equal(r["rewards"][keep], rec.rew[keep])
    if rec.trunc.any():
        tt, nn = np.nonzero(rec.trunc)
        _, tv = O.policy_outputs(p, rec.term[tt, nn])
        want = np.array([O.bootstrap_reward(rec.rew[t, n], gamma, v) for t, n, v in zip(tt, nn, tv)], np.float32)
        assert np.max(np.abs(r["rewards"][tt, nn] - want)) < 1e-4 * max(1.0, float(np.max(np.abs(want))))
    assert np.array_equal(r["last_dones"] > 0, rec.done[-1])
    adv, ret = O.gae(r["rewards"], r["values"], r["episode_starts"], r["last_values"], rec.done[-1], gamma, lam)
    assert np.array_equal(r["advantages"], adv) and np.array_equal(r["returns"], ret)
    return rec.done[-1].copy()


@pytest.mark.parametrize("robot,H,N,parts", [("doggo", 256, 192, 2), ("point", 256, 128, 4), ("car", 256, 96, 3), ("drone", 256, 64, 1),
                                             ("doggo", 64, 192, 2), ("point", 64, 1024, 2), ("car", 64, 96, 3), ("turtlebot3", 64, 64, 1),
                                             ("drone", 64, 128, 4),
                                             # ONE row range takes any number of environments: the reference YAMLs' 2 - 16 (half a tile)
                                             ("doggo", 64, 16, 1), ("point", 64, 2, 1), ("doggo", 256, 100, 1), ("car", 256, 7, 1)])
def test_served_host_rollout_matches_the_oracle(robot, H, N, parts):
    """The SERVED host collector (the rollout kernel hands actions over and pulls the host's step through pinned flag words:
    kernels_rollout.h KIND 3; MOBROB_COLLECT_SERVER=2 refuses to fall back) against the oracle, not against another HIP path:
    SB3's collect_rollouts [reached through /root/reference/src/mobrob/rl_control/ppo.py:73-74] on the VecEnv contract of
    /root/reference/src/mobrob/envs/wrapper.py:156-201.  Two rollouts (episode starts and the noise counter carry over), short
    episodes (truncations with a bootstrap in every rollout)."""
    from mobrob_amd.envs.native_env import NativeGoalVecEnv
    from mobrob_amd.envs.wrapper import ROBOT_DIMS
    D, A, _ = ROBOT_DIMS[robot]
    T = 29
    p = O.init_params(D, A, (H, H), (H, H), seed=8)
    e = make_engine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=64, n_epochs=1, seed=5, pi=(H, H), vf=(H, H))
    e.set_params(p)
    env = NativeGoalVecEnv.for_robot(robot, N, time_limit=5, seed=2)
    b = dict(obs=e.pinned((N, D)), clip=e.pinned((N, A)), rew=e.pinned((N,)), done=e.pinned((N,), np.uint8),
             trunc=e.pinned((N,), np.uint8), term=e.pinned((N, D)))
    env.use_buffers(obs=b["obs"], rewards=b["rew"], dones=b["done"], truncated=b["trunc"], terminal_obs=b["term"])
    env.reset()
    rec = _RecordingStepRange(env, b, T, N, D)
    os.environ["MOBROB_COLLECT_SERVER"] = "2"
    os.environ["MOBROB_SERVER_TIMEOUT_S"] = "5"
    try:
        starts = np.ones(N, bool)
        saw_trunc = False
        for _ in range(2):
            rec.begin()
            e.rollout_begin()
            e.part_pipeline(parts, b["obs"], b["clip"], b["rew"], b["done"], b["trunc"], b["term"]).collect(rec.address, env.handle)
            saw_trunc |= bool(rec.trunc.any())
            starts = _check_host_rollout_against_oracle(e, p, rec, starts)
        assert saw_trunc
    finally:
        os.environ.pop("MOBROB_COLLECT_SERVER", None)
        os.environ.pop("MOBROB_SERVER_TIMEOUT_S", None)
    env.close()
    e.close()


@pytest.mark.parametrize("H", [64, 256])
def test_served_host_rollout_from_a_registered_block_matches_the_oracle(H):
    """The served collector on caller-owned memory pinned in place (mobrob_ppo_host_register = hipHostRegister: what ShmVecEnv's shared
    block is) instead of mobrob_ppo_host_alloc buffers: accepted (registered blocks are coherent unless HIP_HOST_COHERENT=0, which
    collect_host_served refuses by name) and held to the oracle like the allocated ones -- the hand-over's acquire is what makes the
    pull see the host's writes whatever the block's caching policy."""
    import mmap
    from mobrob_amd.envs.native_env import NativeGoalVecEnv
    from mobrob_amd.envs.wrapper import ROBOT_DIMS
    robot, N, parts, T = "doggo", 128, 2, 21
    D, A, _ = ROBOT_DIMS[robot]
    p = O.init_params(D, A, (H, H), (H, H), seed=3)
    e = make_engine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=64, n_epochs=1, seed=4, pi=(H, H), vf=(H, H))
    e.set_params(p)
    sizes = dict(obs=N * D * 4, clip=N * A * 4, rew=N * 4, done=N, trunc=N, term=N * D * 4)
    offs, total = {}, 0
    for k, n in sizes.items():
        offs[k] = total
        total += (n + 255) // 256 * 256
    block = mmap.mmap(-1, (total + 4095) // 4096 * 4096)          # page-aligned anonymous memory
    base = np.frombuffer(block, np.uint8)
    e.register_host(base.ctypes.data, len(block))
    view = lambda k, shape, dt: base[offs[k]:offs[k] + sizes[k]].view(dt).reshape(shape)   # noqa: E731
    b = dict(obs=view("obs", (N, D), np.float32), clip=view("clip", (N, A), np.float32), rew=view("rew", (N,), np.float32),
             done=view("done", (N,), np.uint8), trunc=view("trunc", (N,), np.uint8), term=view("term", (N, D), np.float32))
    env = NativeGoalVecEnv.for_robot(robot, N, time_limit=5, seed=9)
    env.use_buffers(obs=b["obs"], rewards=b["rew"], dones=b["done"], truncated=b["trunc"], terminal_obs=b["term"])
    env.reset()
    rec = _RecordingStepRange(env, b, T, N, D)
    os.environ["MOBROB_COLLECT_SERVER"] = "2"
    os.environ["MOBROB_SERVER_TIMEOUT_S"] = "5"
    try:
        starts = np.ones(N, bool)
        for _ in range(2):
            rec.begin()
            e.rollout_begin()
            e.part_pipeline(parts, b["obs"], b["clip"], b["rew"], b["done"], b["trunc"], b["term"]).collect(rec.address, env.handle)
            starts = _check_host_rollout_against_oracle(e, p, rec, starts)
        # while HIP_HOST_COHERENT=0 a registered block (no hipHostMallocCoherent flag) counts as non-coherent: refused by name; buffers from
        # mobrob_ppo_host_alloc carry the flag and stay eligible
        os.environ["HIP_HOST_COHERENT"] = "0"
        try:
            e.rollout_begin()
            with pytest.raises(Exception, match="non-coherent"):
                e.part_pipeline(parts, b["obs"], b["clip"], b["rew"], b["done"], b["trunc"], b["term"]).collect(rec.address, env.handle)
        finally:
            os.environ.pop("HIP_HOST_COHERENT", None)
        # a buffer whose END lies outside the pinned block is refused by name (the first byte alone says nothing about [N][D])
        tail = np.zeros((N, D), np.float32)
        e.rollout_begin()
        with pytest.raises(Exception, match="pageable|pinned"):
            e.part_pipeline(parts, b["obs"], b["clip"], b["rew"], b["done"], b["trunc"], tail).collect(rec.address, env.handle)
    finally:
        os.environ.pop("MOBROB_COLLECT_SERVER", None)
        os.environ.pop("MOBROB_SERVER_TIMEOUT_S", None)
    env.close()
    e.unregister_host(base.ctypes.data)
    e.close()


@pytest.mark.parametrize("H", [64, 256])
def test_served_collector_falls_back_when_a_workgroup_is_not_resident(H):
    """The residency check in front of the first environment step (a workgroup of the serving kernel that is not resident -- another
    tenant on the device -- would leave the others waiting for the host while the host waits for it): on a miss the launches are told
    to stop and the launch-per-step collector runs from the untouched rollout state.  MOBROB_SERVER_RESIDENCY_S=0 makes the check
    miss; the rollout must then equal the MOBROB_COLLECT_SERVER=0 rollout bit for bit (the same kernels ran), twice in a row, and
    MOBROB_COLLECT_SERVER=2 must name the reason instead."""
    from mobrob_amd.envs.native_env import NativeGoalVecEnv
    from mobrob_amd.envs.wrapper import ROBOT_DIMS
    robot, N, parts, T = "car", 128, 2, 19
    D, A, _ = ROBOT_DIMS[robot]
    p = O.init_params(D, A, (H, H), (H, H), seed=1)
    keys = ("obs", "actions", "rewards", "episode_starts", "values", "log_probs", "advantages", "returns", "last_values")
    out = {}
    for mode, res_s in (("0", None), ("1", "0")):
        e = make_engine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=64, n_epochs=1, seed=3, pi=(H, H), vf=(H, H))
        e.set_params(p)
        env = NativeGoalVecEnv.for_robot(robot, N, time_limit=5, seed=4)
        b = dict(obs=e.pinned((N, D)), clip=e.pinned((N, A)), rew=e.pinned((N,)), done=e.pinned((N,), np.uint8),
                 trunc=e.pinned((N,), np.uint8), term=e.pinned((N, D)))
        env.use_buffers(obs=b["obs"], rewards=b["rew"], dones=b["done"], truncated=b["trunc"], terminal_obs=b["term"])
        env.reset()
        os.environ["MOBROB_COLLECT_SERVER"] = mode
        os.environ["MOBROB_SERVER_TIMEOUT_S"] = "5"
        if res_s is not None:
            os.environ["MOBROB_SERVER_RESIDENCY_S"] = res_s
        try:
            res = []
            for _ in range(2):
                e.rollout_begin()
                e.part_pipeline(parts, b["obs"], b["clip"], b["rew"], b["done"], b["trunc"], b["term"]).collect(env.step_range_fn, env.handle)
                res.append({k: e.read(k) for k in keys})
            if res_s is not None:
                os.environ["MOBROB_COLLECT_SERVER"] = "2"
                e.rollout_begin()
                with pytest.raises(Exception, match="not every workgroup"):
                    e.part_pipeline(parts, b["obs"], b["clip"], b["rew"], b["done"], b["trunc"], b["term"]).collect(env.step_range_fn, env.handle)
        finally:
            for k in ("MOBROB_COLLECT_SERVER", "MOBROB_SERVER_TIMEOUT_S", "MOBROB_SERVER_RESIDENCY_S"):
                os.environ.pop(k, None)
        out[mode] = res
        env.close()
        e.close()
    for r in range(2):
        for k in keys:
            assert np.array_equal(out["0"][r][k], out["1"][r][k]), (r, k)


def test_served_host_rollout_gives_up_when_the_environment_fails():
    """A step_range that reports an error in the middle of a served rollout: the host tells the waiting workgroups to stop, the
    queued launches return at once, the call fails -- and the engine serves the next rollout normally."""
    import ctypes
    from mobrob_amd.envs.native_env import NativeGoalVecEnv
    D, A, N, T = 58, 12, 64, 40
    e = make_engine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=64, n_epochs=1, seed=1, pi=(256, 256), vf=(256, 256))
    e.set_params(O.init_params(D, A, (256, 256), (256, 256), seed=2))
    env = NativeGoalVecEnv.for_robot("doggo", N, time_limit=50, seed=1)
    b = dict(obs=e.pinned((N, D)), clip=e.pinned((N, A)), rew=e.pinned((N,)), done=e.pinned((N,), np.uint8),
             trunc=e.pinned((N,), np.uint8), term=e.pinned((N, D)))
    env.use_buffers(obs=b["obs"], rewards=b["rew"], dones=b["done"], truncated=b["trunc"], terminal_obs=b["term"])
    env.reset()
    calls = [0]
    FN = ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
                          ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p)
    inner = ctypes.cast(ctypes.c_void_p(env.step_range_fn), FN)

    def failing(h, i0, i1, a, o, r, d, tr, to):
        calls[0] += 1
        return -7 if calls[0] == 25 else inner(h, i0, i1, a, o, r, d, tr, to)
    cb = FN(failing)
    os.environ["MOBROB_COLLECT_SERVER"] = "2"
    os.environ["MOBROB_SERVER_TIMEOUT_S"] = "5"
    try:
        pipe = e.part_pipeline(2, b["obs"], b["clip"], b["rew"], b["done"], b["trunc"], b["term"])
        e.rollout_begin()
        with pytest.raises(Exception, match="step_range returned -7"):
            pipe.collect(ctypes.cast(cb, ctypes.c_void_p).value, env.handle)
        e.rollout_begin()
        pipe.collect(env.step_range_fn, env.handle)     # and the next rollout is served
        assert np.isfinite(e.read("advantages")).all()
    finally:
        os.environ.pop("MOBROB_COLLECT_SERVER", None)
        os.environ.pop("MOBROB_SERVER_TIMEOUT_S", None)
    env.close()
    e.close()


def test_served_host_rollout_is_bounded_when_the_host_stalls_and_refuses_shapes_it_cannot_serve():
    """The waits of the served collector are bounded on BOTH sides: an environment that stalls longer than MOBROB_SERVER_TIMEOUT_S
    makes the waiting workgroups give up (error word in pinned memory, abort word for the queued launches) and the call fails
    instead of hanging the device; the next rollout is served again.  And MOBROB_COLLECT_SERVER=2 names the reason when a shape
    cannot be served (row ranges that are not whole 32-row tiles) instead of silently taking the launch-per-step path."""
    import ctypes
    import time as _time
    from mobrob_amd.envs.native_env import NativeGoalVecEnv
    D, A, N, T = 58, 12, 64, 12
    e = make_engine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=64, n_epochs=1, seed=1, pi=(256, 256), vf=(256, 256))
    e.set_params(O.init_params(D, A, (256, 256), (256, 256), seed=2))
    env = NativeGoalVecEnv.for_robot("doggo", N, time_limit=50, seed=1)
    b = dict(obs=e.pinned((N, D)), clip=e.pinned((N, A)), rew=e.pinned((N,)), done=e.pinned((N,), np.uint8),
             trunc=e.pinned((N,), np.uint8), term=e.pinned((N, D)))
    env.use_buffers(obs=b["obs"], rewards=b["rew"], dones=b["done"], truncated=b["trunc"], terminal_obs=b["term"])
    env.reset()
    calls = [0]
    FN = ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
                          ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p)
    inner = ctypes.cast(ctypes.c_void_p(env.step_range_fn), FN)

    def stalling(h, i0, i1, a, o, r, d, tr, to):
        calls[0] += 1
        if calls[0] == 5:
            _time.sleep(1.5)
        return inner(h, i0, i1, a, o, r, d, tr, to)
    cb = FN(stalling)
    os.environ["MOBROB_COLLECT_SERVER"] = "2"
    os.environ["MOBROB_SERVER_TIMEOUT_S"] = "0.3"
    try:
        pipe = e.part_pipeline(2, b["obs"], b["clip"], b["rew"], b["done"], b["trunc"], b["term"])
        e.rollout_begin()
        t0 = _time.time()
        with pytest.raises(Exception, match="gave up waiting for the host"):
            pipe.collect(ctypes.cast(cb, ctypes.c_void_p).value, env.handle)
        assert _time.time() - t0 < 20
        os.environ["MOBROB_SERVER_TIMEOUT_S"] = "5"
        e.rollout_begin()
        pipe.collect(env.step_range_fn, env.handle)
        assert np.isfinite(e.read("advantages")).all()
        e.rollout_begin()
        with pytest.raises(Exception, match="not whole 32-row tiles"):
            e.part_pipeline(4, b["obs"], b["clip"], b["rew"], b["done"], b["trunc"], b["term"]).collect(env.step_range_fn, env.handle)
    finally:
        os.environ.pop("MOBROB_COLLECT_SERVER", None)
        os.environ.pop("MOBROB_SERVER_TIMEOUT_S", None)
    env.close()
    e.close()


def test_served_rollouts_of_two_engines_share_the_compute_units_through_a_lease():
    """Every workgroup of a served rollout stays resident until the host has stepped all n_steps, so two engines of one process
    that collect at the same time must fit the device's compute units TOGETHER: the second one (160 + 160 tiles > 256 CUs) is told
    so in required mode (and takes the launch-per-step path otherwise) instead of queueing behind a kernel that waits for a host."""
    import ctypes
    import threading
    import time as _time
    from mobrob_amd.envs.native_env import NativeGoalVecEnv
    D, A, N, T = 14, 2, 32 * 160, 6
    p = O.init_params(D, A, (256, 256), (256, 256), seed=3)
    FN = ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
                          ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p)
    rigs = []
    for k in range(2):
        e = make_engine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=1024, n_epochs=1, seed=k, pi=(256, 256), vf=(256, 256))
        e.set_params(p)
        env = NativeGoalVecEnv.for_robot("point", N, time_limit=50, seed=k)
        b = dict(obs=e.pinned((N, D)), clip=e.pinned((N, A)), rew=e.pinned((N,)), done=e.pinned((N,), np.uint8),
                 trunc=e.pinned((N,), np.uint8), term=e.pinned((N, D)))
        env.use_buffers(obs=b["obs"], rewards=b["rew"], dones=b["done"], truncated=b["trunc"], terminal_obs=b["term"])
        env.reset()
        rigs.append((e, env, b))
    inner = ctypes.cast(ctypes.c_void_p(rigs[0][1].step_range_fn), FN)
    first = [True]

    def slow_first_step(h, i0, i1, a, o, r, d, tr, to):
        if first[0]:
            first[0] = False
            _time.sleep(0.6)          # engine 0's kernel is resident and waiting for this while engine 1 asks for its lease
        return inner(h, i0, i1, a, o, r, d, tr, to)
    cb = FN(slow_first_step)
    res = {}

    def run(k, fn):
        e, env, b = rigs[k]
        try:
            e.rollout_begin()
            e.part_pipeline(1 if k == 0 else 2, b["obs"], b["clip"], b["rew"], b["done"], b["trunc"], b["term"]).collect(fn, env.handle)
            res[k] = "ok"
        except Exception as ex:  # noqa: BLE001
            res[k] = str(ex)
    os.environ["MOBROB_COLLECT_SERVER"] = "2"
    os.environ["MOBROB_SERVER_TIMEOUT_S"] = "10"
    try:
        t0 = threading.Thread(target=run, args=(0, ctypes.cast(cb, ctypes.c_void_p).value))
        t0.start()
        _time.sleep(0.2)
        run(1, rigs[1][1].step_range_fn)
        t0.join()
        assert res[0] == "ok", res
        assert "serving another engine" in res[1], res
        os.environ["MOBROB_COLLECT_SERVER"] = "1"      # not required: the same situation falls back and completes
        first[0] = True
        t0 = threading.Thread(target=run, args=(0, ctypes.cast(cb, ctypes.c_void_p).value))
        t0.start()
        _time.sleep(0.2)
        run(1, rigs[1][1].step_range_fn)
        t0.join()
        assert res[0] == "ok" and res[1] == "ok", res
        assert np.isfinite(rigs[1][0].read("advantages")).all() and np.isfinite(rigs[0][0].read("advantages")).all()
    finally:
        os.environ.pop("MOBROB_COLLECT_SERVER", None)
        os.environ.pop("MOBROB_SERVER_TIMEOUT_S", None)
    for e, env, _ in rigs:
        env.close()
        e.close()


def test_part_rollout_protocol_and_counter_continuity():
    """One part == the whole batch; a rollout cannot be finished while a part lags; and a whole-batch rollout after a
    pipelined one continues the same noise sequence as after a whole-batch one."""
    from mobrob_amd.envs.native_env import NativeGoalVecEnv
    D, A, N, T = 14, 2, 40, 6
    p = O.init_params(D, A, seed=8)
    res = {}
    for first in ("whole", "one_part", "two_parts"):
        e = make_engine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=40, n_epochs=1, seed=2)
        e.set_params(p)
        env = NativeGoalVecEnv.for_robot("point", N, time_limit=4, seed=1)
        b = dict(obs=e.pinned((N, D)), clip=e.pinned((N, A)), rew=e.pinned((N,)), done=e.pinned((N,), np.uint8),
                 trunc=e.pinned((N,), np.uint8), term=e.pinned((N, D)))
        env.use_buffers(obs=b["obs"], rewards=b["rew"], dones=b["done"], truncated=b["trunc"], terminal_obs=b["term"])
        env.reset()

        def whole():
            e.rollout_begin()
            for _ in range(T):
                e.act(b["obs"], out_clipped=b["clip"], want_all=False)
                nt = env.step_arrays(b["clip"])[5]
                e.store(b["rew"], b["done"], b["trunc"] if nt else None, b["term"] if nt else None)
            e.finish_rollout(b["obs"], b["done"])

        if first == "whole":
            whole()
        else:
            parts = 1 if first == "one_part" else 2
            pipe = e.part_pipeline(parts, b["obs"], b["clip"], b["rew"], b["done"], b["trunc"], b["term"])
            e.rollout_begin()
            for q in range(parts):
                pipe.act(q)
            for t in range(T):
                for q in range(parts):
                    pipe.wait(q)
                    nt = env.step_range(*pipe.bounds[q], b["clip"])
                    if t == T - 1 and q == parts - 1 and parts == 2:
                        with pytest.raises(Exception):  # the last part has not stored its last step yet
                            e.finish_rollout(b["obs"], b["done"])
                    pipe.store(q, nt > 0, pull_next_obs=(t % 2 == 0))  # both pull variants
                    if t + 1 < T:
                        pipe.act(q)
            e.finish_rollout(b["obs"], b["done"])
        first_ro = {k: e.read(k) for k in ("actions", "rewards", "advantages")}
        whole()  # second rollout: always the whole-batch loop
        res[first] = (first_ro, {k: e.read(k) for k in ("actions", "rewards", "advantages")})
        env.close()
        e.close()
    for mode in ("one_part", "two_parts"):
        for i in (0, 1):
            for k in res["whole"][i]:
                assert np.array_equal(res["whole"][i][k], res[mode][i][k]), (mode, i, k)


def test_synthetic_collect_statistics_and_consistency():
    """Device-resident env source: statistics of the generator and self-consistency of the stored rollout."""
    D, A, N, T = 58, 12, 512, 64
    p = O.init_params(D, A, seed=1)
    e = make_engine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=4096, n_epochs=1, seed=123)
    e.set_params(p)
    e.collect_synthetic(p_term=1 / 20.0, time_limit=30)
    e.synchronize()
    obs = e.read("obs")
    assert abs(obs.mean()) < 0.01 and abs(obs.std() - 1.0) < 0.01
    es = e.read("episode_starts")
    assert np.all(es[0] == 1.0)  # _last_episode_starts starts all-True
    rate = es[1:].mean()
    assert 0.03 < rate < 0.09  # p_term 0.05 (+ truncations)
    rew = e.read("rewards")
    assert 0.0 < np.median(rew) < 0.06
    # stored values / log-probs are the policy's outputs on the stored observations and actions
    flat = obs[:T].reshape(T * N, D)
    mean, val = O.policy_outputs(p, flat)
    acts = e.read("actions").reshape(T * N, A)
    assert scaled_err(e.read("values").reshape(-1), val) < 1e-4
    assert np.allclose(e.read("log_probs").reshape(-1), O.gaussian_log_prob(mean, p["log_std"], acts), rtol=1e-4, atol=1e-3)
    z = (acts - mean) / np.exp(p["log_std"])
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1.0) < 0.01  # Philox + Box-Muller eps
    adv, ret = O.gae(rew, e.read("values"), es, e.read("last_values"), e.read("last_dones") > 0, 0.99, 0.95)
    assert np.array_equal(e.read("advantages"), adv)
    # second rollout continues from the last observation of the first
    last = obs[T].copy()
    e.collect_synthetic(p_term=1 / 20.0, time_limit=30)
    assert np.array_equal(e.read("obs")[0], last)
    st = e.train(None)
    assert np.isfinite(st["loss"]) and st["n_minibatches"] == e.n_minibatches
    e.close()


@pytest.mark.parametrize("H", [64, 256, 32])  # H=64 / H=256 fused families and the generic path
def test_device_rollout_time_limit_bootstrap(H):
    """Truncated rows of the device env source: reward += gamma * V(terminal_obs), computed inside the env-step launch
    (SB3 collect_rollouts' TimeLimit.truncated branch; oracle bootstrap_reward)."""
    D, A, N, T, TL = 26, 2, 96, 20, 10
    p = O.init_params(D, A, (H, H), (H, H), seed=2)
    p["value_net.bias"] = np.array([7.0], np.float32)  # makes the bootstrap term visible: gamma * V ~ 6.9
    e = make_engine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=480, n_epochs=1, pi=(H, H), vf=(H, H), seed=9)
    e.set_params(p)
    e.collect_synthetic(p_term=0.0, time_limit=TL)  # no terminations: every env is truncated at steps 9 and 19
    e.synchronize()
    rew, es = e.read("rewards"), e.read("episode_starts")
    assert np.all(es[TL] == 1.0) and np.all(es[1:TL] == 0.0)
    tr = e.read("truncated")
    assert np.all(tr == 1)  # latest step (t = 19) truncated everywhere
    tobs = e.read("terminal_obs")[:, :D]
    _, v = O.policy_outputs(p, tobs)
    tv = e.read("terminal_values")
    assert scaled_err(tv, v) < 1e-4
    raw = rew[T - 1].astype(np.float64) - np.float64(np.float32(0.99) * tv)
    assert np.all(np.abs(raw - 0.03) < 0.6) and abs(raw.mean() - 0.03) < 0.05  # what is left is the N(0.03, 0.1^2) draw
    plain = np.delete(rew, [TL - 1, T - 1], axis=0)
    assert np.all(np.abs(plain - 0.03) < 0.6)
    assert np.all(rew[TL - 1] > 5.0) and np.all(rew[T - 1] > 5.0)
    # terminal observations are not the stored next observations (those are the reset draws)
    assert not np.allclose(e.read("obs")[T][:, :D], tobs)
    e.close()


@pytest.mark.parametrize("kind", ["synthetic", "goal"])
@pytest.mark.parametrize("H,D,A,N,T", [(256, 58, 12, 200, 24), (256, 26, 2, 64, 24), (256, 12, 18, 33, 24),
                                       (256, 43, 2, 6200, 18),   # 194 rollout blocks: no side-stream overlap, Dp = 48
                                       (256, 58, 12, 70, 40),    # 3 chunks of 16 steps + a short one
                                       (64, 14, 2, 100, 24), (64, 43, 2, 37, 24), (64, 58, 12, 300, 10)])
def test_persistent_rollout_equals_per_step_rollout(kind, H, D, A, N, T):
    """One persistent launch for all T steps vs one launch per step: same Philox counters, same arithmetic ->
    bit-identical rollout buffers (observations, actions, log-probs, rewards incl. bootstrap, episode starts);
    values come from the batched pass and agree to float tolerance; two consecutive rollouts (carried state)."""
    from mobrob_amd.envs.vec_env import DeviceGoalVecEnv
    TL = 7
    p = O.init_params(D, A, (H, H), (H, H), seed=8)
    p["value_net.bias"] = np.array([2.0], np.float32)
    res = {}
    for persistent in (True, False):
        # forward_x3=False: both kernels on the f32 matrix pipe (the persistent kernel's default is the split-bf16 form of the
        # hidden layers, equal to rounding only: test_x3_forward_kernels_are_float32_accurate)
        e = make_engine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=256, n_epochs=1, pi=(H, H), vf=(H, H), seed=21,
                        rollout_persistent=persistent, forward_x3=False)
        e.set_params(p)
        out = []
        for _ in range(2):
            if kind == "synthetic":
                e.collect_synthetic(p_term=0.05, time_limit=TL)
            else:
                DeviceGoalVecEnv(N, D, A, 2 if D < 12 or A != 18 else 3, time_limit=TL).collect(e)
            e.synchronize()
            out.append({k: e.read(k) for k in ("obs", "actions", "log_probs", "rewards", "episode_starts", "values",
                                               "last_values", "last_dones", "advantages", "truncated", "terminal_values")})
        res[persistent] = out
        e.close()
    for a, b in zip(res[True], res[False]):
        for k in ("obs", "actions", "log_probs", "episode_starts", "last_dones", "truncated"):
            assert np.array_equal(a[k], b[k]), k
        assert np.max(np.abs(a["rewards"] - b["rewards"])) < 1e-5  # bootstrapped rows: V through a different reduction order
        assert np.max(np.abs(a["values"] - b["values"])) < 1e-4 and np.max(np.abs(a["last_values"] - b["last_values"])) < 1e-4
        assert np.max(np.abs(a["advantages"] - b["advantages"])) < 1e-3
        assert a["episode_starts"][1:].sum() > 0  # the comparison covers resets and truncations


@pytest.mark.parametrize("D,A,N,T", [(58, 12, 200, 24), (26, 2, 64, 24), (12, 18, 33, 24), (43, 2, 300, 18)])
def test_x3_forward_kernels_are_float32_accurate(D, A, N, T):
    """The persistent rollout and the batched value pass of 256-wide nets multiply on the bf16 matrix pipe: float32 operands
    split into three bf16 pieces, six piece products, float32 accumulation (`forward_x3`, kernels_fused.h gemm_x3_r32 /
    gemm_x3_r64).  Claim: float32 RESULTS -- against a float64 evaluation of the same network on the stored observations the
    stored log-probs and values are as close as those of the f32-pipe kernels (same noise, so the actions are compared too)."""
    H = 256
    p = O.init_params(D, A, (H, H), (H, H), seed=8)
    rng = np.random.default_rng(3)
    p["log_std"] = rng.normal(-0.3, 0.2, A).astype(np.float32)
    p["action_net.weight"] *= 30          # means of order one
    p["value_net.weight"] *= 5
    p64 = type(p)((k, v.astype(np.float64)) for k, v in p.items())
    res = {}
    for x3 in (True, False):
        e = make_engine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=256, n_epochs=1, pi=(H, H), vf=(H, H), seed=21,
                        forward_x3=x3)
        e.set_params(p)
        e.collect_synthetic(p_term=0.05, time_limit=9)
        e.synchronize()
        buf = {k: e.read(k) for k in ("obs", "actions", "log_probs", "values")}
        e.close()
        obs = buf["obs"][:T].reshape(T * N, D).astype(np.float64)

        def mlp64(net):                                              # float64 reference, written out
            h = np.tanh(obs @ p64[f"mlp_extractor.{net}.0.weight"].T + p64[f"mlp_extractor.{net}.0.bias"])
            return np.tanh(h @ p64[f"mlp_extractor.{net}.2.weight"].T + p64[f"mlp_extractor.{net}.2.bias"])
        mean = mlp64("policy_net") @ p64["action_net.weight"].T + p64["action_net.bias"]
        val = (mlp64("value_net") @ p64["value_net.weight"].T + p64["value_net.bias"])[:, 0]
        sd = np.exp(p64["log_std"])
        d = buf["actions"].reshape(T * N, A).astype(np.float64) - mean
        lp = np.sum(-(d * d) / (2.0 * sd * sd) - np.log(sd) - 0.5 * np.log(2.0 * np.pi), axis=1)
        res[x3] = dict(buf=buf, lp_err=float(np.max(np.abs(buf["log_probs"].ravel() - lp))),
                       v_err=float(np.max(np.abs(buf["values"][:T].ravel() - val))),
                       v_scale=float(np.max(np.abs(val))), lp_scale=float(np.max(np.abs(lp))))
    a, b = res[True], res[False]
    # synthetic observations do not depend on the actions: both runs saw the same inputs and drew the same noise
    assert np.array_equal(a["buf"]["obs"], b["buf"]["obs"])
    assert np.max(np.abs(a["buf"]["actions"] - b["buf"]["actions"])) < 2e-5 * max(1.0, float(np.max(np.abs(b["buf"]["actions"]))))
    # accuracy against float64: the x3 kernels are held to the bound the f32-pipe kernels meet, and to 1.5x their actual error
    for key, scale in (("lp_err", "lp_scale"), ("v_err", "v_scale")):
        assert a[key] < 1e-5 * max(1.0, a[scale]), (key, a[key], a[scale])
        assert a[key] <= 1.5 * b[key] + 1e-7 * max(1.0, a[scale]), (key, a[key], b[key])


@pytest.mark.parametrize("D,A", [(58, 12), (26, 2), (14, 2)])
def test_x3_gradient_kernel_is_float32_accurate(D, A):
    """k_fused_train<.., X3>: forward of the hidden layers, dh1 and dW2 as six bf16 products of three-way split float32
    operands.  Against the oracle's float64-accumulated gradient of the same minibatch its error (scaled by the tensor's largest
    entry) is held to 2e-5 like the all-f32 kernel's and stays within a small factor of that kernel's actual error (both are ~1e-6:
    the rounding noise of 8192-term float32 sums), tensor by tensor."""
    H, T, N, B = 256, 32, 256, 8192
    rng = np.random.default_rng(11)
    p = O.init_params(D, A, (H, H), (H, H), seed=3)
    p["log_std"] = rng.normal(-0.3, 0.2, A).astype(np.float32)
    p["action_net.weight"] *= 30
    buf, lv, dones = _consistent_rollout(p, T, N, D, A, seed=9)
    h = O.Hyper(gamma=0.99, gae_lambda=0.95, ent_coef=0.01, n_epochs=1, batch_size=B, learning_rate=3e-4)
    buf["advantages"], buf["returns"] = O.gae(buf["rewards"], buf["values"], buf["episode_starts"], lv, dones, h.gamma, h.gae_lambda)
    idx = rng.permutation(T * N)
    _, og, aux = O.loss_and_grads(p, *O.gather_minibatch(buf, idx[:B]), h, acc=np.float64)
    lo, hi = 1.0 - h.clip_range, 1.0 + h.clip_range
    near = (np.abs(aux["ratio"] - lo) < 2e-5) | (np.abs(aux["ratio"] - hi) < 2e-5)
    if near.any():   # rows within float32 rounding of a clip boundary flip between two correct implementations: move them off it
        assert int(near.sum()) <= 4, int(near.sum())       # (a 4e-5-wide band holds ~1e-4 of 8192 rows: a handful at most)
        t, n = O.flat_to_tn(idx[:B][near], T)
        buf["log_probs"][t, n] -= np.float32(0.01)         # the engines below load the moved buffer
        _, og, aux = O.loss_and_grads(p, *O.gather_minibatch(buf, idx[:B]), h, acc=np.float64)
        assert not ((np.abs(aux["ratio"] - lo) < 2e-5) | (np.abs(aux["ratio"] - hi) < 2e-5)).any()
    errs = {}
    for x3 in (True, False):
        e = make_engine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=B, n_epochs=1, pi=(H, H), vf=(H, H),
                        gamma=h.gamma, gae_lambda=h.gae_lambda, ent_coef=h.ent_coef, learning_rate=h.learning_rate, forward_x3=x3)
        assert e.x3_mode() & 3 == (3 if x3 else 0)
        e.set_params(p)
        e.load_rollout(buf, lv, dones)
        e.epoch_begin(idx)
        e.minibatch_grad(0)
        got = e.unflatten(e.read("grads"))
        errs[x3] = {k: scaled_err(got[k], og[k]) for k in og}
        e.close()
    for k in og:
        assert errs[True][k] < 2e-5, (k, errs[True][k])
        assert errs[True][k] <= 4.0 * errs[False][k] + 1e-6, (k, errs[True][k], errs[False][k])  # both ~1e-6: rounding noise of 8192-term sums


def test_rollouts_beyond_4gib_use_the_64bit_generic_kernels():
    """Maximum sizes: the fused kernels address rollout rows with 32-bit byte offsets; an observation buffer of
    >= 4 GiB must fall back to the generic (64-bit indexed) kernels and still match the oracle."""
    D, A, H, N, T = 58, 12, 256, 8192, 2050   # obs: 2051 * 8192 * 64 * 4 B = 4.3 GB
    p = O.init_params(D, A, (H, H), (H, H), seed=5)
    e = make_engine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=65536, n_epochs=1, pi=(H, H), vf=(H, H), seed=2)
    e.set_params(p)
    # write one rollout step worth of data at the END of the buffers (beyond the 4 GiB mark) through the host path
    rng = np.random.default_rng(0)
    e.rollout_begin()
    obs = rng.standard_normal((N, D)).astype(np.float32)
    raw, clipped, values, logp = e.act(obs)
    mean, val = O.policy_outputs(p, obs)
    assert scaled_err(values, val) < 1e-4
    assert np.allclose(logp, O.gaussian_log_prob(mean, p["log_std"], raw), rtol=1e-4, atol=1e-3)
    # the profile ids of the fused path must stay silent: this engine runs the generic GEMM chain
    e.profile(True)
    e.act(obs)
    e.synchronize()
    assert e.profile_read()["act"][1] >= 1
    e.close()


def test_profile_phase_selection():
    """engine.profile(only=[...]) brackets just the named phases (what bench.py does for the dominant kernel); the
    per-phase call counts are what the roofline line divides by."""
    D, A, H, N, T, B, E = 14, 2, 64, 64, 16, 256, 2
    e = make_engine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=B, n_epochs=E, pi=(H, H), vf=(H, H), seed=3)
    e.set_params(O.init_params(D, A, (H, H), (H, H), seed=1))
    nmb = E * (N * T // B)
    e.profile(True, only=["train_grad"])
    e.collect_synthetic(p_term=0.1, time_limit=10)
    e.train(None)
    pr = e.profile_read()
    # (this shape's epochs run as ONE co-operative launch each -- k_epoch64 -- while only the dominant kernel is bracketed: one bracket
    #  per epoch; bracketing every phase keeps the three launches per step, which is what the phases are)
    assert e.update_mode() == 1 and pr["train_grad"][1] == E and pr["train_grad"][0] > 0.0
    assert all(pr[k][1] == 0 for k in pr if k != "train_grad")
    e.profile(True)
    e.collect_synthetic(p_term=0.1, time_limit=10)
    e.train(None)
    pr = e.profile_read()
    assert e.update_mode() == 0 and pr["train_grad"][1] == nmb and pr["apply"][1] == nmb and pr["grad_reduce"][1] == nmb and pr["gae"][1] == 1
    e.profile(False)
    e.train(None)
    assert all(v[1] == 0 for v in e.profile_read().values())
    e.close()


@pytest.mark.parametrize("H", [256, 64, 16])
@pytest.mark.parametrize("N,T,B", [(1, 2, 2), (1, 1, 1), (3, 5, 7), (33, 3, 64)])
def test_tiny_and_ragged_shapes_run_end_to_end(H, N, T, B):
    """Degenerate sizes (one env, one step, minibatch larger than / not dividing the rollout) through rollout,
    GAE and update on every kernel family, against the oracle."""
    D, A = 12, 18
    p = O.init_params(D, A, (H, H), (H, H), seed=1)
    e = make_engine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=B, n_epochs=2, pi=(H, H), vf=(H, H), seed=3)
    e.set_params(p)
    e.collect_synthetic(p_term=0.3, time_limit=2)
    e.synchronize()
    buf = {k: e.read(k) for k in ("actions", "rewards", "episode_starts", "values", "log_probs", "advantages", "returns")}
    buf["obs"] = e.read("obs")[:T]
    adv, ret = O.gae(buf["rewards"], buf["values"], buf["episode_starts"], e.read("last_values"), e.read("last_dones") > 0, 0.99, 0.95)
    assert np.array_equal(buf["advantages"], adv)
    rng = np.random.default_rng(0)
    perms = np.stack([rng.permutation(T * N) for _ in range(2)])
    st = e.train(perms)
    assert np.isfinite(st["loss"]) or T * N == 1   # a single sample has no advantage std: SB3 also yields nan there
    q = {k: v.copy() for k, v in p.items()}
    h = O.Hyper(n_epochs=2, batch_size=B)
    O.train(q, O.AdamState.zeros_like(q), buf, h, perms)
    got = e.get_params()
    if T * N > 1:
        for k in q:
            assert np.max(np.abs(got[k] - q[k])) < 2e-4, (k, float(np.max(np.abs(got[k] - q[k]))))
    e.close()


@pytest.mark.parametrize("H", [256, 64])
def test_graph_replay_equals_eager_enqueue(H):
    """The captured hipGraph (persistent rollout chunks + side-stream value pass + GAE) replays exactly what the
    eager enqueue does, rollout after rollout (device-resident counters advance inside the graph)."""
    D, A, N, T = 26, 2, 96, 40
    p = O.init_params(D, A, (H, H), (H, H), seed=3)
    outs = []
    for graph in (True, False):
        e = make_engine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=512, n_epochs=1, pi=(H, H), vf=(H, H), seed=5,
                        rollout_graph=graph)
        e.set_params(p)
        res = []
        for _ in range(3):
            e.collect_synthetic(p_term=0.05, time_limit=15)
            e.synchronize()
            res.append({k: e.read(k) for k in ("obs", "actions", "rewards", "values", "advantages", "last_values")})
        outs.append(res)
        e.close()
    for a, b in zip(*outs):
        for k in a:
            assert np.array_equal(a[k], b[k]), k
    assert not np.array_equal(outs[0][0]["obs"], outs[0][1]["obs"])  # consecutive rollouts differ (counters advanced)


def test_many_engine_lifecycles_in_one_process():
    """Create / roll out / update / destroy 60 engines of every kernel family in one process (graphs, side streams,
    events, arenas are all released; a leak or a stale handle shows up as a crash or an allocation failure here)."""
    rng = np.random.default_rng(0)
    for i in range(60):
        H = (256, 64, 32)[i % 3]
        D, A = ((58, 12), (14, 2), (26, 2), (12, 18))[i % 4]
        e = make_engine(obs_dim=D, act_dim=A, n_envs=int(rng.integers(1, 200)), n_steps=int(rng.integers(1, 12)) * 2,
                        batch_size=64, n_epochs=1, pi=(H, H), vf=(H, H), seed=i, rollout_persistent=bool(i % 2))
        e.set_params(O.init_params(D, A, (H, H), (H, H), seed=i))
        for _ in range(2):
            e.collect_synthetic(p_term=0.1, time_limit=5)
            st = e.train(None)
        assert np.isfinite(st["loss"])
        e.close()


@pytest.mark.parametrize("fast", [True, False])
def test_whole_iteration_matches_torch_golden(fast):
    """tests/golden/train_loop.npz: GAE + 2 epochs x 4 minibatches (last one short) computed with NumPy + torch from
    the real doggo checkpoint (weights and Adam state) -- the engine must reproduce advantages bit for bit and the
    losses / parameters / Adam moments within tolerance."""
    import os
    from collections import OrderedDict
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "train_loop.npz"))
    T, N, B, E, D, A = (int(x) for x in g["shape"])
    gamma, lam, clip, ent_coef, vf_coef, max_norm, lr, eps = (float(x) for x in g["hyper"])
    keys = O.param_keys()
    e = make_engine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=B, n_epochs=E, gamma=gamma, gae_lambda=lam,
                    clip_range=clip, ent_coef=ent_coef, vf_coef=vf_coef, max_grad_norm=max_norm, learning_rate=lr,
                    adam_eps=eps, fast_kernels=fast)
    e.set_params(OrderedDict((k, g[f"p0/{k}"]) for k in keys))
    e.set_optimizer_state(OrderedDict((k, g[f"m0/{k}"]) for k in keys), OrderedDict((k, g[f"v0/{k}"]) for k in keys),
                          int(g["adam_step"]))
    buf = {k: g[k] for k in ("obs", "actions", "rewards", "episode_starts", "values", "log_probs")}
    e.load_rollout(buf, g["last_values"], g["dones"])
    e.compute_gae()
    assert np.array_equal(e.read("advantages"), g["advantages"]) and np.array_equal(e.read("returns"), g["returns"])
    e.train(g["perms"])
    got = e.get_params()
    for k in keys:
        assert np.max(np.abs(got[k] - g[f"p1/{k}"])) < 1e-5, (k, float(np.max(np.abs(got[k] - g[f"p1/{k}"]))))
    m, v, step = e.get_optimizer_state()
    assert step == int(g["adam_step"]) + E * 4
    for k in keys:
        assert scaled_err(m[k], g[f"m1/{k}"]) < 1e-3, k
    e.close()


def test_error_paths():
    from mobrob_amd.engine import PPOEngine
    with pytest.raises(ValueError):
        PPOEngine(obs_dim=4, act_dim=2, n_envs=2, n_steps=2, pi=(30, 30))
    e = make_engine(obs_dim=4, act_dim=2, n_envs=2, n_steps=2, batch_size=2, n_epochs=1)
    with pytest.raises(Exception):
        e.epoch_begin(None)  # rollout not finished
    with pytest.raises(ValueError):
        e.set_flat_params(np.zeros(3, np.float32))
    # a caller's permutation is validated on the host before a kernel scatters through it
    buf, lv, dones = synthetic_rollout(2, 2, 4, 2, seed=1)
    e.load_rollout(buf, lv, dones)
    e.compute_gae()
    for bad, what in (([0, 1, 2, 4], "outside"), ([0, 1, -1, 2], "outside"), ([0, 1, 1, 2], "twice")):
        with pytest.raises(Exception, match=what):
            e.epoch_begin(np.array(bad, np.int64))
    e.epoch_begin(np.array([3, 1, 0, 2], np.int64))
    e.close()


# ------------------------------------------------------------------------------------------------
# fused fast path (hidden width 256): same oracle, plus a differential check against the generic kernels
# ------------------------------------------------------------------------------------------------
def _consistent_rollout(p, T, N, D, A, seed):
    rng = np.random.default_rng(seed)
    buf, lv, dones = synthetic_rollout(T, N, D, A, seed=seed)
    flat_obs = buf["obs"].reshape(T * N, D)
    mean, val = O.policy_outputs(p, flat_obs)
    acts = (mean + rng.standard_normal((T * N, A)).astype(np.float32) * np.exp(p["log_std"])).astype(np.float32)
    buf["actions"] = acts.reshape(T, N, A)
    buf["log_probs"] = (O.gaussian_log_prob(mean, p["log_std"], acts) + rng.normal(0, 0.1, T * N)).astype(np.float32).reshape(T, N)
    buf["values"] = (val + rng.normal(0, 0.1, T * N)).astype(np.float32).reshape(T, N)
    return buf, lv, dones


@pytest.mark.parametrize("D,A", [(58, 12), (14, 2), (26, 2), (12, 18), (43, 2)])
def test_fused_act_matches_oracle(D, A):
    N, H = 200, 256  # 200 rows: 3 full tiles + a partial one
    rng = np.random.default_rng(D)
    p = O.init_params(D, A, (H, H), (H, H), seed=D)
    p["log_std"] = rng.normal(0.2, 0.3, A).astype(np.float32)
    for k in p:
        if k.endswith("bias"):
            p[k] = rng.normal(0, 0.1, p[k].shape).astype(np.float32)
    obs = rng.standard_normal((N, D)).astype(np.float32) * 2
    eps = rng.standard_normal((N, A)).astype(np.float32)
    e = make_engine(obs_dim=D, act_dim=A, n_envs=N, n_steps=2, batch_size=64, n_epochs=1, pi=(H, H), vf=(H, H))
    e.set_params(p)
    a_raw, a_clip, val, lp = e.act(obs, eps)
    o_raw, o_clip, o_val, o_lp = O.act(p, obs, eps)
    assert scaled_err(a_raw, o_raw) < 1e-4 and scaled_err(val, o_val) < 1e-4
    assert np.allclose(lp, o_lp, rtol=1e-4, atol=1e-3)
    assert np.array_equal(a_clip, np.clip(a_raw, -1, 1))
    det, v2 = e.predict(obs, deterministic=True, want_values=True)
    assert np.allclose(det, np.clip(O.policy_outputs(p, obs)[0], -1, 1), atol=1e-4) and scaled_err(v2, o_val) < 1e-4
    e.close()


@pytest.mark.parametrize("cfg", [dict(D=58, A=12, T=8, N=13, B=104, E=1),        # one partial-tile minibatch (104 = 64+40)
                                 dict(D=58, A=12, T=50, N=200, B=10000, E=1),    # 157 tiles -> 128 WGs, some with 2 tiles
                                 dict(D=14, A=2, T=30, N=100, B=1000, E=2),      # 3 minibatches, DP=16
                                 dict(D=43, A=2, T=20, N=64, B=512, E=1),        # DP=48
                                 dict(D=26, A=2, T=9, N=7, B=63, E=2),           # DP=32, single tile, count < 64
                                 dict(D=12, A=18, T=12, N=40, B=200, E=1),       # head wider than 16 -> 32-wide head path
                                 dict(D=30, A=16, T=10, N=33, B=330, E=1)])      # head exactly 16 (16x16x4 path, no padding)
def test_fused_train_matches_oracle_and_generic(cfg):
    D, A, T, N, B, E = (cfg[k] for k in "DATNBE")
    H = 256
    rng = np.random.default_rng(17)
    p0 = O.init_params(D, A, (H, H), (H, H), seed=3)
    p0["log_std"] = rng.normal(-0.3, 0.2, A).astype(np.float32)
    p0["action_net.weight"] *= 30
    buf, lv, dones = _consistent_rollout(p0, T, N, D, A, seed=9)
    h = O.Hyper(gamma=0.99, gae_lambda=0.95, ent_coef=0.01, n_epochs=E, batch_size=B, learning_rate=3e-4)
    buf["advantages"], buf["returns"] = O.gae(buf["rewards"], buf["values"], buf["episode_starts"], lv, dones, h.gamma, h.gae_lambda)
    perms = np.stack([rng.permutation(T * N) for _ in range(E)])
    results = {}
    for fast in (True, False):
        e = make_engine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=B, n_epochs=E, pi=(H, H), vf=(H, H),
                        gamma=h.gamma, gae_lambda=h.gae_lambda, ent_coef=h.ent_coef, learning_rate=h.learning_rate,
                        fast_kernels=fast)
        e.set_params(p0)
        e.load_rollout(buf, lv, dones)
        # first-minibatch gradient
        e.epoch_begin(perms[0])
        e.minibatch_grad(0)
        g0 = e.read("grads")
        e.minibatch_apply()
        e.fetch_step_stats()
        # then the full update from scratch
        e.set_params(p0)
        z = {k: np.zeros_like(v) for k, v in p0.items()}
        e.set_optimizer_state(z, z, 0)
        stats = e.train(perms)
        results[fast] = (g0, e.get_params(), stats)
        e.close()
    p = {k: v.copy() for k, v in p0.items()}
    idx = perms[0][:B]
    _, og, _ = O.loss_and_grads(p, *O.gather_minibatch(buf, idx), h)
    og = O.flatten_params(og)
    scale = float(np.max(np.abs(og)))
    for fast in (True, False):
        assert np.max(np.abs(results[fast][0] - og)) < 1e-4 * max(1.0, scale), fast
    st = O.AdamState.zeros_like(p)
    ostats = O.train(p, st, buf, h, perms)
    nmb = -(-T * N // B)
    for fast in (True, False):
        newp, stats = results[fast][1], results[fast][2]
        for k in p:
            assert np.max(np.abs(newp[k] - p[k])) < 1e-4, (fast, k, float(np.max(np.abs(newp[k] - p[k]))))
        for k in ["policy_loss", "value_loss", "loss", "approx_kl", "clip_fraction", "grad_norm"]:
            ref = float(np.mean([float(s[k]) for s in ostats[-nmb:]]))
            assert abs(stats[k] - ref) < 2e-4 * max(1.0, abs(ref)), (fast, k, stats[k], ref)


def test_fused_gradients_are_run_to_run_deterministic():
    D, A, T, N, B, H = 58, 12, 40, 128, 5120, 256
    p = O.init_params(D, A, (H, H), (H, H), seed=1)
    buf, lv, dones = _consistent_rollout(p, T, N, D, A, seed=2)
    buf["advantages"], buf["returns"] = O.gae(buf["rewards"], buf["values"], buf["episode_starts"], lv, dones, 0.99, 0.95)
    e = make_engine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=B, n_epochs=1, pi=(H, H), vf=(H, H))
    e.set_params(p)
    e.load_rollout(buf, lv, dones)
    e.epoch_begin(np.arange(T * N))
    e.minibatch_grad(0)
    g1 = e.read("grads")
    e.minibatch_grad(0)
    g2 = e.read("grads")
    assert np.array_equal(g1, g2)  # slab reduction, no float atomics
    e.close()


_COUNTED_WAIT_SCRIPT = r"""
import hashlib, json, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from mobrob_amd.engine import PPOEngine
out = {}
for D, A in ((14, 2), (26, 2), (58, 12)):             # observation rows padded to 16 / 32 / 64 columns: the three instantiations
    H, N, T = 256, 256, 128                            # 32 768 rows = 512 tiles: four per workgroup (priming, running dW1 sums)
    e = PPOEngine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=N * T, n_epochs=1, pi=(H, H), vf=(H, H), seed=3)
    assert e.x3_mode() == 7, e.x3_mode()               # k_chain_train
    e.collect_synthetic()
    hs = []
    for rep in range(3):
        e.epoch_begin(None)
        e.minibatch_grad(0)
        hs.append(hashlib.sha256(e.read("grads").tobytes()).hexdigest())
        e.minibatch_apply()
    out[str(D)] = hs
    e.close()
print(json.dumps(out))
"""


def test_chain_counted_waits_equal_full_waits(tmp_path):
    """k_chain_train waits for its LDS-DMA ring and for the dW1 running sums with hand-counted `s_waitcnt vmcnt(N)` (the loads are
    inline asm the compiler does not track).  The validation build (-DMOBROB_CHAIN_VMCNT0: vmcnt(0) at every such wait, built by
    __graft_entry__.build() beside the product library) must produce the SAME gradient bits for all three observation widths,
    over three optimizer steps each (the static half of the check is tests/test_chain_isa.py)."""
    import subprocess
    import sys
    import json
    import __graft_entry__ as G
    script = tmp_path / "counted_waits.py"
    script.write_text(_COUNTED_WAIT_SCRIPT)
    res = {}
    for tag, lib in (("product", G.LIB), ("vmcnt0", G.LIB_VMCNT0)):
        assert os.path.exists(lib), lib
        env = dict(os.environ, MOBROB_PPO_LIB=lib)
        r = subprocess.run([sys.executable, str(script), G.ROOT], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (tag, r.stderr[-2000:])
        res[tag] = json.loads(r.stdout.strip().splitlines()[-1])
    assert res["product"] == res["vmcnt0"], res


_S8_SCRIPT = r"""
import hashlib, json, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from mobrob_amd.engine import PPOEngine
from mobrob_amd.envs.vec_env import DeviceGoalVecEnv
from oracle import ppo_oracle as O
out = {}
for name, D, A in (("point", 14, 2), ("car", 26, 2), ("turtlebot3", 43, 2), ("doggo", 58, 12)):   # padded rows of 16 / 32 / 48 / 64 columns
    H, N, T = 256, 200, 70                         # seven 32-row tiles, the last one ragged; 70 steps = two launches (chunks)
    for kind in ("synthetic", "goal"):
        e = PPOEngine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=N * T, n_epochs=1, pi=(H, H), vf=(H, H), seed=5)
        assert e.x3_mode() & 1
        p = O.init_params(D, A, (H, H), (H, H), seed=2)
        p["action_net.weight"] *= 20
        e.set_params(p)
        for it in range(2):                         # the second rollout continues the first one's episodes
            if kind == "synthetic":
                e.collect_synthetic(p_term=0.02, time_limit=25)
            else:
                DeviceGoalVecEnv.for_robot(name, N, time_limit=25).collect(e)
            e.synchronize()
            hs = hashlib.sha256()
            for k in ("obs", "actions", "log_probs", "rewards", "episode_starts", "values", "advantages", "returns", "last_values"):
                hs.update(e.read(k).tobytes())
            out[f"{name}/{kind}/{it}"] = hs.hexdigest()
        e.close()
print(json.dumps(out))
"""


def test_s8_rollout_equals_the_four_wave_x3_rollout(tmp_path):
    """k_rollout_persistent<.., S8> (eight GEMM waves, W2's leading pieces stationary in registers, h1 split once into bf16 planes)
    against the four-wave x3 form (MOBROB_ROLLOUT_S8=0): the same six products in the same order per accumulator, so EVERY stored
    array of a rollout -- observations, actions, log-probs, rewards, episode starts, values, advantages -- is the same bit for bit,
    for all four padded observation widths, both device env sources, ragged last tile, time-limit bootstraps, two rollouts in a row."""
    import subprocess
    import sys
    import json
    import __graft_entry__ as G
    script = tmp_path / "s8.py"
    script.write_text(_S8_SCRIPT)
    res = {}
    for s8 in ("1", "0"):
        r = subprocess.run([sys.executable, str(script), G.ROOT], env=dict(os.environ, MOBROB_ROLLOUT_S8=s8), capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, (s8, r.stderr[-2000:])
        res[s8] = json.loads(r.stdout.strip().splitlines()[-1])
    assert len(res["1"]) == 16 and res["1"] == res["0"], {k: (res["1"][k][:8], res["0"][k][:8]) for k in res["1"] if res["1"][k] != res["0"][k]}


def test_training_records_follow_every_write_to_the_arrays_they_pack():
    """k_fused_train reads actions / old log-prob / advantage / return / old value of a row from ONE packed record
    (kernels_fused.h, k_build_train_records) built once per rollout.  The arrays stay writable through the C ABI after
    epoch_begin: write_buffer must be seen by the next gradient launch, and so must a write through a device pointer
    obtained from buffer_info (the engine cannot see that one, it re-packs before every launch from then on)."""
    import torch
    from mobrob_amd.parallel import device_tensor
    D, A, T, N, B, H = 58, 12, 16, 64, 1024, 256
    p = O.init_params(D, A, (H, H), (H, H), seed=1)
    buf, lv, dones = _consistent_rollout(p, T, N, D, A, seed=2)
    buf["advantages"], buf["returns"] = O.gae(buf["rewards"], buf["values"], buf["episode_starts"], lv, dones, 0.99, 0.95)
    h = O.Hyper(gamma=0.99, gae_lambda=0.95, ent_coef=0.01, n_epochs=1, batch_size=B, learning_rate=3e-4)
    e = make_engine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=B, n_epochs=1, pi=(H, H), vf=(H, H),
                    gamma=h.gamma, gae_lambda=h.gae_lambda, ent_coef=h.ent_coef, learning_rate=h.learning_rate)
    e.set_params(p)
    e.load_rollout(buf, lv, dones)
    idx = np.arange(T * N)
    e.epoch_begin(idx)

    def check(tag):
        e.minibatch_grad(0)
        got = e.unflatten(e.read("grads"))
        _, og, _ = O.loss_and_grads(p, *O.gather_minibatch(buf, idx[:B]), h, acc=np.float64)
        errs = {k: scaled_err(got[k], og[k]) for k in og}
        assert max(errs.values()) < 1e-4, (tag, errs)
        return got

    g0 = check("as loaded")
    rng = np.random.default_rng(3)
    buf["log_probs"] = buf["log_probs"] + rng.normal(0, 0.1, buf["log_probs"].shape).astype(np.float32)
    e.write("log_probs", buf["log_probs"])                      # after epoch_begin
    g1 = check("write_buffer after epoch_begin")
    assert scaled_err(g1["mlp_extractor.policy_net.2.weight"], g0["mlp_extractor.policy_net.2.weight"]) > 1e-3
    ptr, nbytes = e.device_buffer("advantages")                  # a raw device pointer leaves the library
    adv = device_tensor(ptr, (T, N), torch.float32, torch.device("cuda:0"))
    buf["advantages"] = (buf["advantages"] + rng.normal(0, 0.5, buf["advantages"].shape)).astype(np.float32)  # not affine: survives normalisation
    adv.copy_(torch.from_numpy(buf["advantages"]))
    torch.cuda.synchronize()
    e.epoch_begin(idx)   # the normalisation statistics of a minibatch are taken here; the engine saw no write
    g2 = check("write through the device pointer")
    assert scaled_err(g2["mlp_extractor.policy_net.2.weight"], g1["mlp_extractor.policy_net.2.weight"]) > 1e-3
    ret = device_tensor(e.device_buffer("returns")[0], (T, N), torch.float32, torch.device("cuda:0"))
    buf["returns"] = (buf["returns"] + np.float32(0.25)).astype(np.float32)
    ret.copy_(torch.from_numpy(buf["returns"]))
    torch.cuda.synchronize()
    check("second write through a device pointer, no engine call in between")
    e.close()


# ------------------------------------------------------------------------------------------------
# one workgroup per tile for small minibatches (kernels_split64.h) against the one-wave-per-tile block kernel
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cfg", [dict(D=58, A=12, T=100, N=16, B=100, E=2),    # data/configs/doggo-ppo.yaml: four tiles
                                 dict(D=14, A=2, T=450, N=2, B=100, E=2),      # point-ppo.yaml (DP=16)
                                 dict(D=58, A=12, T=70, N=15, B=100, E=2),     # short last minibatch (50 rows)
                                 dict(D=43, A=2, T=40, N=16, B=128, E=2),      # turtlebot3 (DP=48)
                                 dict(D=26, A=2, T=64, N=4, B=64, E=2),        # car (DP=32), SB3's default batch
                                 dict(D=12, A=18, T=50, N=16, B=100, E=2),     # drone: 18 actions (head > 16)
                                 dict(D=58, A=12, T=64, N=64, B=1000, E=1),    # 32 tiles, the last one 8 rows; 11 block folds
                                 dict(D=14, A=2, T=128, N=32, B=2048, E=1),    # 64 tiles (the default limit), 4-way block sums
                                 dict(D=58, A=12, T=20, N=5, B=33, E=2)])      # one full tile + one row
def test_split_tile_kernel_is_bit_identical_to_the_block_kernel(cfg, monkeypatch):
    """k_split64_train (four waves share a tile; per-tile slabs folded in the block kernel's grouping) against
    k_fused64_train (one wave per tile): gradients of one minibatch, then parameters, both Adam moments and every logged
    statistic after whole train() calls must be the same BITS."""
    D, A, T, N, B, E = (cfg[k] for k in "DATNBE")
    H = 64
    rng = np.random.default_rng(11)
    p0 = O.init_params(D, A, (H, H), (H, H), seed=3)
    p0["log_std"] = rng.normal(-0.2, 0.2, A).astype(np.float32)
    p0["action_net.weight"] *= 20
    buf, lv, dones = _consistent_rollout(p0, T, N, D, A, seed=6)
    h = O.Hyper(gamma=0.99, gae_lambda=0.95, ent_coef=0.01, n_epochs=E, batch_size=B, learning_rate=3e-4)
    buf["advantages"], buf["returns"] = O.gae(buf["rewards"], buf["values"], buf["episode_starts"], lv, dones, h.gamma, h.gae_lambda)
    perms = np.stack([rng.permutation(T * N) for _ in range(E)])
    m0 = {k: rng.normal(0, 1e-3, v.shape).astype(np.float32) for k, v in p0.items()}
    v0 = {k: (1e-6 * (0.5 + rng.random(v.shape))).astype(np.float32) for k, v in p0.items()}
    out = {}
    for split in (True, False):
        if split:
            monkeypatch.delenv("MOBROB_SPLIT64_MAX_TILES", raising=False)
        else:
            monkeypatch.setenv("MOBROB_SPLIT64_MAX_TILES", "0")
        e = make_engine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=B, n_epochs=E, pi=(H, H), vf=(H, H),
                        ent_coef=h.ent_coef, learning_rate=h.learning_rate)
        e.set_params(p0)
        e.set_optimizer_state(m0, v0, 37)
        e.load_rollout(buf, lv, dones)
        e.epoch_begin(perms[0])
        e.minibatch_grad(0)
        g_first = e.read("grads")
        n_mb = e.n_minibatches
        e.minibatch_grad(n_mb - 1)          # the (possibly short) last minibatch
        g_last = e.read("grads")
        stats = e.train(perms)
        out[split] = (g_first, g_last, e.get_flat_params(), e.get_optimizer_state(), stats)
        e.close()
    (ga, gla, pa, (ma, va, sa), sta), (gb, glb, pb, (mb_, vb, sb), stb) = out[True], out[False]
    assert np.array_equal(ga, gb), float(np.max(np.abs(ga - gb)))
    assert np.array_equal(gla, glb), float(np.max(np.abs(gla - glb)))
    assert sa == sb
    assert np.array_equal(pa, pb)
    for k in ma:
        assert np.array_equal(ma[k], mb_[k]) and np.array_equal(va[k], vb[k]), k
    assert sta == stb, (sta, stb)
    p = {k: v.copy() for k, v in p0.items()}
    st = O.AdamState(type(p0)((k, v.copy()) for k, v in m0.items()), type(p0)((k, v.copy()) for k, v in v0.items()), 37)
    O.train(p, st, buf, h, perms)
    ref = O.flatten_params(p)
    assert np.max(np.abs(pa - ref)) < 1e-4, float(np.max(np.abs(pa - ref)))


@pytest.mark.parametrize("cfg", [dict(D=58, A=12, T=100, N=16, B=100, E=3),     # data/configs/doggo-ppo.yaml: four tiles per minibatch, 16 minibatches
                                 dict(D=14, A=2, T=200, N=2, B=96, E=2),        # point-ppo.yaml's envs; a short last minibatch (16 rows)
                                 dict(D=26, A=2, T=64, N=4, B=100, E=2),        # car: DP = 32; 256 rows = 100 + 100 + 56
                                 dict(D=12, A=18, T=50, N=16, B=100, E=2),      # drone: 18 actions (NJ = 10)
                                 dict(D=43, A=2, T=40, N=16, B=100, E=2),       # turtlebot3: DP = 48
                                 dict(D=58, A=12, T=128, N=32, B=2048, E=2)])   # 64 tiles per minibatch: 128 gradient workgroups
def test_epoch_kernel_is_bit_identical_to_the_three_launch_update(cfg):
    """k_epoch64 (csrc/kernels_epoch64.h: every epoch of PPO.train as ONE co-operative launch -- gradient -> grid barrier -> fixed-order slab
    reduction -> grid barrier -> clip + Adam + packs -> grid barrier, per minibatch) against the three launches per optimizer step it
    replaces (`epoch_kernel = 0`): the same device functions in the same order, so parameters, both Adam moments, the step counter and
    every logged statistic must be the same BITS after two train() calls; and both equal the oracle to 1e-4
    [SB3 PPO.train through /root/reference/src/mobrob/rl_control/ppo.py:73-74; shapes /root/reference/data/configs/*-ppo.yaml:12-23]."""
    D, A, T, N, B, E = (cfg[k] for k in "DATNBE")
    H = 64
    rng = np.random.default_rng(17)
    p0 = O.init_params(D, A, (H, H), (H, H), seed=5)
    p0["log_std"] = rng.normal(-0.2, 0.2, A).astype(np.float32)
    p0["action_net.weight"] *= 20
    buf, lv, dones = _consistent_rollout(p0, T, N, D, A, seed=8)
    h = O.Hyper(gamma=0.99, gae_lambda=0.95, ent_coef=0.01, n_epochs=E, batch_size=B, learning_rate=3e-4)
    buf["advantages"], buf["returns"] = O.gae(buf["rewards"], buf["values"], buf["episode_starts"], lv, dones, h.gamma, h.gae_lambda)
    perms = np.stack([[rng.permutation(T * N) for _ in range(E)] for _ in range(2)])
    m0 = {k: rng.normal(0, 1e-3, v.shape).astype(np.float32) for k, v in p0.items()}
    v0 = {k: (1e-6 * (0.5 + rng.random(v.shape))).astype(np.float32) for k, v in p0.items()}
    out = {}
    for epoch in (True, False):
        e = make_engine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=B, n_epochs=E, pi=(H, H), vf=(H, H),
                        ent_coef=h.ent_coef, learning_rate=h.learning_rate)
        if not epoch:
            e.set_hyper(epoch_kernel=0)
        e.set_params(p0)
        e.set_optimizer_state(m0, v0, 37)
        e.load_rollout(buf, lv, dones)
        stats = [e.train(perms[0]), e.train(perms[1])]
        assert e.update_mode() == (1 if epoch else 0)
        out[epoch] = (e.get_flat_params(), e.get_optimizer_state(), stats, e.read("grads"))
        e.close()
    (pa, (ma, va, sa), sta, ga), (pb, (mb_, vb, sb), stb, gb) = out[True], out[False]
    assert sa == sb == 37 + 2 * E * -(-(T * N) // B)
    assert np.array_equal(pa, pb), float(np.max(np.abs(pa - pb)))
    for k in ma:
        assert np.array_equal(ma[k], mb_[k]) and np.array_equal(va[k], vb[k]), k
    assert sta == stb, (sta, stb)
    assert np.array_equal(ga, gb)
    p = {k: v.copy() for k, v in p0.items()}
    st = O.AdamState(type(p0)((k, v.copy()) for k, v in m0.items()), type(p0)((k, v.copy()) for k, v in v0.items()), 37)
    O.train(p, st, buf, h, perms[0])
    O.train(p, st, buf, h, perms[1])
    ref = O.flatten_params(p)
    assert np.max(np.abs(pa - ref)) < 1e-4, float(np.max(np.abs(pa - ref)))


def test_epoch_kernel_gives_up_at_a_barrier_and_the_update_is_rerun_as_three_launches(monkeypatch, capfd):
    """Every wait of k_epoch64's grid barrier is bounded.  With MOBROB_EPOCH_TIMEOUT_S so small that no barrier can be passed in time the
    launch raises its abort word and every workgroup leaves -- and mobrob_ppo_train, which took a snapshot of the optimizer's state,
    restores it and runs the SAME update as three launches per step: the call succeeds, the result is the three-launch engine's bit
    for bit, the engine keeps that form afterwards.  The asynchronous form (train_enqueue) has no retry: there the abort fails the
    next synchronising call instead of passing a half-applied update on."""
    D, A, T, N, B, E, H = 14, 2, 64, 4, 64, 2, 64
    p0 = O.init_params(D, A, (H, H), (H, H), seed=2)
    zeros = {k: np.zeros_like(v) for k, v in p0.items()}
    out = {}
    for mode in ("abort", "three"):
        e = make_engine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=B, n_epochs=E, pi=(H, H), vf=(H, H), seed=9)
        e.set_params(p0)
        e.set_optimizer_state(zeros, zeros, 0)
        if mode == "three":
            e.set_hyper(epoch_kernel=0)
        else:
            monkeypatch.setenv("MOBROB_EPOCH_TIMEOUT_S", "1e-9")
        e.collect_synthetic()
        st1 = e.train(None)
        monkeypatch.delenv("MOBROB_EPOCH_TIMEOUT_S", raising=False)
        assert e.update_mode() == 0                       # (after the abort the engine keeps the three launches)
        e.collect_synthetic()
        st2 = e.train(None)
        m, v, step = e.get_optimizer_state()
        out[mode] = (e.get_flat_params(), m, v, step, st1, st2)
        if mode == "abort":
            assert "re-run as three launches" in capfd.readouterr().err
            # the asynchronous form: no snapshot, the abort surfaces at the next synchronising call
            e.set_hyper(epoch_kernel=1)
            monkeypatch.setenv("MOBROB_EPOCH_TIMEOUT_S", "1e-9")
            e.collect_synthetic()
            e.train_enqueue()
            with pytest.raises(Exception, match="gave up at a grid barrier"):
                e.synchronize()
            monkeypatch.delenv("MOBROB_EPOCH_TIMEOUT_S")
        e.close()
    (pa, ma, va, sa, a1, a2), (pb, mb, vb, sb, b1, b2) = out["abort"], out["three"]
    assert sa == sb and np.array_equal(pa, pb) and a1 == b1 and a2 == b2
    for k in ma:
        assert np.array_equal(ma[k], mb[k]) and np.array_equal(va[k], vb[k]), k


@pytest.mark.parametrize("kind", ["synthetic", "goal"])
@pytest.mark.parametrize("D,A,N,T", [(58, 12, 16, 60),     # data/configs/doggo-ppo.yaml: half a tile
                                     (14, 2, 2, 80),       # point-ppo.yaml: two envs
                                     (43, 2, 37, 24),      # DP = 48, a ragged second tile
                                     (12, 18, 100, 24),    # drone: 18 actions
                                     (26, 2, 1024, 16),    # car, 32 tiles (BASELINE config 2's env count)
                                     (14, 2, 2048, 128)])  # 64 tiles x 128 steps: cut into chunks, value pass on the side stream
def test_rollout64_tile_kernel_is_bit_identical_to_the_one_wave_kernel(kind, D, A, N, T, monkeypatch):
    """k_rollout64_tile (one workgroup per 32-env tile, forward split over two waves, weight fragments in registers)
    against k_rollout64_persistent (one wave per tile): every rollout buffer incl. the bootstrapped rewards, the
    truncation flags and the carried state over two consecutive rollouts must be the same BITS."""
    from mobrob_amd.envs.vec_env import DeviceGoalVecEnv
    H, TL = 64, 7
    p = O.init_params(D, A, (H, H), (H, H), seed=8)
    p["value_net.bias"] = np.array([2.0], np.float32)
    res = {}
    for tile in (True, False):
        if tile:
            monkeypatch.delenv("MOBROB_ROLLOUT64_TILE_MAX", raising=False)
        else:
            monkeypatch.setenv("MOBROB_ROLLOUT64_TILE_MAX", "0")
        e = make_engine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=256, n_epochs=1, pi=(H, H), vf=(H, H), seed=21)
        e.set_params(p)
        out = []
        for _ in range(2):
            if kind == "synthetic":
                e.collect_synthetic(p_term=0.05, time_limit=TL)
            else:
                DeviceGoalVecEnv(N, D, A, 2 if D < 12 or A != 18 else 3, time_limit=TL).collect(e)
            e.synchronize()
            out.append({k: e.read(k) for k in ("obs", "actions", "log_probs", "rewards", "episode_starts", "values", "last_values",
                                               "last_dones", "advantages", "returns", "truncated", "terminal_values", "terminal_obs")})
        res[tile] = out
        e.close()
    for a, b in zip(res[True], res[False]):
        for k in a:
            assert np.array_equal(a[k], b[k]), k
        assert a["episode_starts"][1:].sum() > 0 and a["truncated"].size  # resets and truncations were part of it


# ------------------------------------------------------------------------------------------------
# persistent two-wave workgroups for large minibatches (kernels_pair64.h) against the block kernel and the oracle
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cfg", [dict(D=14, A=2, T=64, N=128, B=4096, E=1),     # BASELINE config 2's robot: 128 tiles (DP=16)
                                 dict(D=58, A=12, T=64, N=128, B=8192, E=1),    # doggo 2x64 (DP=64): 256 tiles = one full minibatch
                                 dict(D=43, A=2, T=50, N=100, B=3000, E=1),     # turtlebot3 (DP=48): ragged last tile, short last minibatch
                                 dict(D=26, A=2, T=40, N=600, B=24000, E=1),    # car (DP=32): 750 tiles > 512 workgroups -> two tiles per pair
                                 dict(D=12, A=18, T=64, N=64, B=2100, E=2),     # drone: 18 actions (head > 16)
                                 dict(D=14, A=2, T=67, N=32, B=2144, E=1),      # 67 tiles: odd -> the last workgroup's second pair idles
                                 dict(D=28, A=4, T=80, N=32, B=2560, E=1),      # 4 actions: two action pairs per lane, constants in registers
                                 dict(D=30, A=7, T=80, N=32, B=2560, E=1)])     # odd action count: a padded action in the last pair
def test_pair_kernel_matches_the_block_kernel_and_the_oracle(cfg, monkeypatch):
    """k_pair64_train (two waves per tile, four workgroups per CU, accumulators over the workgroup's tiles) against
    k_fused64_train: the same per-tile arithmetic in another tile order -> gradients agree to rounding (1e-5 of each
    tensor's scale); a whole train() matches the oracle like the other paths."""
    D, A, T, N, B, E = (cfg[k] for k in "DATNBE")
    H = 64
    rng = np.random.default_rng(12)
    p0 = O.init_params(D, A, (H, H), (H, H), seed=4)
    p0["log_std"] = rng.normal(-0.2, 0.2, A).astype(np.float32)
    p0["action_net.weight"] *= 20
    buf, lv, dones = _consistent_rollout(p0, T, N, D, A, seed=7)
    h = O.Hyper(gamma=0.99, gae_lambda=0.95, ent_coef=0.01, n_epochs=E, batch_size=B, learning_rate=3e-4)
    buf["advantages"], buf["returns"] = O.gae(buf["rewards"], buf["values"], buf["episode_starts"], lv, dones, h.gamma, h.gae_lambda)
    perms = np.stack([rng.permutation(T * N) for _ in range(E)])
    m0 = {k: rng.normal(0, 1e-3, v.shape).astype(np.float32) for k, v in p0.items()}
    v0 = {k: (1e-6 * (0.5 + rng.random(v.shape))).astype(np.float32) for k, v in p0.items()}
    out = {}
    for pair in (True, False):
        if pair:
            monkeypatch.delenv("MOBROB_PAIR64_MIN_TILES", raising=False)
        else:
            monkeypatch.setenv("MOBROB_PAIR64_MIN_TILES", "0")
        e = make_engine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=B, n_epochs=E, pi=(H, H), vf=(H, H),
                        ent_coef=h.ent_coef, learning_rate=h.learning_rate)
        e.set_params(p0)
        e.set_optimizer_state(m0, v0, 37)
        e.load_rollout(buf, lv, dones)
        e.epoch_begin(perms[0])
        e.minibatch_grad(0)
        g_first = e.read("grads")
        e.minibatch_grad(e.n_minibatches - 1)
        g_last = e.read("grads")
        e.minibatch_grad(0)
        g_again = e.read("grads")
        stats = e.train(perms)
        out[pair] = (g_first, g_last, g_again, e.get_flat_params(), stats)
        e.close()
    (ga, gla, ga2, pa, sta), (gb, glb, _, pb, stb) = out[True], out[False]
    assert np.array_equal(ga, ga2)                                  # run-to-run deterministic
    offs = np.cumsum([0] + [v.size for v in p0.values()])
    for g1, g2 in ((ga, gb), (gla, glb)):
        for i, k in enumerate(p0):
            a_, b_ = g1[offs[i]:offs[i + 1]], g2[offs[i]:offs[i + 1]]
            assert np.max(np.abs(a_ - b_)) <= 1e-5 * max(np.max(np.abs(b_)), 1e-12), (k, float(np.max(np.abs(a_ - b_))), float(np.max(np.abs(b_))))
    assert np.max(np.abs(pa - pb)) < 2e-6, float(np.max(np.abs(pa - pb)))
    for k in ("policy_loss", "value_loss", "approx_kl", "clip_fraction"):
        assert abs(sta[k] - stb[k]) <= 1e-5 * max(1.0, abs(stb[k])), (k, sta[k], stb[k])
    p = {k: v.copy() for k, v in p0.items()}
    st = O.AdamState(type(p0)((k, v.copy()) for k, v in m0.items()), type(p0)((k, v.copy()) for k, v in v0.items()), 37)
    O.train(p, st, buf, h, perms)
    ref = O.flatten_params(p)
    assert np.max(np.abs(pa - ref)) < 1e-4, float(np.max(np.abs(pa - ref)))


@pytest.mark.parametrize("H", [64, 256])
@pytest.mark.parametrize("D,A", [(5, 3), (20, 4), (37, 2), (53, 7)])
def test_every_observation_width_up_to_64_runs_the_fused_kernels(D, A, H):
    """Observations are padded to the next of the 16 / 32 / 48 / 64 columns the fused kernels are built for (zeros), so widths
    like 5, 20, 37, 53 no longer fall back to the generic GEMM chain: act / rollout / train match the oracle, and the
    profile shows the fused gradient kernel doing the work."""
    T, N, B, E = 16, 24, 100, 2
    rng = np.random.default_rng(23)
    p0 = O.init_params(D, A, (H, H), (H, H), seed=6)
    p0["log_std"] = rng.normal(-0.3, 0.2, A).astype(np.float32)
    p0["action_net.weight"] *= 20
    buf, lv, dones = _consistent_rollout(p0, T, N, D, A, seed=10)
    h = O.Hyper(gamma=0.99, gae_lambda=0.95, ent_coef=0.01, n_epochs=E, batch_size=B, learning_rate=3e-4)
    buf["advantages"], buf["returns"] = O.gae(buf["rewards"], buf["values"], buf["episode_starts"], lv, dones, h.gamma, h.gae_lambda)
    perms = np.stack([rng.permutation(T * N) for _ in range(E)])
    e = make_engine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=B, n_epochs=E, pi=(H, H), vf=(H, H),
                    ent_coef=h.ent_coef, learning_rate=h.learning_rate)
    e.set_params(p0)
    obs = rng.standard_normal((N, D)).astype(np.float32)
    e.rollout_begin()
    raw, clipped, values, logp = e.act(obs)
    mean, val = O.policy_outputs(p0, obs)
    assert scaled_err(values, val) < 1e-4
    assert np.allclose(logp, O.gaussian_log_prob(mean, p0["log_std"], raw), rtol=1e-4, atol=1e-3)
    e.load_rollout(buf, lv, dones)
    e.profile(True)
    e.train(perms)
    pr = e.profile_read()
    assert pr["train_grad"][1] == E * (-(-T * N // B))          # the fused gradient kernel ran for every minibatch
    p = {k: v.copy() for k, v in p0.items()}
    O.train(p, O.AdamState.zeros_like(p), buf, h, perms)
    got = e.get_params()
    for k in p:
        assert np.max(np.abs(got[k] - p[k])) < 1e-4, (k, float(np.max(np.abs(got[k] - p[k]))))
    e.close()


@pytest.mark.parametrize("H", [64, 256])
@pytest.mark.parametrize("D,A,N,T,B,E", [(1, 1, 1, 3, 2, 2),        # one env, one feature, one action, two-row minibatches
                                         (64, 32, 5, 7, 16, 1),     # the widest observation / head the fused kernels take
                                         (65, 3, 4, 6, 8, 1),       # one feature more: generic GEMM chain
                                         (3, 1, 33, 2, 64, 2),      # minibatch of 64 rows, the second one 2 rows
                                         (58, 12, 3, 1, 3, 1)])     # a single step of three envs
def test_extreme_shapes_match_the_oracle_and_roll_out(D, A, N, T, B, E, H):
    """Smallest and widest shapes the engine accepts: a whole train() equals the oracle, and the device rollout
    (synthetic source, forced truncations) produces finite advantages twice in a row."""
    rng = np.random.default_rng(1)
    p0 = O.init_params(D, A, (H, H), (H, H), seed=2)
    buf, lv, dones = _consistent_rollout(p0, T, N, D, A, seed=3)
    h = O.Hyper(n_epochs=E, batch_size=B, ent_coef=0.01)
    buf["advantages"], buf["returns"] = O.gae(buf["rewards"], buf["values"], buf["episode_starts"], lv, dones, h.gamma, h.gae_lambda)
    perms = np.stack([rng.permutation(T * N) for _ in range(E)])
    e = make_engine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=B, n_epochs=E, pi=(H, H), vf=(H, H), ent_coef=0.01)
    e.set_params(p0)
    e.load_rollout(buf, lv, dones)
    e.train(perms)
    got = e.get_params()
    p = {k: v.copy() for k, v in p0.items()}
    O.train(p, O.AdamState.zeros_like(p), buf, h, perms)
    for k in p:
        assert np.max(np.abs(got[k] - p[k])) < 1e-5, (k, float(np.max(np.abs(got[k] - p[k]))))
    e.collect_synthetic(p_term=0.1, time_limit=5)
    e.collect_synthetic(p_term=0.1, time_limit=5)
    assert np.isfinite(e.read("advantages")).all() and np.isfinite(e.read("returns")).all()
    e.close()
