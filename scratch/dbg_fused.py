import sys, numpy as np
sys.path.insert(0, '.')
from oracle import ppo_oracle as O
from tests.util import synthetic_rollout
from tests.test_engine_gpu import _consistent_rollout, make_engine
for (D, A, T, N, B) in [(14, 2, 30, 100, 1000), (43, 2, 20, 64, 512), (26, 2, 9, 7, 63), (14, 12, 30, 100, 1000), (58, 2, 30, 100, 1000)]:
    H = 256
    rng = np.random.default_rng(17)
    p0 = O.init_params(D, A, (H, H), (H, H), seed=3)
    p0["log_std"] = rng.normal(-0.3, 0.2, A).astype(np.float32)
    p0["action_net.weight"] *= 30
    buf, lv, dones = _consistent_rollout(p0, T, N, D, A, seed=9)
    h = O.Hyper(ent_coef=0.01, n_epochs=1, batch_size=B)
    buf["advantages"], buf["returns"] = O.gae(buf["rewards"], buf["values"], buf["episode_starts"], lv, dones, h.gamma, h.gae_lambda)
    perm = rng.permutation(T * N)
    e = make_engine(obs_dim=D, act_dim=A, n_envs=N, n_steps=T, batch_size=B, n_epochs=1, pi=(H, H), vf=(H, H), ent_coef=0.01)
    e.set_params(p0); e.load_rollout(buf, lv, dones); e.epoch_begin(perm); e.minibatch_grad(0)
    g = e.unflatten(e.read("grads"))
    _, og, _ = O.loss_and_grads(p0, *O.gather_minibatch(buf, perm[:B]), h)
    print("D,A,B", D, A, B)
    for k in g:
        err = np.abs(g[k] - og[k])
        if err.max() > 1e-5:
            idx = np.argwhere(err > 1e-5)
            print("  ", k, g[k].shape, "max err", err.max(), "n bad", len(idx), "first bad", idx[:6].tolist())
    e.close()
