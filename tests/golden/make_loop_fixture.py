#!/usr/bin/env python3
"""Generate tests/golden/train_loop.npz: one whole SB3 iteration -- RolloutBuffer GAE, env-major flatten,
permuted minibatches with a short last one, E epochs of PPO.train -- computed with the third-party kernels
stable-baselines3 2.0.0 uses (NumPy for the rollout buffer, torch for policy / loss / autograd / clip_grad_norm_ /
Adam), starting from the REAL doggo checkpoint (weights + Adam state) of the reference.

Run in the build container only (needs /root/reference and torch CPU); written independently of oracle/ppo_oracle.py
so that it cross-checks the oracle's loop structure (minibatch order, per-minibatch normalisation, Adam step
counting), not just single steps.  The buffer code follows SB3's common/buffers.py statement by statement.
"""
import io
import json
import os
import sys
import zipfile

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_fixtures import REF, TorchMirror  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(1)


def main():
    z = zipfile.ZipFile(f"{REF}/doggo-ppo.zip")
    d = json.loads(z.read("data"))
    sd = torch.load(io.BytesIO(z.read("policy.pth")), map_location="cpu", weights_only=True)
    opt_sd = torch.load(io.BytesIO(z.read("policy.optimizer.pth")), map_location="cpu", weights_only=True)
    net = TorchMirror(sd)
    pg = opt_sd["param_groups"][0]
    opt = torch.optim.Adam(net.parameters(), lr=pg["lr"], eps=pg["eps"], betas=tuple(pg["betas"]))
    opt.load_state_dict(opt_sd)
    keys = list(sd.keys())
    # snapshot the initial state NOW: load_state_dict may alias the checkpoint tensors, which opt.step() then mutates
    init = {}
    for i, k in enumerate(keys):
        init[f"p0/{k}"] = sd[k].numpy().copy()
        init[f"m0/{k}"] = opt_sd["state"][i]["exp_avg"].numpy().copy()
        init[f"v0/{k}"] = opt_sd["state"][i]["exp_avg_sq"].numpy().copy()
    adam_step0 = int(opt_sd["state"][0]["step"].item())
    D, A = sd["mlp_extractor.policy_net.0.weight"].shape[1], sd["log_std"].shape[0]
    T, N, B, E = 12, 6, 20, 2
    gamma, lam = float(d["gamma"]), float(d["gae_lambda"])
    clip, ent_coef, vf_coef, max_norm = 0.2, 0.01, float(d["vf_coef"]), float(d["max_grad_norm"])
    g = torch.Generator().manual_seed(77)
    rs = np.random.RandomState(5)

    # ---- rollout buffer filled the way collect_rollouts fills it (synthetic env stream) ----
    obs = (0.5 * torch.randn(T, N, D, generator=g)).numpy()
    rewards = rs.normal(0.03, 0.3, (T, N)).astype(np.float32)
    dones_seq = rs.rand(T, N) < 0.15                      # dones returned by env.step at each t
    episode_starts = np.zeros((T, N), np.float32)
    episode_starts[0] = 1.0
    episode_starts[1:] = dones_seq[:-1]                   # buffer stores the PREVIOUS step's dones
    actions = np.zeros((T, N, A), np.float32)
    values = np.zeros((T, N), np.float32)
    log_probs = np.zeros((T, N), np.float32)
    with torch.no_grad():
        for t in range(T):
            o = torch.from_numpy(obs[t])
            dist, mean = net.dist(o)
            a = mean + torch.randn(N, A, generator=g) * dist.scale
            actions[t] = a.numpy()
            values[t] = net.values(o).flatten().numpy()
            log_probs[t] = dist.log_prob(a).sum(dim=1).numpy()
        last_obs = 0.5 * torch.randn(N, D, generator=g)
        last_values = net.values(last_obs).flatten().numpy()
    dones = dones_seq[-1]

    # ---- RolloutBuffer.compute_returns_and_advantage (SB3 common/buffers.py) ----
    advantages = np.zeros((T, N), np.float32)
    last_gae_lam = 0
    for step in reversed(range(T)):
        if step == T - 1:
            next_non_terminal = 1.0 - dones            # bool array -> float64
            next_values = last_values
        else:
            next_non_terminal = 1.0 - episode_starts[step + 1]
            next_values = values[step + 1]
        delta = rewards[step] + gamma * next_values * next_non_terminal - values[step]
        last_gae_lam = delta + gamma * lam * next_non_terminal * last_gae_lam
        advantages[step] = last_gae_lam
    returns = advantages + values

    # ---- RolloutBuffer.get: swap_and_flatten (env-major) + permutation ----
    def flat(x):
        shape = x.shape
        if len(shape) < 3:
            shape = (*shape, 1)
        return x.swapaxes(0, 1).reshape(shape[0] * shape[1], *shape[2:])

    f_obs, f_act = flat(obs), flat(actions)
    f_val, f_lp, f_adv, f_ret = flat(values).flatten(), flat(log_probs).flatten(), flat(advantages).flatten(), flat(returns).flatten()
    perms = np.stack([rs.permutation(T * N) for _ in range(E)])

    stats = []
    named = dict(net.named_parameters())
    for ep in range(E):
        idx = perms[ep]
        start = 0
        while start < T * N:
            mb = idx[start:start + B]
            start += B
            o = torch.from_numpy(f_obs[mb]); a = torch.from_numpy(f_act[mb])
            old_lp = torch.from_numpy(f_lp[mb]); adv = torch.from_numpy(f_adv[mb]); ret = torch.from_numpy(f_ret[mb])
            dist, _ = net.dist(o)
            log_prob = dist.log_prob(a).sum(dim=1)
            entropy = dist.entropy().sum(dim=1)
            v = net.values(o).flatten()
            if len(adv) > 1:
                adv = (adv - adv.mean()) / (adv.std() + 1e-8)
            ratio = torch.exp(log_prob - old_lp)
            pl = -torch.min(adv * ratio, adv * torch.clamp(ratio, 1 - clip, 1 + clip)).mean()
            vl = torch.nn.functional.mse_loss(ret, v)
            el = -torch.mean(entropy)
            loss = pl + ent_coef * el + vf_coef * vl
            with torch.no_grad():
                lr_ = log_prob - old_lp
                kl = torch.mean((torch.exp(lr_) - 1) - lr_).item()
                cf = torch.mean((torch.abs(ratio - 1) > clip).float()).item()
            opt.zero_grad()
            loss.backward()
            gn = torch.nn.utils.clip_grad_norm_(net.parameters(), max_norm).item()
            opt.step()
            stats.append([pl.item(), vl.item(), el.item(), loss.item(), kl, cf, gn, len(mb)])

    out = dict(obs=obs, actions=actions, rewards=rewards, episode_starts=episode_starts, values=values, log_probs=log_probs,
               last_values=last_values, dones=dones, advantages=advantages, returns=returns, perms=perms,
               stats=np.array(stats, np.float64), hyper=np.array([gamma, lam, clip, ent_coef, vf_coef, max_norm, pg["lr"], pg["eps"]], np.float64),
               shape=np.array([T, N, B, E, D, A], np.int64), adam_step=np.int64(adam_step0))
    out.update(init)
    new_opt = opt.state_dict()
    for i, k in enumerate(keys):
        out[f"p1/{k}"] = named[k].detach().numpy().copy()
        out[f"m1/{k}"] = new_opt["state"][i]["exp_avg"].numpy().copy()
    np.savez_compressed(f"{OUT}/train_loop.npz", **out)
    print("train_loop.npz:", os.path.getsize(f"{OUT}/train_loop.npz") // 1024, "KiB;", len(stats), "optimizer steps; last loss", stats[-1][3])


if __name__ == "__main__":
    main()
