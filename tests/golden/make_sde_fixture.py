#!/usr/bin/env python3
"""Generate tests/golden/sde_cases.npz: PPO(use_sde=True) -- generalised state-dependent exploration -- through torch's own ops,
written after stable-baselines3 2.0.0's StateDependentNoiseDistribution (common/distributions.py) with its defaults (full_std=True,
use_expln=False, squash_output=False, learn_features=False, epsilon=1e-6), statement by statement:

    sample_weights        weights_dist = Normal(zeros_like(std), std); exploration_matrices = weights_dist.rsample((n_envs,))
    proba_distribution    variance = mm(latent_sde.detach() ** 2, std ** 2); Normal(mean_actions, sqrt(variance + epsilon))
    get_noise             bmm(latent_sde.unsqueeze(1), exploration_matrices).squeeze(1)
    log_prob / entropy    sum_independent_dims(distribution.log_prob(actions) / .entropy())

and PPO.train's loss block + clip_grad_norm_ + optim.Adam(eps=1e-5) on top (the reference splats `ppo_kwargs` into PPO verbatim:
/root/reference/src/mobrob/rl_control/ppo.py:58; its README points at SB3's PPO for "all supported parameters").
Needs torch (CPU) only; independent of oracle/ppo_oracle.py.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_arch_fixture import ACTS, Policy, D, A, N, B  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(1)
EPSILON = 1e-6
# (name, activation, pi, vf, full_std, use_expln, log_std_init)
CASES = [("sde_tanh_2x2", "tanh", (32, 24), (24, 16), True, False, -2.0), ("sde_relu_1_3", "relu", (24,), (32, 32, 16), True, False, -2.0),
         ("sde_elu_4", "elu", (16, 16, 24, 16), (16, 16, 16, 16), True, False, -2.0),
         # appended later (the cases above keep their seeds): the two non-default options of the distribution
         ("sde_expln", "tanh", (32, 24), (24, 16), True, True, 0.0), ("sde_shared_std", "relu", (24, 16), (32,), False, False, -1.5),
         ("sde_shared_expln", "tanh", (16,), (16, 16), False, True, 0.0)]


def get_std(log_std, full_std, use_expln, latent_sde_dim, action_dim):
    """StateDependentNoiseDistribution.get_std, statement by statement."""
    if use_expln:
        below_threshold = torch.exp(log_std) * (log_std <= 0)
        safe_log_std = log_std * (log_std > 0) + EPSILON
        above_threshold = (torch.log1p(safe_log_std) + 1.0) * (log_std > 0)
        std = below_threshold + above_threshold
    else:
        std = torch.exp(log_std)
    if full_std:
        return std
    return torch.ones(latent_sde_dim, action_dim) * std


def dist_of(net, obs):
    latent = net.mlp_extractor.policy_net(obs)
    mean = net.action_net(latent)
    std = get_std(net.log_std, net.full_std, net.use_expln, latent.shape[1], mean.shape[1])
    variance = torch.mm(latent.detach() ** 2, std ** 2)                # learn_features=False
    return torch.distributions.Normal(mean, torch.sqrt(variance + EPSILON)), mean, latent


def main():
    out = {}
    for ci, (name, act, pi, vf, full_std, use_expln, ls_init) in enumerate(CASES):
        torch.manual_seed(900 + ci)
        g = torch.Generator().manual_seed(1900 + ci)
        net = Policy(ACTS[act], pi, vf)
        HL = pi[-1]
        net.full_std, net.use_expln = full_std, use_expln
        net.log_std = torch.nn.Parameter(torch.ones(HL, A if full_std else 1) * ls_init)     # proba_distribution_net(log_std_init)
        with torch.no_grad():
            for k, p in net.named_parameters():
                if k.endswith(".bias"):
                    p.add_(0.1 * torch.randn(p.shape, generator=g))
            net.log_std.add_((0.3 if ci < 3 else 0.6) * torch.randn(HL, A, generator=g)[:, :net.log_std.shape[1]])
            net.action_net.weight.mul_(30.0)
        keys = [k for k, _ in net.named_parameters()]
        assert keys[0] == "log_std"
        o = {"activation": np.array(act), "pi": np.array(pi), "vf": np.array(vf), "full_std": np.array(full_std), "use_expln": np.array(use_expln)}
        for k, p in net.named_parameters():
            o[f"p/{k}"] = p.detach().numpy().copy()
        obs = 1.5 * torch.randn(N, D, generator=g)
        with torch.no_grad():
            std = get_std(net.log_std, full_std, use_expln, HL, A)
            weights_dist = torch.distributions.Normal(torch.zeros_like(std), std)
            z = torch.randn(N, HL, A, generator=g)
            theta = weights_dist.loc + z * weights_dist.scale          # rsample((N,)) with the draws kept
            dist, mean, latent = dist_of(net, obs)
            noise = torch.bmm(latent.unsqueeze(1), theta).squeeze(1)
            actions = mean + noise
            o["fwd/obs"], o["fwd/z"], o["fwd/mean"] = obs.numpy().copy(), z.numpy().copy(), mean.numpy().copy()
            o["fwd/value"] = net.value_net(net.mlp_extractor.value_net(obs)).flatten().numpy().copy()
            o["fwd/actions"], o["fwd/log_prob"] = actions.numpy().copy(), dist.log_prob(actions).sum(dim=1).numpy().copy()
            o["fwd/entropy"] = dist.entropy().sum(dim=1).numpy().copy()
            o["fwd/single"] = (mean + torch.mm(latent, theta[0])).numpy().copy()   # get_noise with exploration_mat (foreign batch)
        mb_obs = 1.5 * torch.randn(B, D, generator=g)
        with torch.no_grad():
            dist, mean, latent = dist_of(net, mb_obs)
            mb_act = mean + torch.randn(B, A, generator=g) * dist.scale
            lp = dist.log_prob(mb_act).sum(dim=1)
            mb_old_lp = lp + 0.15 * torch.randn(B, generator=g)
            v0 = net.value_net(net.mlp_extractor.value_net(mb_obs)).flatten()
            mb_old_v = v0 + 0.05 * torch.randn(B, generator=g)
            mb_adv = 0.5 + 2.0 * torch.randn(B, generator=g)
            mb_ret = v0 + torch.randn(B, generator=g)
        for k, t in [("obs", mb_obs), ("actions", mb_act), ("old_log_prob", mb_old_lp), ("old_values", mb_old_v),
                     ("advantages", mb_adv), ("returns", mb_ret)]:
            o[f"mb/{k}"] = t.numpy().copy()
        lr, clip, ent_coef, vf_coef, max_norm = 3e-4, 0.2, 0.01, 0.5, 0.5
        opt = torch.optim.Adam(net.parameters(), lr=lr, eps=1e-5)
        dist, _, _ = dist_of(net, mb_obs)
        log_prob = dist.log_prob(mb_act).sum(dim=1)
        entropy = dist.entropy().sum(dim=1)
        values = net.value_net(net.mlp_extractor.value_net(mb_obs)).flatten()
        advantages = (mb_adv - mb_adv.mean()) / (mb_adv.std() + 1e-8)
        ratio = torch.exp(log_prob - mb_old_lp)
        policy_loss = -torch.min(advantages * ratio, advantages * torch.clamp(ratio, 1 - clip, 1 + clip)).mean()
        clip_fraction = torch.mean((torch.abs(ratio - 1) > clip).float())
        value_loss = torch.nn.functional.mse_loss(mb_ret, values)
        entropy_loss = -torch.mean(entropy)
        loss = policy_loss + ent_coef * entropy_loss + vf_coef * value_loss
        with torch.no_grad():
            lr_ = log_prob - mb_old_lp
            approx_kl = torch.mean((torch.exp(lr_) - 1) - lr_)
        opt.zero_grad()
        loss.backward()
        named = dict(net.named_parameters())
        for k in keys:
            o[f"step/grad/{k}"] = named[k].grad.numpy().copy()
        total = torch.nn.utils.clip_grad_norm_(net.parameters(), max_norm)
        opt.step()
        for k, v in [("loss", loss), ("policy_loss", policy_loss), ("value_loss", value_loss), ("entropy_loss", entropy_loss),
                     ("approx_kl", approx_kl), ("clip_fraction", clip_fraction), ("grad_norm", total)]:
            o[f"step/{k}"] = np.float64(v.item())
        for k in keys:
            o[f"step/p/{k}"] = named[k].detach().numpy().copy()
        for k, v in o.items():
            out[f"{name}/{k}"] = v
        print(f"{name}: loss={loss.item():.6f} |g|={total.item():.5f} |g_log_std|={named['log_std'].grad.norm().item():.5f} "
              f"clipfrac={clip_fraction.item():.2f}")
    out["cases"] = np.array([c[0] for c in CASES])
    out["hyper"] = np.array([3e-4, 0.2, 0.01, 0.5, 0.5, 1e-5])
    path = f"{OUT}/sde_cases.npz"
    np.savez_compressed(path, **out)
    print(f"{path}: {len(out)} arrays, {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
