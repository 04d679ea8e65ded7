#!/usr/bin/env python3
"""Generate tests/golden/arch_cases.npz: `policy_kwargs` beyond the reference YAMLs' two tanh layers -- the other
`activation_fn` modules the engine accepts and `net_arch` depths up to eight -- driven through torch's OWN modules
(torch.nn.Linear + the activation module, torch.distributions.Normal, autograd, clip_grad_norm_, torch.optim.Adam with
SB3's eps = 1e-5), i.e. the third-party kernels stable-baselines3 2.0.0 would run for such a policy (the reference splats
`ppo_kwargs` into PPO verbatim: /root/reference/src/mobrob/rl_control/ppo.py:58).

Needs torch (CPU) only; independent of oracle/ppo_oracle.py so that it pins the oracle's activation / depth handling.
Per case `<name>/`: p/<key> initial parameters (orthogonal, SB3's gains), fwd/* (obs, eps, mean, value, actions, log_prob),
mb/* one minibatch of B = 100, step/* loss terms, every gradient (pre-clip), total norm, parameters and moments after one
Adam step from zero moments.
"""
import os

import numpy as np
import torch

OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(1)

ACTS = {"tanh": torch.nn.Tanh, "relu": torch.nn.ReLU, "elu": torch.nn.ELU, "leakyrelu": torch.nn.LeakyReLU,
        "sigmoid": torch.nn.Sigmoid, "softplus": torch.nn.Softplus, "softsign": torch.nn.Softsign,
        "hardtanh": torch.nn.Hardtanh, "relu6": torch.nn.ReLU6}

# (name, activation, pi widths, vf widths)
CASES = [(f"act_{a}", a, (32, 24), (24, 32, 16)) for a in ACTS] + [
    ("depth4_tanh", "tanh", (32, 24, 16, 24), (24, 16, 24, 16)),
    ("depth5_3_elu", "elu", (24, 24, 16, 16, 8), (32, 16, 8)),
    ("depth8_relu", "relu", (24, 16, 16, 16, 16, 16, 16, 8), (16,) * 8),
    ("depth1_6_tanh", "tanh", (40,), (16, 24, 16, 8, 16, 8)),
    # appended later (the cases above keep their seeds): activations whose derivative needs the pre-activation
    ("act_silu", "silu", (32, 24), (24, 32, 16)), ("act_gelu", "gelu", (32, 24), (24, 32, 16)), ("act_mish", "mish", (32, 24), (24, 32, 16)),
]
ACTS.update(silu=torch.nn.SiLU, gelu=torch.nn.GELU, mish=torch.nn.Mish)
D, A, N, B = 14, 3, 40, 100


class Policy(torch.nn.Module):
    """SB3 ActorCriticPolicy's module tree for a Box action space (parameter names and registration order)."""

    def __init__(self, act_cls, pi, vf):
        super().__init__()
        self.log_std = torch.nn.Parameter(torch.zeros(A))
        self.mlp_extractor = torch.nn.Module()

        def seq(widths):
            mods, prev = [], D
            for w in widths:
                mods += [torch.nn.Linear(prev, w), act_cls()]
                prev = w
            return torch.nn.Sequential(*mods)

        self.mlp_extractor.policy_net = seq(pi)
        self.mlp_extractor.value_net = seq(vf)
        self.action_net = torch.nn.Linear(pi[-1], A)
        self.value_net = torch.nn.Linear(vf[-1], 1)
        # ActorCriticPolicy._build, ortho_init: gain sqrt(2) for the extractor, 0.01 for the action head, 1 for the value head
        for mod, gain in ((self.mlp_extractor, np.sqrt(2)), (self.action_net, 0.01), (self.value_net, 1.0)):
            for m in mod.modules():
                if isinstance(m, torch.nn.Linear):
                    torch.nn.init.orthogonal_(m.weight, gain=gain)
                    m.bias.data.fill_(0.0)

    def dist(self, obs):
        mean = self.action_net(self.mlp_extractor.policy_net(obs))
        return torch.distributions.Normal(mean, torch.ones_like(mean) * self.log_std.exp()), mean

    def values(self, obs):
        return self.value_net(self.mlp_extractor.value_net(obs))


def main():
    out = {}
    for ci, (name, act, pi, vf) in enumerate(CASES):
        torch.manual_seed(100 + ci)
        g = torch.Generator().manual_seed(500 + ci)
        net = Policy(ACTS[act], pi, vf)
        with torch.no_grad():   # a trained-looking state: non-zero biases and log_std, a head that moves the mean
            for k, p in net.named_parameters():
                if k.endswith(".bias"):
                    p.add_(0.1 * torch.randn(p.shape, generator=g))
            net.log_std.add_(-0.3 + 0.2 * torch.randn(A, generator=g))
            net.action_net.weight.mul_(30.0)
        keys = [k for k, _ in net.named_parameters()]
        o = {"activation": np.array(act), "pi": np.array(pi), "vf": np.array(vf)}
        for k, p in net.named_parameters():
            o[f"p/{k}"] = p.detach().numpy().copy()
        scale = 6.0 if act == "relu6" else 1.5       # wide enough to reach hardtanh's / relu6's flat parts
        obs = scale * torch.randn(N, D, generator=g)
        with torch.no_grad():
            dist, mean = net.dist(obs)
            eps = torch.randn(N, A, generator=g)
            actions = mean + eps * dist.scale
            o["fwd/obs"], o["fwd/eps"], o["fwd/mean"] = obs.numpy().copy(), eps.numpy().copy(), mean.numpy().copy()
            o["fwd/value"] = net.values(obs).flatten().numpy().copy()
            o["fwd/actions"], o["fwd/log_prob"] = actions.numpy().copy(), dist.log_prob(actions).sum(dim=1).numpy().copy()
        mb_obs = scale * torch.randn(B, D, generator=g)
        with torch.no_grad():
            dist, mean = net.dist(mb_obs)
            mb_act = mean + torch.randn(B, A, generator=g) * dist.scale
            lp = dist.log_prob(mb_act).sum(dim=1)
            mb_old_lp = lp + 0.15 * torch.randn(B, generator=g)
            v0 = net.values(mb_obs).flatten()
            mb_old_v = v0 + 0.05 * torch.randn(B, generator=g)
            mb_adv = 0.5 + 2.0 * torch.randn(B, generator=g)
            mb_ret = v0 + torch.randn(B, generator=g)
        for k, t in [("obs", mb_obs), ("actions", mb_act), ("old_log_prob", mb_old_lp), ("old_values", mb_old_v),
                     ("advantages", mb_adv), ("returns", mb_ret)]:
            o[f"mb/{k}"] = t.numpy().copy()
        lr, clip, ent_coef, vf_coef, max_norm = 3e-4, 0.2, 0.01, 0.5, 0.5
        opt = torch.optim.Adam(net.parameters(), lr=lr, eps=1e-5)
        # PPO.train's loss block, statement by statement (stable_baselines3/ppo/ppo.py, 2.0.0)
        dist, _ = net.dist(mb_obs)
        log_prob = dist.log_prob(mb_act).sum(dim=1)
        entropy = dist.entropy().sum(dim=1)
        values = net.values(mb_obs).flatten()
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
        st = opt.state_dict()["state"]
        for i, k in enumerate(keys):
            o[f"step/p/{k}"] = named[k].detach().numpy().copy()
            o[f"step/m/{k}"] = st[i]["exp_avg"].numpy().copy()
            o[f"step/v/{k}"] = st[i]["exp_avg_sq"].numpy().copy()
        for k, v in o.items():
            out[f"{name}/{k}"] = v
        print(f"{name}: loss={loss.item():.6f} |g|={total.item():.5f} clipfrac={clip_fraction.item():.2f}")
    out["cases"] = np.array([c[0] for c in CASES])
    out["hyper"] = np.array([3e-4, 0.2, 0.01, 0.5, 0.5, 1e-5])   # lr, clip_range, ent_coef, vf_coef, max_grad_norm, adam eps
    path = f"{OUT}/arch_cases.npz"
    np.savez_compressed(path, **out)
    print(f"{path}: {len(out)} arrays, {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
