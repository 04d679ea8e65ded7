"""Torch-CPU restatement of the reference's PPO iteration (TEST / MEASUREMENT INFRASTRUCTURE ONLY: bench.py's cpu_baseline leg).

The reference's arithmetic runs inside stable-baselines3 2.0.0 on torch-CPU (`device: cpu` in every YAML,
/root/reference/data/configs/doggo-ppo.yaml:24; `torch.set_num_threads(1)` at /root/reference/examples/train.py:13).  SB3 is
not installable here, so this module strings the SAME torch kernels together in SB3's order -- `nn.Linear` / `nn.Tanh` MLPs
(`ActorCriticPolicy` with separate pi / vf extractors), `distributions.Normal`, autograd, `clip_grad_norm_`,
`optim.Adam(eps=1e-5)` -- around the NumPy pieces SB3 itself runs in NumPy (RolloutBuffer GAE, env-major flatten, permutation):
the honest "what a CPU gets out of this algorithm with torch" column beside the NumPy oracle's.  It follows
oracle/ppo_oracle.py statement for statement (same synthetic env source, same GAE function) and is checked against it in
tests/test_oracle.py::test_torch_baseline_matches_the_oracle.  Never imported by the product."""
from __future__ import annotations

import numpy as np

from . import ppo_oracle as O


def _torch():
    import torch
    return torch


class TorchPolicy:
    """SB3 `ActorCriticPolicy` (MlpPolicy, net_arch=dict(pi=[H, H], vf=[H, H]), tanh) as plain torch modules."""

    def __init__(self, params):
        torch = _torch()
        nn = torch.nn
        p = params

        def mlp(prefix):
            mods, i = [], 0
            while f"mlp_extractor.{prefix}.{2 * i}.weight" in p:
                w = p[f"mlp_extractor.{prefix}.{2 * i}.weight"]
                lin = nn.Linear(w.shape[1], w.shape[0])
                with torch.no_grad():
                    lin.weight.copy_(torch.from_numpy(w))
                    lin.bias.copy_(torch.from_numpy(p[f"mlp_extractor.{prefix}.{2 * i}.bias"]))
                mods += [lin, nn.Tanh()]
                i += 1
            return nn.Sequential(*mods)

        def head(wk, bk):
            w = p[wk]
            lin = nn.Linear(w.shape[1], w.shape[0])
            with torch.no_grad():
                lin.weight.copy_(torch.from_numpy(w))
                lin.bias.copy_(torch.from_numpy(p[bk]))
            return lin

        self.log_std = nn.Parameter(torch.from_numpy(np.array(p["log_std"], np.float32)))
        self.pi, self.vf = mlp("policy_net"), mlp("value_net")
        self.action_net, self.value_net = head("action_net.weight", "action_net.bias"), head("value_net.weight", "value_net.bias")
        # SB3 registration order: log_std, pi.*, vf.*, action_net, value_net
        self.params = [self.log_std] + list(self.pi.parameters()) + list(self.vf.parameters()) + \
            list(self.action_net.parameters()) + list(self.value_net.parameters())

    def dist_and_value(self, obs):
        torch = _torch()
        mean = self.action_net(self.pi(obs))
        value = self.value_net(self.vf(obs))[:, 0]
        return torch.distributions.Normal(mean, torch.ones_like(mean) * self.log_std.exp()), value

    def state(self):
        keys = O.param_keys()
        return {k: t.detach().numpy().copy() for k, t in zip(keys, self.params)}


def collect_rollout(policy, env, last_obs, last_starts, T, h, eps_source):
    """OnPolicyAlgorithm.collect_rollouts with the torch policy (no_grad forward, NumPy buffer) -- mirrors O.collect_rollout."""
    torch = _torch()
    N, D = last_obs.shape
    A = policy.log_std.shape[0]
    f = np.float32
    buf = dict(obs=np.zeros((T, N, D), f), actions=np.zeros((T, N, A), f), rewards=np.zeros((T, N), f),
               episode_starts=np.zeros((T, N), f), values=np.zeros((T, N), f), log_probs=np.zeros((T, N), f))
    dones = np.zeros(N, bool)
    for t in range(T):
        with torch.no_grad():
            dist, value = policy.dist_and_value(torch.from_numpy(last_obs))
            actions = dist.mean + torch.from_numpy(np.asarray(eps_source(t), f)) * dist.stddev      # Normal.rsample with supplied noise
            logp = dist.log_prob(actions).sum(dim=1)
        a = actions.numpy()
        new_obs, rewards, dones, trunc, terminal_obs = env.step(np.clip(a, -1.0, 1.0))
        rewards = rewards.astype(f).copy()
        if trunc.any():
            with torch.no_grad():
                tv = policy.dist_and_value(torch.from_numpy(terminal_obs[trunc]))[1].numpy()
            rewards[trunc] = np.array([O.bootstrap_reward(r, h.gamma, v) for r, v in zip(rewards[trunc], tv)], f)
        buf["obs"][t], buf["actions"][t], buf["rewards"][t] = last_obs, a, rewards
        buf["episode_starts"][t], buf["values"][t], buf["log_probs"][t] = last_starts.astype(f), value.numpy(), logp.numpy()
        last_obs, last_starts = new_obs, dones
    with torch.no_grad():
        last_values = policy.dist_and_value(torch.from_numpy(last_obs))[1].numpy()
    buf["advantages"], buf["returns"] = O.gae(buf["rewards"], buf["values"], buf["episode_starts"], last_values, dones, h.gamma, h.gae_lambda)
    return buf, last_obs, last_starts


def train(policy, opt, buf, h, perms):
    """PPO.train: autograd + clip_grad_norm_ + Adam, minibatches from the env-major flatten (mirrors O.train)."""
    torch = _torch()
    T, N = buf["rewards"].shape
    total = T * N
    stats = []
    for e in range(h.n_epochs):
        perm = np.asarray(perms[e])
        for s in range(0, total, h.batch_size):
            obs, act, old_v, old_lp, adv, ret = (torch.from_numpy(np.ascontiguousarray(x)) for x in O.gather_minibatch(buf, perm[s:s + h.batch_size]))
            dist, values = policy.dist_and_value(obs)
            log_prob = dist.log_prob(act).sum(dim=1)
            entropy = dist.entropy().sum(dim=1)
            if h.normalize_advantage and len(adv) > 1:
                adv = (adv - adv.mean()) / (adv.std() + 1e-8)
            ratio = torch.exp(log_prob - old_lp)
            pl = -torch.min(adv * ratio, adv * torch.clamp(ratio, 1 - h.clip_range, 1 + h.clip_range)).mean()
            vl = torch.nn.functional.mse_loss(ret, values)
            el = -entropy.mean()
            loss = pl + h.ent_coef * el + h.vf_coef * vl
            opt.zero_grad()
            loss.backward()
            gn = torch.nn.utils.clip_grad_norm_(policy.params, h.max_grad_norm)
            opt.step()
            stats.append(dict(loss=float(loss.detach()), policy_loss=float(pl.detach()), value_loss=float(vl.detach()), grad_norm=float(gn)))
    return stats


def make_optimizer(policy, h):
    return _torch().optim.Adam(policy.params, lr=h.learning_rate, betas=(h.beta1, h.beta2), eps=h.adam_eps)
