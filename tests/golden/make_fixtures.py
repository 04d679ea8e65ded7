#!/usr/bin/env python3
"""Generate tests/golden/<env>.npz from the reference checkpoints (run in the BUILD container only).

Needs /root/reference (read-only) and torch (CPU).  Nothing from /root/reference travels to the
GPU box except the arrays this script derives.  Recipe = SURVEY.md Appendix B.

What is extracted (real reference artefacts, /root/reference/data/policies/<env>-ppo.zip):
  p/<key>      the 13 trained weight tensors (policy.pth)
  m/<key>, v/<key>, adam_step     Adam exp_avg / exp_avg_sq / step (policy.optimizer.pth)
  last_obs     real simulator observations `_last_obs` (cast to float32 as SB3's obs.float() does)
  hyper/*      the hyper-parameters the run actually used (`data` JSON)
  ep_r, ep_l, ep_t   last-100-episode Monitor stats

What is computed here with the SAME third-party kernels stable-baselines3 2.0.0 calls
(torch.nn.Linear/Tanh, torch.distributions.Normal, autograd, clip_grad_norm_, torch.optim.Adam
loaded with the checkpoint's real optimizer state) -- the golden outputs the oracle and the HIP
path are both checked against:
  fwd/*        eps, mean, value, actions, clipped, log_prob, entropy on last_obs
  mb/*         a seeded synthetic minibatch (B=100 as in data/configs/*.yaml) built around last_obs
  step/*       loss terms, 13 gradients (pre-clip), total grad norm, post-Adam params and moments
"""
import base64
import io
import json
import os
import pickle
import sys
import warnings
import zipfile

import numpy as np
import torch

warnings.filterwarnings("ignore")
REF = "/root/reference/data/policies"
OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(1)


def unpickle(d, key):
    return pickle.loads(base64.b64decode(d[key][":serialized:"]))


class TorchMirror(torch.nn.Module):
    """Module tree with SB3's parameter names so the checkpoint state_dict loads verbatim."""

    def __init__(self, sd):
        super().__init__()
        A = sd["log_std"].shape[0]
        self.log_std = torch.nn.Parameter(torch.zeros(A))
        self.mlp_extractor = torch.nn.Module()

        def seq(prefix):
            mods, i = [], 0
            while f"mlp_extractor.{prefix}.{2 * i}.weight" in sd:
                o, n = sd[f"mlp_extractor.{prefix}.{2 * i}.weight"].shape
                mods += [torch.nn.Linear(n, o), torch.nn.Tanh()]
                i += 1
            return torch.nn.Sequential(*mods)

        self.mlp_extractor.policy_net = seq("policy_net")
        self.mlp_extractor.value_net = seq("value_net")
        self.action_net = torch.nn.Linear(sd["action_net.weight"].shape[1], A)
        self.value_net = torch.nn.Linear(sd["value_net.weight"].shape[1], 1)
        self.load_state_dict(sd)

    def dist(self, obs):
        latent_pi = self.mlp_extractor.policy_net(obs)
        mean = self.action_net(latent_pi)
        std = torch.ones_like(mean) * self.log_std.exp()
        return torch.distributions.Normal(mean, std), mean

    def values(self, obs):
        return self.value_net(self.mlp_extractor.value_net(obs))


def main():
    for env in ["point", "car", "doggo", "drone", "turtlebot3"]:
        z = zipfile.ZipFile(f"{REF}/{env}-ppo.zip")
        d = json.loads(z.read("data"))
        sd = torch.load(io.BytesIO(z.read("policy.pth")), map_location="cpu", weights_only=True)
        opt_sd = torch.load(io.BytesIO(z.read("policy.optimizer.pth")), map_location="cpu", weights_only=True)
        last_obs_raw = unpickle(d, "_last_obs")
        last_obs = np.asarray(last_obs_raw).astype(np.float32)
        ep = list(unpickle(d, "ep_info_buffer"))
        out = {}
        keys = list(sd.keys())
        for i, k in enumerate(keys):
            out[f"p/{k}"] = sd[k].numpy().copy()
            out[f"m/{k}"] = opt_sd["state"][i]["exp_avg"].numpy().copy()
            out[f"v/{k}"] = opt_sd["state"][i]["exp_avg_sq"].numpy().copy()
        out["adam_step"] = np.int64(int(opt_sd["state"][0]["step"].item()))
        out["last_obs"] = last_obs
        out["last_obs_dtype"] = np.array(str(np.asarray(last_obs_raw).dtype))
        for k in ["gamma", "gae_lambda", "ent_coef", "vf_coef", "max_grad_norm", "learning_rate"]:
            out[f"hyper/{k}"] = np.float64(d[k])
        for k in ["n_envs", "n_steps", "batch_size", "n_epochs", "num_timesteps", "_n_updates"]:
            out[f"hyper/{k}"] = np.int64(d[k])
        pg = opt_sd["param_groups"][0]
        out["hyper/adam_eps"] = np.float64(pg["eps"])
        out["hyper/beta1"], out["hyper/beta2"] = np.float64(pg["betas"][0]), np.float64(pg["betas"][1])
        out["hyper/clip_range"] = np.float64(0.2)  # constant schedule (blob not loadable)
        out["ep_r"] = np.array([e["r"] for e in ep], np.float64)
        out["ep_l"] = np.array([e["l"] for e in ep], np.int64)
        out["ep_t"] = np.array([e["t"] for e in ep], np.float64)

        net = TorchMirror(sd)
        g = torch.Generator().manual_seed(1234)
        obs_t = torch.from_numpy(last_obs)
        N, A = last_obs.shape[0], sd["log_std"].shape[0]
        with torch.no_grad():
            dist, mean = net.dist(obs_t)
            eps = torch.randn(N, A, generator=g)
            actions = mean + eps * dist.scale
            logp = dist.log_prob(actions).sum(dim=1)
            ent = dist.entropy().sum(dim=1)
            val = net.values(obs_t).flatten()
        out["fwd/eps"], out["fwd/mean"], out["fwd/value"] = eps.numpy(), mean.numpy(), val.numpy()
        out["fwd/actions"], out["fwd/clipped"] = actions.numpy(), np.clip(actions.numpy(), -1.0, 1.0)
        out["fwd/log_prob"], out["fwd/entropy"] = logp.numpy(), ent.numpy()

        # ---- one PPO.train minibatch step, B = 100 ----
        B = 100
        rows = torch.randint(0, N, (B,), generator=g)
        mb_obs = obs_t[rows] + 0.1 * torch.randn(B, last_obs.shape[1], generator=g)
        with torch.no_grad():
            dist, mean = net.dist(mb_obs)
            mb_act = mean + torch.randn(B, A, generator=g) * dist.scale
            lp = dist.log_prob(mb_act).sum(dim=1)
            mb_old_lp = lp + 0.15 * torch.randn(B, generator=g)  # ratios spread around 1, some clipped
            v0 = net.values(mb_obs).flatten()
            mb_old_v = v0 + 0.05 * torch.randn(B, generator=g)
            mb_adv = 0.5 + 2.0 * torch.randn(B, generator=g)
            mb_ret = v0 + torch.randn(B, generator=g)
        for k, t in [("obs", mb_obs), ("actions", mb_act), ("old_log_prob", mb_old_lp), ("old_values", mb_old_v),
                     ("advantages", mb_adv), ("returns", mb_ret)]:
            out[f"mb/{k}"] = t.numpy().copy()

        opt = torch.optim.Adam(net.parameters(), lr=pg["lr"], eps=pg["eps"], betas=tuple(pg["betas"]))
        opt.load_state_dict(opt_sd)
        clip, ent_coef, vf_coef = 0.2, float(d["ent_coef"]), float(d["vf_coef"])
        # literal restatement of PPO.train's loss block with torch ops
        dist, _ = net.dist(mb_obs)
        log_prob = dist.log_prob(mb_act).sum(dim=1)
        entropy = dist.entropy().sum(dim=1)
        values = net.values(mb_obs).flatten()
        advantages = (mb_adv - mb_adv.mean()) / (mb_adv.std() + 1e-8)
        ratio = torch.exp(log_prob - mb_old_lp)
        pl1 = advantages * ratio
        pl2 = advantages * torch.clamp(ratio, 1 - clip, 1 + clip)
        policy_loss = -torch.min(pl1, pl2).mean()
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
            out[f"step/grad/{k}"] = named[k].grad.numpy().copy()
        total = torch.nn.utils.clip_grad_norm_(net.parameters(), float(d["max_grad_norm"]))
        opt.step()
        out["step/loss"], out["step/policy_loss"] = loss.item(), policy_loss.item()
        out["step/value_loss"], out["step/entropy_loss"] = value_loss.item(), entropy_loss.item()
        out["step/approx_kl"], out["step/clip_fraction"] = approx_kl.item(), clip_fraction.item()
        out["step/grad_norm"] = total.item()
        out["step/adv_norm"] = advantages.numpy().copy()
        out["step/ratio"] = ratio.detach().numpy().copy()
        new_opt = opt.state_dict()
        for i, k in enumerate(keys):
            out[f"step/p/{k}"] = named[k].detach().numpy().copy()
            out[f"step/m/{k}"] = new_opt["state"][i]["exp_avg"].numpy().copy()
            out[f"step/v/{k}"] = new_opt["state"][i]["exp_avg_sq"].numpy().copy()
        np.savez_compressed(f"{OUT}/{env}.npz", **out)
        sz = os.path.getsize(f"{OUT}/{env}.npz")
        print(f"{env}: {len(out)} arrays, {sz / 1024:.0f} KiB, loss={loss.item():.6f} |g|={total.item():.5f} "
              f"clipfrac={clip_fraction.item():.2f}")


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("reference checkpoints not present; fixtures are committed, nothing to do")
    main()
