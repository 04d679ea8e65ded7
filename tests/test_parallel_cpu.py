"""CPU (gloo, world_size 2): the data-parallel update loop of mobrob_amd/parallel.py.

The HIP engine cannot run here, so the loop is driven with a NumPy backend that implements the same protocol
with the oracle's arithmetic.  What is tested is the exchange logic that the GPU ranks execute unchanged:
  * advantage statistics all-reduced once per epoch -> global mean/std,
  * one gradient all-reduce (sum) per optimizer step, losses scaled by 1/B_global,
  * identical clip + Adam on every rank -> replicas stay identical, and equal to the single-process oracle
    on the union minibatch.
"""
import os
import socket
import sys
from collections import OrderedDict

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ppo_oracle as O  # noqa: E402
from tests.util import synthetic_rollout  # noqa: E402


class OracleBackend:
    """mobrob_amd.parallel backend protocol on top of the NumPy oracle (test double for the HIP engine)."""

    def __init__(self, params, buf, hyper, world):
        self.p, self.buf, self.h, self.world = params, buf, hyper, world
        self.st = O.AdamState.zeros_like(params)
        self.T, self.N = buf["rewards"].shape
        self.bl = hyper.batch_size // world
        self.n_minibatches = -(-self.T * self.N // self.bl)
        self.n_epochs = hyper.n_epochs
        self.shapes = OrderedDict((k, v.shape) for k, v in params.items())
        self._adv = torch.zeros(self.n_minibatches, 4, dtype=torch.float64)
        self._grad = torch.zeros(sum(v.size for v in params.values()), dtype=torch.float32)

    def epoch_begin(self, perm):
        self.perm = np.asarray(perm)
        for mb in range(self.n_minibatches):
            idx = self.perm[mb * self.bl:(mb + 1) * self.bl]
            a = O.gather_minibatch(self.buf, idx)[4].astype(np.float64)
            self._adv[mb] = torch.tensor([a.sum(), (a * a).sum(), len(a), 0.0])

    def advstat_tensor(self):
        return self._adv

    def minibatch_grad(self, mb):
        s, s2, n, _ = self._adv[mb].tolist()
        mean = s / n
        std = np.sqrt(max((s2 - n * mean * mean) / (n - 1), 0.0))
        idx = self.perm[mb * self.bl:(mb + 1) * self.bl]
        _, grads, _ = O.loss_and_grads(self.p, *O.gather_minibatch(self.buf, idx), self.h,
                                       adv_mean_std=(mean, std), denom=int(n))
        self._grad.copy_(torch.from_numpy(O.flatten_params(grads)))

    def grad_tensor(self):
        return self._grad

    def minibatch_apply(self):
        g = O.unflatten_params(self._grad.numpy(), self.shapes)
        g, _ = O.clip_grad_norm(g, self.h.max_grad_norm)
        O.adam_step(self.p, g, self.st, self.h.learning_rate, self.h.beta1, self.h.beta2, self.h.adam_eps)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mobrob_amd.parallel import train_data_parallel
    D, A, T, N, B, E = 14, 2, 12, 8, 32, 2  # N envs PER RANK; global minibatch 32 = 16 per rank
    p = O.init_params(D, A, seed=4)
    p["log_std"] = np.full(A, -0.5, np.float32)
    buf, lv, dones = synthetic_rollout(T, N, D, A, seed=100 + rank)  # each rank owns different envs
    mean, val = O.policy_outputs(p, buf["obs"].reshape(T * N, D))
    buf["log_probs"] = O.gaussian_log_prob(mean, p["log_std"], buf["actions"].reshape(T * N, A)).reshape(T, N)
    buf["values"] = val.reshape(T, N)
    h = O.Hyper(n_epochs=E, batch_size=B, ent_coef=0.01)
    buf["advantages"], buf["returns"] = O.gae(buf["rewards"], buf["values"], buf["episode_starts"], lv, dones, h.gamma, h.gae_lambda)
    rng = np.random.default_rng(7 + rank)
    perms = np.stack([rng.permutation(T * N) for _ in range(E)])
    be = OracleBackend(p, buf, h, world)
    train_data_parallel(be, perms)
    np.savez(out.format(rank=rank), perms=perms, flat=O.flatten_params(p), **{f"buf_{k}": v for k, v in buf.items()})
    dist.destroy_process_group()


def test_two_rank_update_equals_single_process_on_union_batch(tmp_path):
    world, port = 2, _free_port()
    out = str(tmp_path / "rank{rank}.npz")
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    r = [np.load(out.format(rank=i)) for i in range(world)]
    assert np.array_equal(r[0]["flat"], r[1]["flat"])  # replicas stay identical without a broadcast
    # single-process reference: every global minibatch is the union of the ranks' local slices
    D, A, T, N, B, E = 14, 2, 12, 8, 32, 2
    p = O.init_params(D, A, seed=4)
    p["log_std"] = np.full(A, -0.5, np.float32)
    st = O.AdamState.zeros_like(p)
    h = O.Hyper(n_epochs=E, batch_size=B, ent_coef=0.01)
    bufs = [{k[4:]: r[i][k] for k in r[i].files if k.startswith("buf_")} for i in range(world)]
    bl = B // world
    for e in range(E):
        for mb in range(T * N // bl):
            parts = [O.gather_minibatch(bufs[i], r[i]["perms"][e][mb * bl:(mb + 1) * bl]) for i in range(world)]
            batch = tuple(np.concatenate([parts[i][j] for i in range(world)]) for j in range(6))
            O.train_minibatch(p, st, batch, h)
    assert np.max(np.abs(O.flatten_params(p) - r[0]["flat"])) < 2e-6


def test_single_process_loop_needs_no_process_group():
    from mobrob_amd.parallel import train_data_parallel
    D, A, T, N = 6, 2, 5, 4
    p = O.init_params(D, A, seed=1)
    buf, lv, dones = synthetic_rollout(T, N, D, A, seed=3)
    h = O.Hyper(n_epochs=1, batch_size=10)
    buf["advantages"], buf["returns"] = O.gae(buf["rewards"], buf["values"], buf["episode_starts"], lv, dones, 0.99, 0.95)
    q = {k: v.copy() for k, v in p.items()}
    be = OracleBackend(p, buf, h, 1)
    perm = np.arange(T * N)[None]
    train_data_parallel(be, perm)
    O.train(q, O.AdamState.zeros_like(q), buf, h, perm)
    assert np.max(np.abs(O.flatten_params(p) - O.flatten_params(q))) < 1e-6
