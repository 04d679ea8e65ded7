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
        self.P = sum(v.size for v in params.values())
        # like the engine's exchange message: [P] gradient followed by loss sums (here just the approx_kl sum)
        self._grad = torch.zeros(self.P + 1, dtype=torch.float32)

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
        stats, grads, _ = O.loss_and_grads(self.p, *O.gather_minibatch(self.buf, idx), self.h,
                                           adv_mean_std=(mean, std), denom=int(n))
        self._grad[:self.P].copy_(torch.from_numpy(O.flatten_params(grads)))
        self._grad[self.P] = float(stats["approx_kl"])  # local sum / B_global: the ranks' values ADD to the global mean

    def grad_tensor(self):
        return self._grad

    def minibatch_apply(self):
        if self.h.target_kl is not None and float(self._grad[self.P]) > 1.5 * self.h.target_kl:
            return True  # SB3: continue_training = False; break -- before optimizer.step()
        g = O.unflatten_params(self._grad[:self.P].numpy(), self.shapes)
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


# ---- target_kl across ranks: the stop decision is taken from the GLOBAL approx_kl on every rank alike ----------------
def _kl_worker(rank, world, port, out, target):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mobrob_amd.parallel import train_data_parallel
    p, buf, h, perms = _kl_case(rank, target)
    be = OracleBackend(p, buf, h, world)
    info = train_data_parallel(be, perms)
    np.savez(out.format(rank=rank), flat=O.flatten_params(p), info=np.array(info, dtype=np.int64), adam_step=be.st.step)
    dist.destroy_process_group()


def _kl_case(rank, target):
    D, A, T, N, B, E = 14, 2, 12, 8, 32, 3
    p = O.init_params(D, A, seed=4)
    p["log_std"] = np.full(A, -0.5, np.float32)
    buf, lv, dones = synthetic_rollout(T, N, D, A, seed=100 + rank)
    mean, val = O.policy_outputs(p, buf["obs"].reshape(T * N, D))
    buf["log_probs"] = O.gaussian_log_prob(mean, p["log_std"], buf["actions"].reshape(T * N, A)).reshape(T, N)
    buf["values"] = val.reshape(T, N)
    h = O.Hyper(n_epochs=E, batch_size=B, ent_coef=0.0, learning_rate=3e-3, target_kl=target)
    buf["advantages"], buf["returns"] = O.gae(buf["rewards"], buf["values"], buf["episode_starts"], lv, dones, h.gamma, h.gae_lambda)
    perms = np.stack([np.random.default_rng(7 + rank + 10 * e).permutation(T * N) for e in range(E)])
    return p, buf, h, perms


def _union_reference(world, target):
    data = [_kl_case(r, target) for r in range(world)]
    p, h = {k: v.copy() for k, v in data[0][0].items()}, data[0][2]
    st = O.AdamState.zeros_like(p)
    bl, total, kls = h.batch_size // world, 12 * 8, []
    for ep in range(h.n_epochs):
        for mb in range(total // bl):
            parts = [O.gather_minibatch(data[i][1], data[i][3][ep][mb * bl:(mb + 1) * bl]) for i in range(world)]
            batch = tuple(np.concatenate([parts[i][j] for i in range(world)]) for j in range(6))
            stats = O.train_minibatch(p, st, batch, h)
            kls.append(float(stats["approx_kl"]))
            if stats.get("early_stop"):
                return p, st, kls, ep + 1
    return p, st, kls, h.n_epochs


def test_target_kl_stops_every_rank_at_the_step_the_union_batch_stops(tmp_path):
    world = 2
    _, _, kls, _ = _union_reference(world, None)
    first = next(i for i in range(2, len(kls)) if kls[i] > max(kls[:i]) * 1.05)       # a crossing with margin on both sides
    target = (max(kls[:first]) + kls[first]) / 2 / 1.5
    p_ref, st_ref, kls_stop, epochs = _union_reference(world, target)
    assert len(kls_stop) == first + 1 and st_ref.step == first
    out = str(tmp_path / "kl{rank}.npz")
    mp.spawn(_kl_worker, args=(world, _free_port(), out, target), nprocs=world, join=True)
    r = [np.load(out.format(rank=i)) for i in range(world)]
    for i in range(world):
        assert r[i]["info"].tolist() == [epochs, 1, first] and int(r[i]["adam_step"]) == first
    assert np.array_equal(r[0]["flat"], r[1]["flat"])
    assert np.max(np.abs(O.flatten_params(p_ref) - r[0]["flat"])) < 2e-6


# ---- communicator set-up: a rank that cannot take part must not leave the others inside ncclCommInitRank ------------
class _FakeEngine:
    """What agree_and_init_comm needs of PPOEngine; `fail_at` makes one step raise on this rank."""

    class cfg:  # noqa: N801 - mirrors engine.cfg.world_size
        world_size = 2

    def __init__(self, fail_at=None):
        self.fail_at, self.calls = fail_at, []

    def comm_prepare(self):
        self.calls.append("prepare")
        if self.fail_at == "prepare":
            raise RuntimeError("RCCL not found (simulated)")

    def comm_unique_id(self):
        self.calls.append("id")
        return bytes(range(128))

    def comm_init(self, uid, rank=None, nranks=None):
        self.calls.append(("init", bytes(uid), rank, nranks))
        if self.fail_at == "init":
            raise RuntimeError("ncclCommInitRank failed (simulated)")


def _comm_worker(rank, world, port, out, scenario):
    import warnings
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mobrob_amd.parallel import agree_and_init_comm
    group = None
    if scenario == "subgroup":            # ranks 1 and 2 of a 3-rank job form the communicator: group rank 0 is global rank 1
        group = dist.new_group([1, 2])
        if rank == 0:
            dist.destroy_process_group()
            return
    e = _FakeEngine(fail_at={"prepare_fails_on_rank1": "prepare", "init_fails_on_rank1": "init"}.get(scenario) if rank == 1 else None)
    if scenario == "size_mismatch":
        e.cfg = type("cfg", (), {"world_size": 4})
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        ready = agree_and_init_comm(e, group, None)
    np.savez(out.format(rank=rank), ready=ready, warned=len(w), calls=np.array([repr(c) for c in e.calls]))
    dist.destroy_process_group()


@pytest.mark.parametrize("scenario", ["ok", "prepare_fails_on_rank1", "init_fails_on_rank1", "size_mismatch", "subgroup"])
def test_communicator_setup_agrees_before_the_blocking_collective(scenario, tmp_path):
    world = 3 if scenario == "subgroup" else 2
    out = str(tmp_path / "c{rank}.npz")
    mp.spawn(_comm_worker, args=(world, _free_port(), out, scenario), nprocs=world, join=True)   # would hang if a rank waited alone
    ranks = [1, 2] if scenario == "subgroup" else [0, 1]
    r = {i: np.load(out.format(rank=i)) for i in ranks}
    calls = {i: [c for c in r[i]["calls"].tolist()] for i in ranks}
    want_ready = scenario in ("ok", "subgroup")
    assert all(bool(r[i]["ready"]) == want_ready for i in ranks)
    if scenario in ("prepare_fails_on_rank1", "size_mismatch"):
        assert not any("init" in c for i in ranks for c in calls[i])        # nobody entered the collective
    if scenario == "prepare_fails_on_rank1":
        assert int(r[0]["warned"]) == 1                                       # rank 0 says why the job is slower
    if want_ready:
        uid = repr(bytes(range(128)))
        for gr, i in enumerate(ranks):                                        # rank IN THE GROUP, size OF THE GROUP
            assert calls[i][-1] == repr(("init", bytes(range(128)), gr, 2)), calls[i]
        assert "'id'" in calls[ranks[0]] and "'id'" not in calls[ranks[1]]
