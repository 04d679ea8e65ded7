"""GPU: two data-parallel ranks (two processes, gloo rendezvous, both on cuda:0 -- RCCL refuses two ranks on one
device, the collective's arithmetic is the same sum) drive the REAL engine through mobrob_amd/parallel.py.
Checks what only shows up with world_size > 1: global advantage statistics, 1/B_global loss scaling, gradient sum,
identical clip + Adam on every rank -> replicas bit-identical and equal to single-process SB3 arithmetic on the
union minibatch (oracle)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ppo_oracle as O  # noqa: E402
from tests.test_parallel_cpu import _free_port  # noqa: E402
from tests.util import synthetic_rollout  # noqa: E402

pytestmark = pytest.mark.gpu

CASES = {"h256": dict(D=58, A=12, H=256, T=12, N=40, B=320, E=2), "h64": dict(D=14, A=2, H=64, T=10, N=24, B=96, E=2),
         "generic": dict(D=26, A=2, H=32, T=8, N=16, B=64, E=1)}


def _rank_data(c, rank):
    D, A, H, T, N = c["D"], c["A"], c["H"], c["T"], c["N"]
    p = O.init_params(D, A, (H, H), (H, H), seed=4)
    p["log_std"] = np.full(A, -0.5, np.float32)
    buf, lv, dones = synthetic_rollout(T, N, D, A, seed=100 + rank)
    mean, val = O.policy_outputs(p, buf["obs"].reshape(T * N, D))
    buf["log_probs"] = O.gaussian_log_prob(mean, p["log_std"], buf["actions"].reshape(T * N, A)).reshape(T, N)
    buf["values"] = val.reshape(T, N)
    h = O.Hyper(n_epochs=c["E"], batch_size=c["B"], ent_coef=0.01)
    buf["advantages"], buf["returns"] = O.gae(buf["rewards"], buf["values"], buf["episode_starts"], lv, dones, h.gamma, h.gae_lambda)
    perms = np.stack([np.random.default_rng(7 + rank + 10 * e).permutation(T * N) for e in range(c["E"])])
    return p, buf, lv, dones, h, perms


def _worker(rank, world, port, case, out):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mobrob_amd.engine import PPOEngine
    from mobrob_amd.parallel import EngineBackend, train_data_parallel
    c = CASES[case]
    p, buf, lv, dones, h, perms = _rank_data(c, rank)
    H = c["H"]
    e = PPOEngine(obs_dim=c["D"], act_dim=c["A"], n_envs=c["N"], n_steps=c["T"], batch_size=c["B"], n_epochs=c["E"],
                  pi=(H, H), vf=(H, H), ent_coef=h.ent_coef, device_id=0, rank=rank, world_size=world)
    e.set_params(p)
    e.load_rollout(buf, lv, dones)
    be = EngineBackend(e)
    train_data_parallel(be, perms)
    torch.cuda.synchronize()
    np.savez(out.format(rank=rank), flat=e.get_flat_params())
    e.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("case", list(CASES))
def test_two_engine_ranks_equal_single_process_union_batch(case, tmp_path):
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    out = str(tmp_path / "rank{rank}.npz")
    mp.spawn(_worker, args=(world, port, case, out), nprocs=world, join=True)
    r = [np.load(out.format(rank=i))["flat"] for i in range(world)]
    if case != "generic":  # fused paths are deterministic -> replicas stay bit-identical without a broadcast
        assert np.array_equal(r[0], r[1])
    assert np.max(np.abs(r[0] - r[1])) < 1e-6
    c = CASES[case]
    data = [_rank_data(c, i) for i in range(world)]
    p = {k: v.copy() for k, v in data[0][0].items()}
    st = O.AdamState.zeros_like(p)
    h = data[0][4]
    bl = c["B"] // world
    total = c["T"] * c["N"]
    for ep in range(c["E"]):
        for mb in range(-(-total // bl)):
            parts = [O.gather_minibatch(data[i][1], data[i][5][ep][mb * bl:(mb + 1) * bl]) for i in range(world)]
            batch = tuple(np.concatenate([parts[i][j] for i in range(world)]) for j in range(6))
            O.train_minibatch(p, st, batch, h)
    ref = O.flatten_params(p)
    assert np.max(np.abs(ref - r[0])) < 1e-4, float(np.max(np.abs(ref - r[0])))
