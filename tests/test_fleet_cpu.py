"""CPU: host logic of the mixed fleet -- arena sizing pass and the interleaved data-parallel loop (gloo, world 2)."""
import os
import sys

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ppo_oracle as O  # noqa: E402
from tests.test_parallel_cpu import OracleBackend, _free_port  # noqa: E402
from tests.util import synthetic_rollout  # noqa: E402


def test_device_bytes_is_a_host_only_sizing_pass():
    from mobrob_amd.engine import PPOEngine
    from mobrob_amd.fleet import ROBOT_DIMS
    kw = dict(n_envs=256, n_steps=128, batch_size=4096)
    sizes = {n: PPOEngine.device_bytes(obs_dim=d, act_dim=a, **kw) for n, (d, a) in ROBOT_DIMS.items()}
    for n, (d, a) in ROBOT_DIMS.items():
        rollout = 128 * 256 * (8 * ((d + 7) // 8) + a + 6) * 4  # obs (padded) + actions + 6 scalars per transition
        assert sizes[n] % 256 == 0 and sizes[n] > rollout
    assert sizes["doggo"] > sizes["turtlebot3"] > sizes["car"]  # wider observations -> larger segment
    big = PPOEngine.device_bytes(obs_dim=26, act_dim=2, n_envs=512, n_steps=128, batch_size=4096)
    assert big > sizes["car"]


def _segment(rank, world, D, A, T, N, B, E, seed):
    p = O.init_params(D, A, seed=seed)
    p["log_std"] = np.full(A, -0.3, np.float32)
    buf, lv, dones = synthetic_rollout(T, N, D, A, seed=50 * seed + rank)
    mean, val = O.policy_outputs(p, buf["obs"].reshape(T * N, D))
    buf["log_probs"] = O.gaussian_log_prob(mean, p["log_std"], buf["actions"].reshape(T * N, A)).reshape(T, N)
    buf["values"] = val.reshape(T, N)
    h = O.Hyper(n_epochs=E, batch_size=B, ent_coef=0.01)
    buf["advantages"], buf["returns"] = O.gae(buf["rewards"], buf["values"], buf["episode_starts"], lv, dones, h.gamma, h.gae_lambda)
    perms = np.stack([np.random.default_rng(seed + 9 * rank + e).permutation(T * N) for e in range(E)])
    return p, buf, h, perms


SEGS = [dict(D=26, A=2, T=10, N=4, B=16, E=2, seed=1), dict(D=12, A=18, T=6, N=6, B=24, E=1, seed=2),
        dict(D=43, A=2, T=8, N=5, B=20, E=2, seed=3)]  # ragged: different dims, minibatch counts and epoch counts


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mobrob_amd.fleet import train_fleet_data_parallel
    from mobrob_amd.parallel import train_data_parallel
    res = {}
    inter, perms = [], []
    for s in SEGS:
        p, buf, h, pm = _segment(rank, world, **s)
        inter.append(OracleBackend(p, buf, h, world))
        perms.append(pm)
    train_fleet_data_parallel(inter, [None] * len(inter), perms=perms)
    for i, s in enumerate(SEGS):  # the same segments, one after the other, through the single-learner loop
        p, buf, h, pm = _segment(rank, world, **s)
        train_data_parallel(OracleBackend(p, buf, h, world), pm)
        res[f"seq{i}"] = O.flatten_params(p)
        res[f"fleet{i}"] = O.flatten_params(inter[i].p)
    np.savez(out.format(rank=rank), **res)
    dist.destroy_process_group()


def test_interleaved_fleet_loop_equals_per_segment_loops(tmp_path):
    world, port = 2, _free_port()
    out = str(tmp_path / "rank{rank}.npz")
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    r = [np.load(out.format(rank=i)) for i in range(world)]
    for i in range(len(SEGS)):
        assert np.array_equal(r[0][f"fleet{i}"], r[1][f"fleet{i}"])      # replicas identical
        assert np.array_equal(r[0][f"fleet{i}"], r[0][f"seq{i}"])        # interleaving changes nothing
