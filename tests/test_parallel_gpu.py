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


def _learn_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mobrob_amd.rl_control.ppo import PPOCtrl
    cfg = {"ppo_kwargs": {"policy": "MlpPolicy", "n_steps": 64, "batch_size": 4096, "n_epochs": 4, "gamma": 0.99,
                          "gae_lambda": 0.95, "ent_coef": 0.0, "clip_range": 0.2,
                          "policy_kwargs": {"net_arch": {"pi": [64, 64], "vf": [64, 64]}}},
           "env_name": "point", "time_limit": 100, "n_envs": 128, "vec_env_type": "device_goal", "enable_gui": False,
           "seed": 0}
    ctrl = PPOCtrl.from_config(cfg)
    ppo = ctrl.ppo
    assert ppo.world_size == world and ppo.rank == rank
    hist = []
    for _ in range(25):
        ppo.learn(total_timesteps=64 * 128 * world, reset_num_timesteps=False)
        st = ppo.device_episode_stats
        hist.append((st["episodes"], st["goals"]))
    ctrl.save_model(out.format(rank="model"))  # only rank 0 writes
    np.savez(out.format(rank=rank), flat=ppo.engine.get_flat_params(), hist=np.array(hist), steps=ppo.num_timesteps,
             obs0=ppo.engine.read("obs")[0])
    ppo.engine.close()
    dist.destroy_process_group()


def test_ppo_learn_is_data_parallel_under_a_process_group(tmp_path):
    """examples/train.py under torchrun: PPOCtrl picks the process group up, shards the envs, all-reduces the gradient;
    both replicas end with identical weights although they saw different environments, and the task is learned."""
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    out = str(tmp_path / "r{rank}.npz")
    mp.spawn(_learn_worker, args=(world, port, out), nprocs=world, join=True)
    r = [np.load(out.format(rank=i)) for i in range(world)]
    assert np.array_equal(r[0]["flat"], r[1]["flat"])
    assert not np.array_equal(r[0]["obs0"], r[1]["obs0"])          # different env shards
    assert int(r[0]["steps"]) == 25 * 64 * 128 * world              # time/total_timesteps counts the whole job
    assert os.path.exists(out.format(rank="model") + ".zip") or os.path.exists(out.format(rank="model"))
    h = r[0]["hist"].astype(np.float64)
    first, last = h[:4].sum(0), h[-4:].sum(0)
    assert last[1] / max(last[0], 1) > first[1] / max(first[0], 1) + 0.2, (first, last)


class _NoComm:
    """An engine whose RCCL communicator cannot be created (librccl not loadable, version mismatch ...)."""

    def __init__(self, e):
        self._e = e

    def __getattr__(self, name):
        return getattr(self._e, name)

    def comm_init(self, uid):
        raise RuntimeError("ncclCommInitRank failed (simulated)")


def _rccl_world1_worker(rank, world, port, out):
    """One rank, "nccl" backend: the C loop with the engine's own RCCL communicator (ncclCommInitRank from an id made
    by mobrob_ppo_comm_unique_id, ncclAllReduce on the engine's stream) against the single-rank C call and against
    the Python protocol loop with torch.distributed collectives."""
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    from mobrob_amd.engine import PPOEngine
    from mobrob_amd.parallel import EngineBackend, train_data_parallel
    res = {}
    for case in ("h256", "h64"):
        c = CASES[case]
        p, buf, lv, dones, h, perms = _rank_data(c, 0)
        H = c["H"]
        e = PPOEngine(obs_dim=c["D"], act_dim=c["A"], n_envs=c["N"], n_steps=c["T"], batch_size=c["B"], n_epochs=c["E"],
                      pi=(H, H), vf=(H, H), ent_coef=h.ent_coef, device_id=0)
        e.load_rollout(buf, lv, dones)
        z = {k: np.zeros_like(v) for k, v in p.items()}
        be = EngineBackend(e)
        for mode in ("single", "c_rccl", "python_torch", "no_comm"):
            e.set_params(p)
            e.set_optimizer_state(z, z, 0)
            if mode == "single":
                e.train(perms)
            elif mode == "no_comm":  # the engine cannot create its communicator: every rank falls back to torch's collectives
                import warnings
                broken = EngineBackend(e)
                broken.e = _NoComm(e)
                with warnings.catch_warnings(record=True) as w:
                    warnings.simplefilter("always")
                    train_data_parallel(broken, perms, force_collectives=True)
                assert any("falling back" in str(x.message) for x in w)
            else:
                train_data_parallel(be, perms, force_collectives=True, python_loop=(mode == "python_torch"))
            torch.cuda.synchronize()
            res[f"{case}/{mode}"] = e.get_flat_params()
        e.close()
    np.savez(out, **res)
    dist.destroy_process_group()


def test_c_loop_with_rccl_equals_the_single_rank_call(tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / "rccl.npz")
    mp.spawn(_rccl_world1_worker, args=(1, _free_port(), out), nprocs=1, join=True)
    r = np.load(out)
    for case in ("h256", "h64"):
        assert np.array_equal(r[f"{case}/c_rccl"], r[f"{case}/single"]), case
        assert np.array_equal(r[f"{case}/python_torch"], r[f"{case}/single"]), case
        assert np.array_equal(r[f"{case}/no_comm"], r[f"{case}/single"]), case   # fallback when the communicator is unavailable
