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
         "generic": dict(D=26, A=2, H=32, T=8, N=16, B=64, E=1),
         # the generic chain's newer keyword surface under data parallel: unequal depths, ELU, gSDE (log_std [HL][A] in the all-reduced vector)
         "generic_sde": dict(D=26, A=2, H=32, pi=(32, 24), vf=(32,), act="elu", sde=True, T=8, N=16, B=64, E=2)}


def _arch(c):
    return c.get("pi", (c["H"], c["H"])), c.get("vf", (c["H"], c["H"])), c.get("act", "tanh"), bool(c.get("sde", False))


def _rank_data(c, rank):
    D, A, T, N = c["D"], c["A"], c["T"], c["N"]
    pi, vf, act, sde = _arch(c)
    p = O.init_params(D, A, pi, vf, seed=4)
    p["log_std"] = np.full((pi[-1], A) if sde else A, -0.5, np.float32)
    buf, lv, dones = synthetic_rollout(T, N, D, A, seed=100 + rank)
    flat = buf["obs"].reshape(T * N, D)
    mean, val = O.policy_outputs(p, flat, activation=act)
    if sde:
        sigma = O.sde_sigma(O.mlp_latents(p, flat, activation=act)[0][-1], p["log_std"])
        buf["log_probs"] = O.normal_log_prob(mean, sigma, buf["actions"].reshape(T * N, A)).reshape(T, N)
    else:
        buf["log_probs"] = O.gaussian_log_prob(mean, p["log_std"], buf["actions"].reshape(T * N, A)).reshape(T, N)
    buf["values"] = val.reshape(T, N)
    h = O.Hyper(n_epochs=c["E"], batch_size=c["B"], ent_coef=0.01, activation=act, use_sde=sde)
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
    pi, vf, act, sde = _arch(c)
    e = PPOEngine(obs_dim=c["D"], act_dim=c["A"], n_envs=c["N"], n_steps=c["T"], batch_size=c["B"], n_epochs=c["E"],
                  pi=pi, vf=vf, activation=act, use_sde=sde, ent_coef=h.ent_coef, device_id=0, rank=rank, world_size=world)
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
    if not case.startswith("generic"):  # fused paths are deterministic -> replicas stay bit-identical without a broadcast
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


def _kl_worker(rank, world, port, case, out, target):
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
                  pi=(H, H), vf=(H, H), ent_coef=h.ent_coef, learning_rate=3e-3, device_id=0, rank=rank, world_size=world)
    e.set_hyper(target_kl=target)
    e.set_params(p)
    e.load_rollout(buf, lv, dones)
    be = EngineBackend(e)
    res = {}
    for mode in ("c_loop",):
        e.set_params(p)
        z = {k: np.zeros_like(v) for k, v in p.items()}
        e.set_optimizer_state(z, z, 0)
        info = train_data_parallel(be, perms, python_loop=(mode == "python_loop"))
        torch.cuda.synchronize()
        st = e.train_stats()
        res[mode + "/flat"] = e.get_flat_params()
        res[mode + "/info"] = np.array(info, dtype=np.int64)
        res[mode + "/adam_step"] = e.get_optimizer_state()[2]
        res[mode + "/stats"] = np.array([st[k] for k in ("policy_loss", "value_loss", "approx_kl", "clip_fraction", "grad_norm")])
    np.savez(out.format(rank=rank), **res)
    e.close()
    dist.destroy_process_group()


def _union_run(c, world, target, lr):
    data = [_rank_data(c, i) for i in range(world)]
    p = {k: v.copy() for k, v in data[0][0].items()}
    st = O.AdamState.zeros_like(p)
    h = O.Hyper(n_epochs=c["E"], batch_size=c["B"], ent_coef=0.01, learning_rate=lr, target_kl=target)
    bl, total, rows = c["B"] // world, c["T"] * c["N"], []
    for ep in range(c["E"]):
        rows_ep = []
        for mb in range(-(-total // bl)):
            parts = [O.gather_minibatch(data[i][1], data[i][5][ep][mb * bl:(mb + 1) * bl]) for i in range(world)]
            batch = tuple(np.concatenate([parts[i][j] for i in range(world)]) for j in range(6))
            s = O.train_minibatch(p, st, batch, h)
            rows_ep.append(s)
            rows.append(s)
            if s.get("early_stop"):
                return p, st, rows, rows_ep, ep + 1
    return p, st, rows, rows_ep, c["E"]


@pytest.mark.parametrize("case", ["h256", "h64"])
def test_target_kl_and_logged_statistics_are_global_under_data_parallel(case, tmp_path):
    """SB3's target_kl with two ranks: the approx_kl sum travels with the gradient, so both ranks stop at the step at
    which single-process SB3 arithmetic on the UNION minibatch stops, and the logged statistics are the union's."""
    import torch.multiprocessing as mp
    c, world = CASES[case], 2
    _, _, rows, _, _ = _union_run(c, world, None, 3e-3)
    kls = [float(r["approx_kl"]) for r in rows]
    first = next(i for i in range(2, len(kls)) if kls[i] > 1.2 * max(kls[:i]))
    target = (max(kls[:first]) + kls[first]) / 2 / 1.5
    p_ref, st_ref, rows, rows_ep, epochs = _union_run(c, world, target, 3e-3)
    assert st_ref.step == first and len(rows) == first + 1
    out = str(tmp_path / "kl{rank}.npz")
    mp.spawn(_kl_worker, args=(world, _free_port(), case, out, target), nprocs=world, join=True)
    r = [np.load(out.format(rank=i)) for i in range(world)]
    ref = O.flatten_params(p_ref)
    for mode in ("c_loop",):
        for i in range(world):
            assert r[i][mode + "/info"].tolist() == [epochs, 1, first], (mode, r[i][mode + "/info"])
            assert int(r[i][mode + "/adam_step"]) == first
        assert np.array_equal(r[0][mode + "/flat"], r[1][mode + "/flat"])
        assert np.max(np.abs(ref - r[0][mode + "/flat"])) < 1e-4
        # statistics of the last epoch that ran: union-batch values on BOTH ranks (the dropped step has no grad norm)
        want = [np.mean([float(s[k]) for s in rows_ep]) for k in ("policy_loss", "value_loss", "approx_kl", "clip_fraction")]
        want.append(np.mean([float(s["grad_norm"]) for s in rows_ep if not s.get("early_stop")]) if len(rows_ep) > 1 else 0.0)
        for i in range(world):
            got = r[i][mode + "/stats"]
            assert np.allclose(got, want, rtol=2e-3, atol=2e-4), (mode, i, got, want)


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

    def comm_init(self, uid, rank=None, nranks=None):
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
                assert broken.exchange == "torch.distributed" and "unavailable" in broken.exchange_selfcheck["rccl"]
            else:
                train_data_parallel(be, perms, force_collectives=True, python_loop=(mode == "python_torch"))
            torch.cuda.synchronize()
            res[f"{case}/{mode}"] = e.get_flat_params()
            if mode == "c_rccl":   # ncclCommCount / ncclCommUserRank of the engine's own communicator, and the C loop's counters
                assert e.comm_info() == (1, 0)
                # the exchange was proven on a known vector at set-up, and says so (bench.py prints both)
                assert be.exchange == "rccl" and be.exchange_selfcheck == {"rccl": "ok"}
                assert e.exchange_selfcheck("rccl") == 0
                with pytest.raises(Exception, match="one-shot exchange is not open"):
                    e.exchange_selfcheck("oneshot")
                nmb = e.n_minibatches
                assert e.last_train_info() == (c["E"], False, c["E"] * nmb)
                calls, nbytes = e.allreduce_counters(reset=True)
                assert calls == c["E"] * (nmb + 1) and nbytes == c["E"] * (nmb * (e.P + 8) * 4 + nmb * 32)
        # SB3's target_kl through all three drivers: same stop step, same bits
        e.set_hyper(target_kl=1e-7, learning_rate=3e-3)     # far below any minibatch's approx_kl after the first step
        for mode in ("single", "c_rccl", "python_torch"):
            e.set_params(p)
            e.set_optimizer_state(z, z, 0)
            if mode == "single":
                e.train(perms)
                info = e.last_train_info()
            else:
                info = train_data_parallel(be, perms, force_collectives=True, python_loop=(mode == "python_torch"))
            torch.cuda.synchronize()
            res[f"{case}/kl/{mode}"] = e.get_flat_params()
            res[f"{case}/kl/{mode}/info"] = np.array(info, dtype=np.int64)
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
        info = r[f"{case}/kl/single/info"].tolist()
        assert info[1] == 1 and 1 <= info[2] < 8, info         # the first step has approx_kl == 0 and is always applied
        for mode in ("c_rccl", "python_torch"):
            assert r[f"{case}/kl/{mode}/info"].tolist() == info, (case, mode)
            assert np.array_equal(r[f"{case}/kl/{mode}"], r[f"{case}/kl/single"]), (case, mode)


# ---- one-shot all-reduce over peer-mapped memory (csrc/oneshot_allreduce.h), validated WITHOUT a second GPU -----------
def _oneshot_worker(rank, world, port, case, out):
    """Two ranks on cuda:0: each exports its exchange buffer (hipIpcGetMemHandle), maps the other's (hipIpcOpenMemHandle)
    and runs the C loop with the one-shot exchange; then the same update through the gloo-callback path."""
    import time
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                      MOBROB_ONESHOT_TIMEOUT_MS="15000")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mobrob_amd.engine import PPOEngine
    from mobrob_amd.parallel import EngineBackend, train_data_parallel
    c = CASES[case]
    p, buf, lv, dones, h, perms = _rank_data(c, rank)
    H = c["H"]
    e = PPOEngine(obs_dim=c["D"], act_dim=c["A"], n_envs=c["N"], n_steps=c["T"], batch_size=c["B"], n_epochs=c["E"],
                  pi=(H, H), vf=(H, H), ent_coef=h.ent_coef, device_id=0, rank=rank, world_size=world)
    e.load_rollout(buf, lv, dones)
    z = {k: np.zeros_like(v) for k, v in p.items()}
    res = {}
    for mode in ("callback", "oneshot"):
        os.environ["MOBROB_ONESHOT_AR"] = "1" if mode == "oneshot" else "0"
        if mode == "oneshot" and rank == 1 and os.environ.get("TEST_ONE_RANK_WITHOUT_SWITCH"):
            os.environ["MOBROB_ONESHOT_AR"] = "0"    # the switch is AGREED: one rank without it sends both to the default
        be = EngineBackend(e)           # the exchange is chosen (and self-checked) once per backend
        ms = []
        for rep in range(3):            # repetitions: sequence numbers, slot reuse and flag monotonicity over many messages
            e.set_params(p)
            e.set_optimizer_state(z, z, 0)
            dist.barrier()
            t0 = time.perf_counter()
            train_data_parallel(be, perms)
            e.synchronize()
            ms.append(1e3 * (time.perf_counter() - t0))
            if rep == 0:
                res[mode + "/flat"] = e.get_flat_params()
                res[mode + "/stats"] = np.array(list(e.train_stats().values()))
            else:
                assert np.array_equal(res[mode + "/flat"], e.get_flat_params()), (mode, rep)
        res[mode + "/ms"] = np.array(ms)
        if mode == "callback" or os.environ.get("TEST_ONE_RANK_WITHOUT_SWITCH"):
            assert be.exchange == "gloo-callback" and be.exchange_selfcheck == {}
        else:
            assert be._oneshot_ready is True
            assert be.exchange == "oneshot" and be.exchange_selfcheck == {"oneshot": "ok"}
            assert e.exchange_selfcheck("oneshot") == 0       # again, between updates: sequence numbers stay in step
            calls, _ = e.allreduce_counters()
            assert calls == 2 * 3 * c["E"] * (e.n_minibatches + 1)
    np.savez(out.format(rank=rank), **res)
    was = be.exchange
    be.close()          # closing handshake of the one-shot exchange: drain, barrier, then unmap (a no-op for the other exchanges)
    assert be.exchange == (None if was == "oneshot" else was)
    e.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("case,fence", [("h256", "0"), ("h64", "0"), ("h256", "1")])
def test_oneshot_all_reduce_through_ipc_equals_the_callback_path(case, fence, tmp_path, monkeypatch):
    """fence "0": write-through publish / system-scope loads (the default); "1": round 3's release / acquire fences."""
    import torch.multiprocessing as mp
    monkeypatch.setenv("MOBROB_ONESHOT_FENCE", fence)
    world = 2
    out = str(tmp_path / "os{rank}.npz")
    mp.spawn(_oneshot_worker, args=(world, _free_port(), case, out), nprocs=world, join=True)
    r = [np.load(out.format(rank=i)) for i in range(world)]
    assert np.array_equal(r[0]["oneshot/flat"], r[1]["oneshot/flat"])            # replicas bit-identical
    assert np.array_equal(r[0]["oneshot/flat"], r[0]["callback/flat"])           # x0 + x1 is the same sum in any transport
    assert np.array_equal(r[0]["oneshot/stats"], r[0]["callback/stats"])
    print(f"\n[{case}] world-2 update on one device, ms per train(): host-staged gloo callback "
          f"{np.min(r[0]['callback/ms']):.2f}, one-shot peer exchange {np.min(r[0]['oneshot/ms']):.2f}")
    rec = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(rec):
        import json
        with open(os.path.join(rec, f"oneshot_vs_callback_{case}_fence{fence}.json"), "w") as f:
            json.dump({"case": CASES[case], "world": world, "publish": "release fence" if fence == "1" else "write-through stores", "device": "both ranks on cuda:0",
                       "ms_per_train_callback": r[0]["callback/ms"].tolist(), "ms_per_train_oneshot": r[0]["oneshot/ms"].tolist()}, f)


def test_oneshot_switch_on_one_rank_only_falls_back_on_both(tmp_path, monkeypatch):
    import torch.multiprocessing as mp
    monkeypatch.setenv("TEST_ONE_RANK_WITHOUT_SWITCH", "1")
    out = str(tmp_path / "sw{rank}.npz")
    mp.spawn(_oneshot_worker, args=(2, _free_port(), "h64", out), nprocs=2, join=True)
    r = [np.load(out.format(rank=i)) for i in range(2)]
    assert np.array_equal(r[0]["oneshot/flat"], r[1]["oneshot/flat"]) and np.array_equal(r[0]["oneshot/flat"], r[0]["callback/flat"])


def _oneshot_dead_peer_worker(rank, world, port, out):
    """Rank 1 sets the exchange up and then never trains: rank 0's kernel must give up after the timeout and the engine
    must report it, not hang the device."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                      MOBROB_ONESHOT_TIMEOUT_MS="300", MOBROB_ONESHOT_AR="1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mobrob_amd._lib import EngineError
    from mobrob_amd.engine import PPOEngine
    from mobrob_amd.parallel import EngineBackend, train_data_parallel
    c = CASES["h64"]
    p, buf, lv, dones, h, perms = _rank_data(c, rank)
    e = PPOEngine(obs_dim=c["D"], act_dim=c["A"], n_envs=c["N"], n_steps=c["T"], batch_size=c["B"], n_epochs=1,
                  pi=(64, 64), vf=(64, 64), device_id=0, rank=rank, world_size=world)
    e.set_params(p)
    e.load_rollout(buf, lv, dones)
    be = EngineBackend(e)
    assert be.choose_exchange() == "oneshot" and be.exchange_selfcheck == {"oneshot": "ok"}   # both ranks alive at set-up
    msg = ""
    if rank == 0:
        try:
            train_data_parallel(be, perms[:1])
            e.synchronize()
        except EngineError as ex:
            msg = str(ex)
    dist.barrier()
    np.savez(out.format(rank=rank), msg=msg)
    e.close()
    dist.destroy_process_group()


def test_oneshot_all_reduce_reports_a_peer_that_never_arrives(tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / "dead{rank}.npz")
    mp.spawn(_oneshot_dead_peer_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert "never published" in str(np.load(out.format(rank=0))["msg"])
