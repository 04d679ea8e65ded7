"""CPU: `ShmVecEnv` (worker processes + one shared block; what `vec_env_type: subproc` selects) against the
in-process `HostVecEnv` on the same seeds -- the reference's SubprocVecEnv / DummyVecEnv pair
(/root/reference/src/mobrob/rl_control/ppo.py:30-33) must be interchangeable."""
import functools
import glob
import os

import numpy as np
import pytest

from mobrob_amd.envs.shm_vec_env import ShmVecEnv, owned_rows
from mobrob_amd.envs.vec_env import HostVecEnv, make_vec_env
from mobrob_amd.envs.wrapper import get_env


def _pair(env_name, n, time_limit, seed, **kw):
    env_kwargs = dict(env_name=env_name, enable_gui=False, terminate_on_goal=True, time_limit=time_limit)
    host = make_vec_env(get_env, n, env_kwargs, HostVecEnv, seed=seed)
    shm = make_vec_env(get_env, n, env_kwargs, functools.partial(ShmVecEnv, **kw), seed=seed)
    return host, shm


def test_rows_are_dealt_in_blocks_round_robin():
    rows = [owned_rows(w, 3, 13, 2) for w in range(3)]
    assert sorted(np.concatenate(rows).tolist()) == list(range(13))
    assert rows[0].tolist() == [0, 1, 6, 7, 12] and rows[1].tolist() == [2, 3, 8, 9] and rows[2].tolist() == [4, 5, 10, 11]
    half = [r[(r >= 0) & (r < 6)] for r in rows]            # any contiguous range keeps every worker busy
    assert all(len(h) == 2 for h in half)


@pytest.mark.parametrize("env_name", ["point", "drone"])
def test_rollout_is_bit_identical_to_the_in_process_vec_env(env_name):
    n, T = 13, 70
    host, shm = _pair(env_name, n, time_limit=20, seed=5, n_workers=3, block=2)
    try:
        assert (shm.num_envs, shm.obs_dim, shm.act_dim) == (host.num_envs, host.obs_dim, host.act_dim)
        assert np.array_equal(shm.action_space.low, host.action_space.low) and shm.observation_space.shape == host.observation_space.shape
        o_h, o_s = host.reset(), shm.reset()
        assert o_s.dtype == np.float32 and np.array_equal(o_h, o_s)
        rng = np.random.default_rng(0)
        n_done = n_trunc = 0
        for t in range(T):
            a = rng.uniform(-1, 1, (n, host.act_dim)).astype(np.float32)
            oh, rh, dh, ih = host.step(a)
            os_, rs, ds, is_ = shm.step(a)
            assert np.array_equal(oh, os_) and np.array_equal(rh, rs) and np.array_equal(dh, ds), t
            for i in range(n):
                assert ih[i].get("TimeLimit.truncated", False) == is_[i].get("TimeLimit.truncated", False)
                if dh[i]:
                    assert np.array_equal(ih[i]["terminal_observation"], is_[i]["terminal_observation"])
                    assert ih[i]["episode"]["r"] == is_[i]["episode"]["r"] and ih[i]["episode"]["l"] == is_[i]["episode"]["l"]
                    n_done += 1
                    n_trunc += ih[i]["TimeLimit.truncated"]
                else:
                    assert "episode" not in is_[i]
        assert n_done >= n * (T // 20) and n_trunc > 0
        st = shm.episode_stats()
        assert st["episodes"] == n_done and len(shm.pop_episodes()) == n_done and shm.pop_episodes() == []
    finally:
        host.close()
        shm.close()


def test_step_range_halves_equal_a_whole_step_and_report_truncations():
    n = 12
    host, shm = _pair("point", n, time_limit=4, seed=1, n_workers=4)
    try:
        host.reset()
        shm.reset()
        rng = np.random.default_rng(1)
        b = shm.buffers()
        for t in range(9):
            a = rng.uniform(-1, 1, (n, 2)).astype(np.float32)
            oh, rh, dh, ih = host.step(a)
            b["clip"][:] = a                                  # the GPU writes clipped actions here
            nt = shm.step_range(0, 5, b["clip"]) + shm.step_range(5, n, b["clip"])
            assert np.array_equal(b["obs"], oh) and np.array_equal(b["rew"], rh) and np.array_equal(b["done"].astype(bool), dh)
            assert nt == sum(bool(i.get("TimeLimit.truncated", False)) for i in ih) == int(b["trunc"].sum())
            for i in np.nonzero(dh)[0]:
                assert np.array_equal(b["term"][i], ih[i]["terminal_observation"])
    finally:
        host.close()
        shm.close()


def test_reseeding_reproduces_a_rollout_and_env_method_reaches_the_instance():
    _, shm = _pair("car", 6, time_limit=50, seed=3, n_workers=2)
    try:
        a = np.full((6, 2), 0.3, np.float32)
        first = shm.reset().copy()
        r1 = [shm.step(a)[1] for _ in range(5)]
        shm.seed(3)
        assert np.array_equal(shm.reset(), first)
        r2 = [shm.step(a)[1] for _ in range(5)]
        assert all(np.array_equal(x, y) for x, y in zip(r1, r2))
        goal = shm.env_method("get_goal", 4)
        assert goal.shape == (2,) and not shm.env_method("reached", 4, 1e-9)
    finally:
        shm.close()


def test_close_leaves_nothing_in_dev_shm_and_a_dead_worker_is_reported():
    before = set(glob.glob("/dev/shm/mobrob_vecenv_*"))
    _, shm = _pair("point", 4, time_limit=10, seed=0, n_workers=2)
    assert set(glob.glob("/dev/shm/mobrob_vecenv_*")) == before   # the name is unlinked once every worker has it mapped
    shm.reset()
    victim = shm._procs[1]
    victim.terminate()                                            # our own child, by handle
    victim.join(timeout=10)
    with pytest.raises(RuntimeError, match="worker 1 died"):
        for _ in range(3):
            shm.step(np.zeros((4, 2), np.float32))
    shm.close()
    assert all(not p.is_alive() for p in [victim])


def test_unknown_environment_fails_at_construction():
    with pytest.raises(Exception):
        ShmVecEnv([functools.partial(get_env, env_name="unicycle")] * 2, n_workers=1)


def test_workers_leave_when_the_learner_process_is_killed(tmp_path):
    """A learner that dies without close() (SIGKILL) must not leave worker processes behind: they notice within their
    wake-up timeout."""
    import signal
    import subprocess
    import sys
    import time
    script = tmp_path / "learner.py"
    script.write_text(
        "import functools, os, sys, time\n"
        f"sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})\n"
        "from mobrob_amd.envs.shm_vec_env import ShmVecEnv\n"
        "from mobrob_amd.envs.wrapper import get_env\n"
        "if __name__ == '__main__':\n"
        "    env = ShmVecEnv([functools.partial(get_env, env_name='point')] * 4, n_workers=2)\n"
        "    env.reset()\n"
        "    print(' '.join(str(p.pid) for p in env._procs), flush=True)\n"
        "    time.sleep(60)\n")
    proc = subprocess.Popen([sys.executable, str(script)], stdout=subprocess.PIPE, text=True)
    pids = [int(x) for x in proc.stdout.readline().split()]
    assert len(pids) == 2 and all(os.path.exists(f"/proc/{p}") for p in pids)
    proc.send_signal(signal.SIGKILL)                      # our own child, by handle
    proc.wait(timeout=10)
    deadline = time.time() + 20
    while time.time() < deadline and any(os.path.exists(f"/proc/{p}") and "zombie" not in open(f"/proc/{p}/status").read() for p in pids):
        time.sleep(0.5)
    alive = [p for p in pids if os.path.exists(f"/proc/{p}") and "zombie" not in open(f"/proc/{p}/status").read()]
    assert not alive, alive
