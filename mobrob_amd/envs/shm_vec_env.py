"""`ShmVecEnv`: any `EnvWrapper`-style environments stepped by worker PROCESSES on all host cores, with the step
results written straight into one shared, GPU-visible block.

This is what `vec_env_type: subproc` selects -- the reference's `SubprocVecEnv`
(/root/reference/src/mobrob/rl_control/ppo.py:30-33, fed by `make_vec_env(get_env, ...)` :37-48) re-thought for
thousands of environments next to a GPU learner:

  * SB3 starts ONE process per environment and moves every observation through a pipe as a pickled array.  Here W
    workers (default: one per USABLE host core: affinity mask and cgroup CPU quota respected) each own n/W environment instances, and observations, rewards, done /
    truncated flags and terminal observations never travel through a pipe: every worker writes its rows into one
    POSIX shared-memory block.  The learner process registers that block with the GPU once
    (`mobrob_ppo_host_register` -> hipHostRegister), so the rollout kernels read observations and write clipped
    actions IN PLACE over PCIe -- the memory a worker writes is the memory the GPU reads.
  * A step crosses no pipe at all: the command ("step rows [i0, i1)"), every worker's acknowledgement (truncated
    count) and the Monitor records of the episodes that ended live in a control area of the same block, and the
    processes meet on POSIX semaphores (one wake-up semaphore per worker, one completion semaphore) -- about 1 us
    per worker instead of a pickled pipe round trip.  Pipes remain for the rare calls (seed, reset, close).
  * `step_range(i0, i1, actions)` steps a row range only, so the pipelined collector (`PartPipeline`) overlaps the
    workers' simulation of one range with the policy kernel of the other.  Rows are dealt to workers in small
    blocks round-robin, so any contiguous range keeps every worker busy.
  * Workers come from a `forkserver` that never touches the GPU; they import only the light `mobrob_amd.envs`
    modules (no torch), run with one BLAS/OpenMP thread each and die with the parent.

Per-environment behaviour is identical to `HostVecEnv` (env i seeded with seed + i at its first reset, Monitor
records, `TimeLimit.truncated`, terminal observation, auto-reset): `tests/test_shm_vec_env.py` checks rollouts are
bit-identical on the same seeds.
"""
from __future__ import annotations

import ctypes
import mmap
import multiprocessing as mp
import os
import time
import uuid

import numpy as np

from .vec_env import VecEnvBase

_ALIGN = 4096


OP_STEP, OP_PIPE = 1, 2   # ctl[3]: step rows [ctl[1], ctl[2])  |  read a command tuple from the pipe


def _layout(n, d, a, n_workers):
    """name -> (offset, shape, dtype) of the arrays inside the shared block, page aligned, and the block size.
    Data rows first (what the GPU reads / writes), then the control area: `ctl` = (sequence number, i0, i1, op),
    `ack[w]` = (sequence number done, truncated rows, episodes ended) on its own cache line per worker, `eps[i]` =
    (env row, return, length, wall time) of the episode row i's owner recorded for it in the current command."""
    spec = [("obs", (n, d), np.float32), ("act", (n, a), np.float32), ("rew", (n,), np.float32),
            ("done", (n,), np.uint8), ("trunc", (n,), np.uint8), ("term", (n, d), np.float32),
            ("ctl", (8,), np.int64), ("ack", (n_workers, 8), np.int64), ("eps", (n, 4), np.float64)]
    out, off = {}, 0
    for name, shape, dt in spec:
        out[name] = (off, shape, np.dtype(dt))
        nbytes = int(np.prod(shape)) * np.dtype(dt).itemsize
        off += -(-nbytes // _ALIGN) * _ALIGN
    return out, max(off, _ALIGN)


def _views(mm, layout):
    return {k: np.frombuffer(mm, dtype=dt, count=int(np.prod(shape)), offset=off).reshape(shape)
            for k, (off, shape, dt) in layout.items()}


def usable_cores():
    """Host cores this process may actually use: the smallest of the processor count, the affinity mask and the
    cgroup CPU quota (containers often show all 256 hardware threads of the node and grant 16 of them)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]          # cgroup v2
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:                                                                      # cgroup v1
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return n


def owned_rows(worker, n_workers, n, block):
    """Rows of worker `worker`: blocks of `block` consecutive rows dealt round-robin."""
    rows = np.arange(n)
    return rows[(rows // block) % n_workers == worker]


def _space_desc(space):
    return (tuple(space.shape), np.asarray(space.low), np.asarray(space.high), np.dtype(space.dtype).str)


def _worker_main(conn, wake, finished, w, n_workers, n, rows, env_fns_blob, block_seeds):
    """Worker process `w`: builds and owns the environments of `rows`, reports their spaces, maps the shared block the
    parent then creates, and serves commands: sleeps on its `wake` semaphore, reads the command from the control
    area, writes results + acknowledgement into the block, posts `finished`."""
    for var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[var] = "1"
    import cloudpickle
    env_fns = cloudpickle.loads(env_fns_blob)
    envs = [fn() for fn in env_fns]
    seeds = list(block_seeds)
    conn.send(("ready", _space_desc(envs[0].observation_space), _space_desc(envs[0].action_space)))
    op, path, d, a = conn.recv()
    assert op == "attach"
    fd = os.open(path, os.O_RDWR)
    layout, size = _layout(n, d, a, n_workers)
    mm = mmap.mmap(fd, size)
    os.close(fd)
    v = _views(mm, layout)
    obs, act, rew, done, trunc, term = (v[k] for k in ("obs", "act", "rew", "done", "trunc", "term"))
    ctl, ack, eps = v["ctl"], v["ack"][w], v["eps"]
    conn.send(("attached",))
    ep_ret = np.zeros(len(rows), np.float64)
    ep_len = np.zeros(len(rows), np.int64)
    t0 = time.time()

    def reset_one(k):
        kw = {}
        if seeds[k] is not None:
            kw["seed"], seeds[k] = seeds[k], None
        o, _ = envs[k].reset(**kw)
        return o

    def step_rows(i0, i1):
        k0, k1 = np.searchsorted(rows, (i0, i1))   # rows is ascending: the owned rows inside [i0, i1) are contiguous
        n_trunc = n_eps = 0
        for k in range(k0, k1):
            i = rows[k]
            o, r, terminated, truncated, _info = envs[k].step(act[i])
            ep_ret[k] += r
            ep_len[k] += 1
            is_done = bool(terminated or truncated)
            is_trunc = bool(truncated and not terminated)
            if is_done:
                term[i] = o
                eps[rows[k0 + n_eps]] = (i, ep_ret[k], ep_len[k], round(time.time() - t0, 6))  # slots: own rows of the range
                n_eps += 1
                ep_ret[k], ep_len[k] = 0.0, 0
                o = reset_one(k)
                n_trunc += is_trunc
            obs[i], rew[i], done[i], trunc[i] = o, r, is_done, is_trunc
        return n_trunc, n_eps

    try:
        learner = mp.parent_process()
        while True:
            if not wake.acquire(timeout=5.0):   # nothing to do: make sure the learner still exists (it may have been killed)
                if learner is not None and not learner.is_alive():
                    break
                continue
            seq, op = int(ctl[0]), int(ctl[3])
            if op == OP_STEP:
                ack[1], ack[2] = step_rows(int(ctl[1]), int(ctl[2]))
            else:
                cmd = conn.recv()
                ack[1] = ack[2] = 0
                if cmd[0] == "reset":
                    ep_ret[:], ep_len[:] = 0.0, 0
                    for k, i in enumerate(rows):
                        obs[i] = reset_one(k)
                elif cmd[0] == "seed":  # (base seed | None): env i <- seed + i at its next reset, action space now
                    for k, i in enumerate(rows):
                        seeds[k] = None if cmd[1] is None else cmd[1] + int(i)
                        envs[k].action_space.seed(seeds[k])
                elif cmd[0] == "call":  # (method name, row, *args) -> envs[row].method(*args), answered on the pipe
                    k = int(np.searchsorted(rows, cmd[2]))
                    conn.send(getattr(envs[k], cmd[1])(*cmd[3:]))
                elif cmd[0] == "close":
                    break
            ack[0] = seq
            finished.release()
    except (EOFError, KeyboardInterrupt):
        pass
    finally:
        for e in envs:
            try:
                e.close()
            except Exception:  # noqa: BLE001
                pass
        del v, obs, act, rew, done, trunc, term, ctl, ack, eps
        try:
            mm.close()
        except BufferError:
            pass


class ShmVecEnv(VecEnvBase):
    def __init__(self, env_fns, seed=None, n_workers=None, block=None, start_method="forkserver"):
        import cloudpickle
        if mp.parent_process() is not None:
            # forkserver / spawn children re-import the launching script: an unguarded script would build its own
            # ShmVecEnv in every worker, recursively
            raise RuntimeError("ShmVecEnv created inside a multiprocessing child: protect the entry point of the "
                               "launching script with `if __name__ == '__main__':`")
        self.num_envs = n = len(env_fns)
        if n < 1:
            raise ValueError("ShmVecEnv needs at least one environment")
        self.n_workers = W = max(1, min(n, int(n_workers or os.environ.get("MOBROB_ENV_WORKERS", 0) or usable_cores())))
        self.block = int(block) if block else max(1, n // (W * 8))
        self._rows = [owned_rows(w, W, n, self.block) for w in range(W)]
        ctx = mp.get_context(start_method)
        if start_method == "forkserver":
            try:
                ctx.set_forkserver_preload(["mobrob_amd.envs.wrapper", "mobrob_amd.envs.shm_vec_env"])
            except Exception:  # noqa: BLE001 - the server may already be running
                pass
        self._conns, self._procs, self._mm, self._path = [], [], None, None
        self._wake, self._finished, self._seq = [ctx.Semaphore(0) for _ in range(W)], ctx.Semaphore(0), 0
        try:
            for w in range(W):
                parent, child = ctx.Pipe(duplex=True)
                rows = self._rows[w]
                blob = cloudpickle.dumps([env_fns[i] for i in rows])
                seeds = [None if seed is None else seed + int(i) for i in rows]
                p = ctx.Process(target=_worker_main, daemon=True,
                                args=(child, self._wake[w], self._finished, w, W, n, rows, blob, seeds))
                p.start()
                child.close()
                self._conns.append(parent)
                self._procs.append(p)
            # the environments are built inside the workers only (a real simulator never loads into the learner
            # process); worker 0 reports the spaces, then the block is sized, created and mapped by everyone
            descs = [self._recv(w) for w in range(W)]
            if any(m[0] != "ready" for m in descs):
                raise RuntimeError(f"ShmVecEnv worker failed to start: {descs!r}")
            from .spaces import Box
            (oshape, olow, ohigh, odt), (ashape, alow, ahigh, adt) = descs[0][1], descs[0][2]
            self.observation_space, self.action_space = Box(olow, ohigh, oshape, np.dtype(odt)), Box(alow, ahigh, ashape, np.dtype(adt))
            self.obs_dim, self.act_dim = int(oshape[0]), int(ashape[0])
            self._layout, self._size = _layout(n, self.obs_dim, self.act_dim, W)
            self._path = f"/dev/shm/mobrob_vecenv_{os.getpid()}_{uuid.uuid4().hex[:12]}"
            fd = os.open(self._path, os.O_CREAT | os.O_EXCL | os.O_RDWR, 0o600)
            try:
                os.ftruncate(fd, self._size)
                self._mm = mmap.mmap(fd, self._size)
            finally:
                os.close(fd)
            for c in self._conns:
                c.send(("attach", self._path, self.obs_dim, self.act_dim))
            if any(self._recv(w)[0] != "attached" for w in range(W)):
                raise RuntimeError("ShmVecEnv worker could not map the shared block")
        except BaseException:
            self.close()
            raise
        finally:
            # every worker has the block mapped (or we are failing): the NAME can go, the memory lives while mapped
            if self._path is not None:
                try:
                    os.unlink(self._path)
                except FileNotFoundError:
                    pass
        self._v = _views(self._mm, self._layout)
        self._obs, self._act, self._rew = self._v["obs"], self._v["act"], self._v["rew"]
        self._done, self._trunc, self._term = self._v["done"], self._v["trunc"], self._v["term"]
        self._ctl, self._ack, self._eps = self._v["ctl"], self._v["ack"], self._v["eps"]
        self._episodes = []          # Monitor records (dicts) not yet handed to the learner
        self._ep_count, self._ep_ret_sum, self._ep_len_sum = 0, 0.0, 0.0
        self._registered_with = None
        if seed is not None:
            self.seed(seed)

    # ---- plumbing --------------------------------------------------------------------------------------
    def _recv(self, w):
        try:
            return self._conns[w].recv()
        except (EOFError, ConnectionResetError) as ex:
            raise RuntimeError(f"ShmVecEnv worker {w} died (exit code {self._procs[w].exitcode})") from ex

    def _command(self, op, i0=0, i1=0, pipe_cmd=None):
        """Publish one command in the control area, wake every worker, wait until all have acknowledged it -> number
        of truncated rows.  Episodes that ended are appended to the Monitor records in environment order."""
        self._seq += 1
        self._ctl[:4] = (self._seq, i0, i1, op)
        if pipe_cmd is not None:
            for w, c in enumerate(self._conns):
                try:
                    c.send(pipe_cmd)
                except (BrokenPipeError, OSError) as ex:
                    raise RuntimeError(f"ShmVecEnv worker {w} died (exit code {self._procs[w].exitcode})") from ex
        for sem in self._wake:
            sem.release()
        for _ in range(self.n_workers):
            while not self._finished.acquire(timeout=1.0):
                dead = [w for w, p in enumerate(self._procs) if not p.is_alive()]
                if dead:
                    raise RuntimeError(f"ShmVecEnv worker {dead[0]} died (exit code {self._procs[dead[0]].exitcode})")
        ack = self._ack
        if op != OP_STEP:
            return 0
        if ack[:, 2].any():
            new = []
            for w in np.nonzero(ack[:, 2])[0]:
                rows = self._rows[w]
                k0 = int(np.searchsorted(rows, i0))
                for slot in rows[k0:k0 + int(ack[w, 2])]:
                    i, r, l, t = self._eps[slot]
                    new.append({"r": float(r), "l": int(l), "t": float(t), "env": int(i)})
            new.sort(key=lambda e: e["env"])
            self._episodes += new
            self._ep_count += len(new)
            self._ep_ret_sum += sum(e["r"] for e in new)
            self._ep_len_sum += sum(e["l"] for e in new)
        return int(ack[:, 1].sum())

    def shared_block(self):
        """(address, bytes) of the block in THIS process -- what the engine registers with the GPU."""
        return ctypes.addressof(ctypes.c_char.from_buffer(self._mm)), self._size

    def buffers(self):
        """The arrays the collector hands to the engine (views of the shared block)."""
        return dict(obs=self._obs, clip=self._act, rew=self._rew, done=self._done, trunc=self._trunc, term=self._term)

    # ---- VecEnv contract -------------------------------------------------------------------------------
    def seed(self, seed=None):
        self._command(OP_PIPE, pipe_cmd=("seed", seed))

    def reset(self):
        self._command(OP_PIPE, pipe_cmd=("reset",))
        return self._obs

    def step_range(self, i0, i1, actions):
        """Step rows [i0, i1) on all workers -> number of truncated rows.  `actions` is the full [n, act_dim] array;
        when it is not the shared action array itself its rows are copied in first."""
        if actions is not self._act:
            self._act[i0:i1] = actions[i0:i1]
        return self._command(OP_STEP, int(i0), int(i1))

    def step_arrays(self, actions):
        """-> (obs, rewards, dones u8, truncated u8, terminal_obs, n_truncated); shared arrays, reused between calls."""
        nt = self.step_range(0, self.num_envs, actions)
        return self._obs, self._rew, self._done, self._trunc, self._term, nt

    def step(self, actions):
        first_new = len(self._episodes)
        obs, rew, done, trunc, term, _ = self.step_arrays(np.asarray(actions, np.float32))
        infos = [{} for _ in range(self.num_envs)]
        for i in np.nonzero(done)[0]:
            infos[i] = {"TimeLimit.truncated": bool(trunc[i]), "terminal_observation": term[i].copy()}
        for ep in self._episodes[first_new:]:
            infos[ep["env"]]["episode"] = {k: ep[k] for k in ("r", "l", "t")}
        return obs.copy(), rew.copy(), done.astype(bool), infos

    def pop_episodes(self):
        """Monitor records {r, l, t} of the episodes that ended since the last call, in completion order."""
        out, self._episodes = [{k: e[k] for k in ("r", "l", "t")} for e in self._episodes], []
        return out

    def episode_stats(self, reset=True):
        n = self._ep_count
        st = {"episodes": n, "goals": None, "ep_rew_mean": self._ep_ret_sum / n if n else float("nan"),
              "ep_len_mean": self._ep_len_sum / n if n else float("nan")}
        if reset:
            self._ep_count, self._ep_ret_sum, self._ep_len_sum = 0, 0.0, 0.0
        return st

    def env_method(self, name, row, *args):
        """Call a method of one environment instance inside its worker (diagnostics, tests)."""
        w = int((row // self.block) % self.n_workers)
        self._seq += 1
        self._ctl[:4] = (self._seq, 0, 0, OP_PIPE)
        self._conns[w].send(("call", name, int(row)) + tuple(args))
        self._wake[w].release()
        out = self._recv(w)
        self._finished.acquire()
        return out

    def close(self):
        if getattr(self, "_mm", None) is not None and getattr(self, "_ctl", None) is not None and self._procs:
            self._seq += 1
            self._ctl[:4] = (self._seq, 0, 0, OP_PIPE)
        for w, c in enumerate(getattr(self, "_conns", [])):
            try:
                c.send(("close",))
                self._wake[w].release()
            except (BrokenPipeError, OSError):
                pass
        for p in getattr(self, "_procs", []):
            p.join(timeout=5)
            if p.is_alive():
                p.terminate()   # our own child, by handle
                p.join(timeout=5)
        for c in getattr(self, "_conns", []):
            c.close()
        self._conns, self._procs = [], []
        if getattr(self, "_registered_with", None) is not None:
            try:
                self._registered_with.unregister_host(self.shared_block()[0])
            except Exception:  # noqa: BLE001 - the engine may already be gone
                pass
            self._registered_with = None
        try:
            os.unlink(self._path)
        except (FileNotFoundError, AttributeError, TypeError):
            pass

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass
