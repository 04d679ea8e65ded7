"""Synchronous data-parallel PPO update across the GPUs of one node.

One process per GPU (`torch.distributed`, backend "nccl" == RCCL over xGMI; "gloo" in the CPU tests).
Rank r owns its own vectorised envs and rollout shard [T, N, .]; nothing about the rollout or GAE is
exchanged.  Per optimizer step there is exactly ONE gradient all-reduce (the flat P-float buffer, 645 KiB
for 2x256) and, once per epoch, one tiny all-reduce of the per-minibatch advantage statistics
(sum, sum of squares, count) so that advantage normalisation, the `mean()` losses and the global-norm
clip equal single-process SB3 arithmetic on the union minibatch.  Every rank then applies the identical
clip + Adam step -> replicas stay bit-identical without a parameter broadcast.

On GPUs the loop runs in C (`mobrob_ppo_train_dp`): grad kernel -> ncclAllReduce on the engine's stream -> clip +
Adam kernels, no interpreter and no host synchronisation between optimizer steps; the engine owns its RCCL
communicator (`mobrob_ppo_comm_init`; the 128-byte id travels through torch.distributed).  Under a gloo group (two
test ranks sharing one GPU) the same C loop calls back into `gloo_all_reduce`.

The exchange protocol itself is also written out below against a small backend interface (`_update_loop`), so that it
can be exercised WITHOUT a GPU by the world_size-2 gloo tests on a NumPy backend:

    backend.epoch_begin(perm_or_None)           -> local advantage partial sums ready
    backend.advstat_tensor()                    -> torch tensor [n_mb, 4] float64 (in-place all-reduce target)
    backend.minibatch_grad(mb)                  -> local gradient of the GLOBAL-mean loss
    backend.grad_tensor()                       -> torch tensor [P (+ 8 loss sums)] float32 (in-place all-reduce target)
    backend.minibatch_apply()                   -> clip_grad_norm_ + Adam; truthy = SB3's target_kl dropped the step
    backend.n_minibatches, backend.n_epochs
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.distributed as dist


def distributed_context(init=True):
    """(world_size, rank, local_rank) of this process.  An already initialised process group wins; otherwise the
    torchrun environment (WORLD_SIZE / RANK / LOCAL_RANK, rendezvous on MASTER_ADDR -- use 127.0.0.1 on one node) is
    used to initialise the "nccl" (= RCCL) group, one process per GPU."""
    import os
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(), dist.get_rank(), int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 or not init:
        return 1, 0, int(os.environ.get("LOCAL_RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    return dist.get_world_size(), dist.get_rank(), local


def agree_all(flag, group=None, device_id=0):
    """True only if `flag` is true on EVERY rank of the group (one MIN all-reduce): how the learn loop agrees to go on."""
    dev = torch.device("cuda", device_id) if dist.get_backend(group) == "nccl" else "cpu"
    t = torch.tensor([int(bool(flag))], dtype=torch.int32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(int(t.item()))


class _DevArray:
    """__cuda_array_interface__ view of an engine-owned device buffer (zero-copy into torch)."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def device_tensor(ptr: int, shape, dtype: torch.dtype, device) -> torch.Tensor:
    typestr = {torch.float32: "<f4", torch.float64: "<f8", torch.int32: "<i4"}[dtype]
    return torch.as_tensor(_DevArray(ptr, shape, typestr), device=device)


class EngineBackend:
    """Adapter: mobrob_amd.engine.PPOEngine -> the protocol above (device buffers exposed as torch tensors).

    Stream ordering: the engine's kernels and the collectives must be ordered on ONE stream.  torch's default
    stream has the handle 0, which `mobrob_ppo_set_stream` reads as "use your own stream", so the backend always
    owns a dedicated `torch.cuda.Stream`: the engine adopts it, and `train_data_parallel` issues the collectives
    with that stream current (ProcessGroupNCCL orders its RCCL launch after the current stream's work and makes
    the current stream wait for the result)."""

    def __init__(self, engine, device=None):
        self.e = engine
        self.device = torch.device("cuda", engine.cfg.device_id) if device is None else device
        torch.cuda.set_device(self.device)
        cur = torch.cuda.current_stream(self.device)
        self.stream = cur if cur.cuda_stream != 0 else torch.cuda.Stream(self.device)
        self.e.set_stream(self.stream.cuda_stream)
        gp, gb = engine.device_buffer("grad_exchange")  # [P] gradient + [8] loss sums: one message per optimizer step
        ap, ab = engine.device_buffer("advstat")
        self._grad = device_tensor(gp, (gb // 4,), torch.float32, self.device)
        self._adv = device_tensor(ap, (ab // 32, 4), torch.float64, self.device)
        self.n_minibatches = engine.n_minibatches
        self.n_epochs = int(engine.cfg.n_epochs)

    def epoch_begin(self, perm=None):
        self.e.epoch_begin(perm)

    def advstat_tensor(self):
        return self._adv

    def minibatch_grad(self, mb):
        self.e.minibatch_grad(mb)

    def grad_tensor(self):
        return self._grad

    def minibatch_apply(self):
        """-> True when SB3's target_kl check dropped the step (the loop ends train() on every rank alike)."""
        return self.e.minibatch_apply_checked()

    # ---- the C loop ------------------------------------------------------------------------------------
    def ensure_comm(self, group=None):
        """RCCL communicator of the engine over the ranks of `group` (collective, once).  Returns True when EVERY rank
        has one; False (on every rank alike) when any rank could not create it -- the caller then runs the protocol
        loop with torch.distributed's own collectives instead of failing the job."""
        state = getattr(self, "_comm_ready", None)
        if state is None:
            state = self._comm_ready = agree_and_init_comm(self.e, group, self.device)
        return state

    def ensure_oneshot(self, group=None):
        """The engine's ONE-SHOT all-reduce over peer-mapped exchange buffers (opt-in: MOBROB_ONESHOT_AR=1): every rank
        exports its buffer's IPC handle, the handles are all-gathered over `group` (any backend), every rank maps the
        others'.  True when every rank is set up; False on every rank alike otherwise (RCCL / the callback then carry
        the sums)."""
        state = getattr(self, "_oneshot_ready", None)
        if state is not None:
            return state
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        ok, handle = 1, None
        try:
            handle = self.e.oneshot_export()
        except Exception as ex:  # noqa: BLE001
            ok, self._oneshot_error = 0, ex
        handles = [None] * world
        dist.all_gather_object(handles, handle, group=group)
        if ok and all(h is not None for h in handles):
            try:
                self.e.oneshot_open(handles, rank, world)
            except Exception as ex:  # noqa: BLE001
                ok, self._oneshot_error = 0, ex
        else:
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=self.device if dist.get_backend(group) == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        self._oneshot_ready = bool(int(flag.item()))
        if not self._oneshot_ready:
            self.e.oneshot_close()
            if rank == 0:
                import warnings
                warnings.warn("one-shot all-reduce unavailable (%r): using the default exchange" % (getattr(self, "_oneshot_error", "another rank failed"),))
        return self._oneshot_ready

    def choose_exchange(self, group=None):
        """Decide ONCE, collectively, which exchange the C loop uses on this group -- and prove it on a known vector before a
        gradient depends on it (the first multi-GPU run must be safe and attributable without anybody watching):

          * the one-shot exchange over peer-mapped memory if it was asked for (MOBROB_ONESHOT_AR=1 on EVERY rank: the switch is
            agreed with a MIN all-reduce, a rank that lacks it cannot strand the others) and its self-check -- a deterministic
            vector per rank through the exchange, compared with the rank-ordered sum -- is bit-equal on every rank;
          * else the engine's RCCL communicator, self-checked the same way ("nccl" groups);
          * else torch.distributed's own collectives from the Python loop ("nccl" groups whose engine communicator failed),
            or the host-staged callback (CPU groups: the two-ranks-on-one-GPU tests).

        -> the name of the exchange; `self.exchange_selfcheck` holds what was measured: {"oneshot": "ok" | "<n> mismatches" |
        "unavailable: ...", "rccl": ...} -- bench.py writes both into its line."""
        if getattr(self, "exchange", None) is not None:
            return self.exchange
        import os
        self.exchange_selfcheck = {}
        self._group = group                  # close() meets the SAME ranks at its barrier (a sub-group stays a sub-group)
        backend = dist.get_backend(group)
        dev = self.device if backend == "nccl" else "cpu"

        def agree(flag):
            t = torch.tensor([int(bool(flag))], dtype=torch.int32, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
            return bool(int(t.item()))

        def checked(which):
            """run the self-check on every rank; ok only if EVERY rank saw zero mismatches"""
            try:
                bad = self.e.exchange_selfcheck(which)
                note = "ok" if bad == 0 else f"{bad} elements of the four check messages differ from the rank-ordered sum"
            except Exception as ex:  # noqa: BLE001 - e.g. a peer that never published (bounded wait inside the kernel)
                bad, note = -1, f"failed: {ex}"
            ok = agree(bad == 0)
            self.exchange_selfcheck[which] = note if ok or bad != 0 else "ok here, failed on another rank"
            return ok

        world = dist.get_world_size(group)
        want_oneshot = agree(os.environ.get("MOBROB_ONESHOT_AR", "0") == "1") and world <= 8
        if want_oneshot:
            if self.ensure_oneshot(group):
                if checked("oneshot"):
                    self.exchange = "oneshot"
                    return self.exchange
                dist.barrier(group=group)          # no rank frees its exchange buffer while a peer may still read it
                self.e.oneshot_close()
            else:
                self.exchange_selfcheck["oneshot"] = "unavailable: %r" % (getattr(self, "_oneshot_error", "another rank failed"),)
        if backend == "nccl":
            if self.ensure_comm(group):
                if checked("rccl"):
                    self.exchange = "rccl"
                    return self.exchange
                import warnings
                warnings.warn("the engine's RCCL communicator failed its self-check (%s): torch.distributed collectives from the "
                              "Python loop instead" % self.exchange_selfcheck.get("rccl"))
            else:
                self.exchange_selfcheck["rccl"] = "unavailable (engine-owned communicator could not be created)"
            self.exchange = "torch.distributed"
        else:
            self.exchange = "gloo-callback"
        return self.exchange

    def close(self, group=None):
        """Closing handshake of the one-shot exchange (collective; a no-op for the other exchanges): a rank that has finished
        its last all-reduce has read every peer's slot, but a peer may still be reading ITS slot -- so every rank drains its
        stream, all meet at a barrier, and only then are the exchange buffers unmapped and freed."""
        if group is None:
            group = getattr(self, "_group", None)   # the group the exchange was chosen on
        if getattr(self, "exchange", None) == "oneshot" and getattr(self, "_oneshot_ready", False):
            self.e.synchronize()
            dist.barrier(group=group)
            self.e.oneshot_close()
            self._oneshot_ready = None      # a later update sets the exchange up again
            self.exchange = None

    def gloo_all_reduce(self, group=None):
        """All-reduce callback for `engine.train_dp` under a CPU process group: stage through the host."""
        def reduce_in_place(ptr, count, dtype, _stream):
            t = device_tensor(ptr, (count,), torch.float64 if dtype == 1 else torch.float32, self.device)
            with torch.cuda.stream(self.stream):
                host = t.cpu()                       # ordered after the engine's kernels on the shared stream
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
                t.copy_(host)
            self.stream.synchronize()
        return reduce_in_place


def agree_and_init_comm(engine, group=None, device=None):
    """The engine-owned RCCL communicator, set up so that a rank which CANNOT take part never leaves the others
    blocked inside ncclCommInitRank:

      1. every rank runs the local, non-collective half (`engine.comm_prepare`: RCCL loadable, device selectable, no
         communicator yet; rank 0 of the group also draws the unique id) and the group size is checked against the
         engine's minibatch split;
      2. ONE all-reduce(MIN) of the success flags -- any failure sends every rank to the fallback before a single
         rank has entered the blocking collective;
      3. the id is broadcast and every rank calls ncclCommInitRank with ITS RANK IN `group` (a sub-group of the job
         gets a communicator of its own size, not of the engine config's);
      4. a second MIN agrees on the outcome of the collective itself.

    Returns True when every rank holds a communicator, False (on every rank alike) otherwise."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    ok, uid, err = 1, None, None
    try:
        if world != int(engine.cfg.world_size):
            raise ValueError(f"process group of {world} ranks, but the engine splits its minibatch for world_size "
                             f"{int(engine.cfg.world_size)}")
        engine.comm_prepare()
        if rank == 0:
            uid = engine.comm_unique_id()
    except Exception as ex:  # noqa: BLE001 - e.g. librccl not loadable from the engine, an engine that already has one
        ok, err = 0, ex

    def agree(flag):
        t = torch.tensor([flag], dtype=torch.int32, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
        return int(t.item())

    if agree(ok):
        box = [uid]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        try:
            engine.comm_init(box[0], rank=rank, nranks=world)
        except Exception as ex:  # noqa: BLE001
            ok, err = 0, ex
        ready = bool(agree(ok))
    else:
        ready = False
    if not ready and rank == 0:
        import warnings
        warnings.warn("engine-owned RCCL communicator unavailable (%r): falling back to torch.distributed all-reduces "
                      "issued from Python, one per optimizer step" % (err if err is not None else "another rank failed",))
    return ready


def train_data_parallel(backend, perms=None, group=None, force_collectives=False, python_loop=False):
    """PPO.train() across ranks.  perms: per-epoch LOCAL permutations ([n_epochs, T*N_local]) or None.
    force_collectives issues the all-reduces even at world size 1 (plumbing self-test).  An `EngineBackend` runs the
    loop in C (RCCL under an "nccl" group, the gloo callback otherwise); other backends, or python_loop=True, run the
    protocol loop below with torch.distributed collectives.
    -> (epochs started, stopped early by target_kl, optimizer steps applied), identical on every rank."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    comm = world > 1 or (force_collectives and dist.is_initialized())
    stream = getattr(backend, "stream", None)
    if isinstance(backend, EngineBackend) and comm and not python_loop:
        exchange = backend.choose_exchange(group)      # decided and self-checked once per backend (collective)
        if exchange in ("oneshot", "rccl"):
            backend.e.train_dp(perms)        # the C loop: sums through peer-mapped memory (csrc/oneshot_allreduce.h) or RCCL
            return backend.e.last_train_info()
        if exchange == "torch.distributed":
            with torch.cuda.stream(stream):
                return _update_loop(backend, perms, group, comm)
        backend.e.train_dp(perms, allreduce=backend.gloo_all_reduce(group))
        return backend.e.last_train_info()
    if stream is not None:  # GPU backend, Python loop: collectives are issued with the engine's stream current
        with torch.cuda.stream(stream):
            return _update_loop(backend, perms, group, comm)
    return _update_loop(backend, perms, group, comm)


def _update_loop(backend, perms, group, comm):
    """-> (epochs started, stopped early by target_kl, optimizer steps applied)."""
    applied = 0
    for ep in range(backend.n_epochs):
        backend.epoch_begin(None if perms is None else perms[ep])
        if comm:
            dist.all_reduce(backend.advstat_tensor(), op=dist.ReduceOp.SUM, group=group)
        for mb in range(backend.n_minibatches):
            backend.minibatch_grad(mb)
            if comm:  # [P] gradient + [8] loss sums: the statistics and the target_kl decision are global
                dist.all_reduce(backend.grad_tensor(), op=dist.ReduceOp.SUM, group=group)
            if backend.minibatch_apply():  # SB3's target_kl: this step and the rest of train() are dropped
                return ep + 1, True, applied
            applied += 1
    return backend.n_epochs, False, applied
