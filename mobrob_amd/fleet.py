"""Mixed fleet: several robot types trained side by side on one GPU (BASELINE config 5).

The reference trains one robot per process -- `examples/train.py:42-46` builds one `PPOCtrl` for one `env_name`
-- so a "fleet" there is N independent processes.  Here the robot types share one process and one device:

* every segment (robot type) is its own PPO learner with its own networks, because observation / action
  widths differ (car 26/2, drone 12/18, turtlebot3 43/2, `wrapper.py` observation spaces);
* all segments' device buffers are carved out of ONE arena allocation (`mobrob_ppo_device_bytes` +
  `mobrob_ppo_create_in_arena`): the ragged rollout segments sit back to back, each addressed with its own
  (obs_dim, act_dim) strides -- `layout()` reports the packing;
* each engine owns a HIP stream; rollouts (`collect_synthetic`, one hipGraph per segment) and updates
  (`train_enqueue`) of different segments are enqueued without waiting, so the small 2x64 networks of one
  robot fill the CUs another leaves idle;
* data parallel: the fleet is sharded like a single learner -- every rank holds all segments with
  n_envs / world environments each, and each segment runs the all-reduce loop of `parallel.py` on its own
  stream (three disjoint parameter sets -> three independent small all-reduces per optimizer step).

Arithmetic per segment is exactly that of a stand-alone engine with the same arguments (tests/test_fleet_gpu.py
checks bit equality), hence exactly the oracle's.
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict

from . import _lib
from .engine import PPOEngine

# (obs_dim, act_dim) of the reference robots: observation_space / action_space of the wrappers
# (/root/reference/src/mobrob/envs/wrapper.py -- PointEnv, CarEnv, DoggoEnv, DroneEnv, Turtlebot3Env) as recorded in
# the reference checkpoints' `data` JSON (tests/golden/*.npz carry the same shapes).
ROBOT_DIMS = OrderedDict(point=(14, 2), car=(26, 2), doggo=(58, 12), drone=(12, 18), turtlebot3=(43, 2))
# mean episode lengths of the reference checkpoints -> per-step termination probability of the synthetic source
ROBOT_P_TERM = dict(point=1 / 119, car=1 / 91, doggo=1 / 107, drone=1 / 568, turtlebot3=1 / 131)

_ALIGN = 256


class FleetSegment:
    def __init__(self, name, obs_dim, act_dim, n_envs, offset, nbytes, engine):
        self.name, self.obs_dim, self.act_dim, self.n_envs = name, obs_dim, act_dim, n_envs
        self.offset, self.nbytes, self.engine = offset, nbytes, engine

    def __repr__(self):
        return (f"FleetSegment({self.name}: obs {self.obs_dim}, act {self.act_dim}, envs {self.n_envs}, "
                f"arena [{self.offset}, {self.offset + self.nbytes}))")


class MixedFleet:
    """segments: iterable of robot names (dims from ROBOT_DIMS) or dicts
    {"name", "obs_dim", "act_dim", "n_envs", **per-segment PPO overrides}; `ppo_kwargs` are the shared
    PPOEngine arguments (n_steps, batch_size, n_epochs, pi, vf, gamma, ...)."""

    def __init__(self, segments, n_envs=1024, device_id=0, rank=0, world_size=1, seed=0, **ppo_kwargs):
        self.lib = _lib.load()
        self.device_id = int(device_id)
        specs = []
        for i, s in enumerate(segments):
            if isinstance(s, str):
                if s not in ROBOT_DIMS:
                    raise ValueError(f"Env {s} not found")  # same message class as get_env (wrapper.py:566)
                s = {"name": s}
            s = dict(s)
            name = s.pop("name")
            d, a = ROBOT_DIMS.get(name, (None, None))
            kw = dict(ppo_kwargs)
            kw.update(obs_dim=s.pop("obs_dim", d), act_dim=s.pop("act_dim", a), n_envs=s.pop("n_envs", n_envs),
                      device_id=device_id, rank=rank, world_size=world_size, seed=s.pop("seed", seed + i))
            kw.update(s)
            if kw["obs_dim"] is None or kw["act_dim"] is None:
                raise ValueError(f"Env {name} not found")
            specs.append((name, kw))
        if not specs:
            raise ValueError("a fleet needs at least one segment")
        sizes = [(PPOEngine.device_bytes(**kw) + _ALIGN - 1) // _ALIGN * _ALIGN for _, kw in specs]
        self.arena_bytes = sum(sizes)
        self._arena = self.lib.mobrob_ppo_device_alloc(self.device_id, C.c_size_t(self.arena_bytes))
        if not self._arena:
            raise _lib.EngineError(self.lib.mobrob_ppo_last_error().decode())
        self.segments: list[FleetSegment] = []
        off = 0
        try:
            for (name, kw), nb in zip(specs, sizes):
                eng = PPOEngine(arena=(self._arena + off, nb), **kw)
                # the segments update CONCURRENTLY on per-segment streams: co-operative epoch launches (spinning workgroups that must all
                # be resident) of three engines at once are not what one device's residency check covers -> three launches per step
                eng.set_hyper(epoch_kernel=0)
                self.segments.append(FleetSegment(name, kw["obs_dim"], kw["act_dim"], kw["n_envs"], off, nb, eng))
                off += nb
        except Exception:
            self.close()
            raise

    # ---- structure ------------------------------------------------------------------------------------
    def __len__(self):
        return len(self.segments)

    def __getitem__(self, key):
        if isinstance(key, str):
            for s in self.segments:
                if s.name == key:
                    return s
            raise KeyError(key)
        return self.segments[key]

    @property
    def engines(self):
        return [s.engine for s in self.segments]

    def layout(self):
        """Packing of the one device arena: per segment (name, byte offset, bytes, obs_dim, act_dim, n_envs)."""
        return [(s.name, s.offset, s.nbytes, s.obs_dim, s.act_dim, s.n_envs) for s in self.segments]

    @property
    def env_steps_per_iteration(self):
        return sum(s.engine.N * s.engine.T for s in self.segments)

    # ---- one PPO iteration over the whole fleet ---------------------------------------------------------
    def collect_synthetic(self, time_limit=1000, p_term=None):
        """Enqueue every segment's rollout + GAE (device env source) on its own stream; returns immediately."""
        for s in self.segments:
            p = ROBOT_P_TERM.get(s.name, 0.01) if p_term is None else (p_term[s.name] if isinstance(p_term, dict) else p_term)
            s.engine.collect_synthetic(p_term=p, time_limit=time_limit)

    def train_enqueue(self):
        """Enqueue every segment's PPO.train(); the segments' kernels overlap on the device."""
        for s in self.segments:
            s.engine.train_enqueue()

    def synchronize(self):
        for s in self.segments:
            s.engine.synchronize()

    def train(self):
        """PPO.train() for every segment; returns {segment name: last-epoch statistics}."""
        self.train_enqueue()
        return OrderedDict((s.name, s.engine.train_stats()) for s in self.segments)

    def iteration(self, time_limit=1000, p_term=None):
        self.collect_synthetic(time_limit=time_limit, p_term=p_term)
        self.train_enqueue()

    def close(self):
        for s in getattr(self, "segments", []):
            s.engine.close()
        self.segments = []
        if getattr(self, "_arena", None):
            self.lib.mobrob_ppo_device_free(C.c_void_p(self._arena))
            self._arena = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def train_fleet_data_parallel(backends, streams, group=None, force_collectives=False, perms=None):
    """One data-parallel update of every segment.  backends[i] = parallel.EngineBackend of segment i created
    under torch stream streams[i]; the segments' loops are interleaved minibatch by minibatch so that their
    kernels and their (disjoint) gradient all-reduces overlap.  perms[i]: per-epoch LOCAL permutations of segment
    i or None (device-drawn); streams[i] None -> current stream (CPU tests)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    comm = world > 1 or (force_collectives and dist.is_initialized())
    n_epochs = max(b.n_epochs for b in backends)
    for ep in range(n_epochs):
        live = [(b, st, i) for i, (b, st) in enumerate(zip(backends, streams)) if ep < b.n_epochs]
        for b, st, i in live:
            with torch.cuda.stream(st):
                b.epoch_begin(None if perms is None or perms[i] is None else perms[i][ep])
                if comm:
                    dist.all_reduce(b.advstat_tensor(), op=dist.ReduceOp.SUM, group=group)
        for mb in range(max(b.n_minibatches for b, _, _ in live)):
            for b, st, _ in live:
                if mb >= b.n_minibatches:
                    continue
                with torch.cuda.stream(st):
                    b.minibatch_grad(mb)
                    if comm:
                        dist.all_reduce(b.grad_tensor(), op=dist.ReduceOp.SUM, group=group)
                    if b.minibatch_apply():
                        raise NotImplementedError("target_kl is not supported by the interleaved fleet update")
