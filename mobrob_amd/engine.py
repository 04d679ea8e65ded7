"""NumPy-facing wrapper of the C ABI: one `PPOEngine` per GPU (one per process in data-parallel runs)."""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict

import numpy as np

from . import _lib
from ._lib import BUF, Config, DroneParams, EpisodeStats, GoalEnv, TrainStats, check

F32 = np.float32
STAT_KEYS = ("policy_loss", "value_loss", "entropy_loss", "loss", "approx_kl", "clip_fraction", "grad_norm")


def _fp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_float))


def _u8(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_uint8))


def _f32c(a, shape=None):
    a = np.ascontiguousarray(a, dtype=F32)
    if shape is not None and a.shape != tuple(shape):
        raise ValueError(f"expected shape {tuple(shape)}, got {a.shape}")
    return a


MAX_HIDDEN = 8   # kMaxHidden (csrc/device_utils.h): hidden layers per network

# policy_kwargs `activation_fn`: lower-cased torch.nn class name -> (MOBROB_ACT_* code, torch.nn class name).  The parameter-free
# element-wise modules with torch's default arguments; tanh is SB3's default for MlpPolicy (every reference YAML).
ACTIVATIONS = OrderedDict([("tanh", (0, "Tanh")), ("relu", (1, "ReLU")), ("elu", (2, "ELU")), ("leakyrelu", (3, "LeakyReLU")),
                           ("sigmoid", (4, "Sigmoid")), ("softplus", (5, "Softplus")), ("softsign", (6, "Softsign")),
                           ("hardtanh", (7, "Hardtanh")), ("relu6", (8, "ReLU6")), ("silu", (9, "SiLU")), ("gelu", (10, "GELU")),
                           ("mish", (11, "Mish"))])


def activation_name(act) -> str:
    """`activation_fn` as a torch.nn class, its name or None (SB3's default, Tanh) -> key of ACTIVATIONS; anything else raises
    NotImplementedError by name."""
    if act is None:
        return "tanh"
    name = getattr(act, "__name__", None) or str(act)
    if name.startswith("<class "):   # the readable side of a checkpoint blob: "<class 'torch.nn.modules.activation.ELU'>"
        name = name.split("'")[1]
    name = name.rsplit(".", 1)[-1].replace("_", "").lower()
    if name not in ACTIVATIONS:
        raise NotImplementedError(f"activation_fn {act!r}: implemented are {', '.join(v[1] for v in ACTIVATIONS.values())} "
                                  "(parameter-free element-wise modules, torch's default arguments)")
    return name


def param_shapes(obs_dim, act_dim, pi, vf, use_sde=False, full_std=True):
    """SB3 `policy.state_dict()` key order and shapes (include/mobrob_ppo.h 'Conventions'); one to eight hidden layers per
    network (`nn.Sequential` indices 0, 2, 4 ...: every Linear is followed by its activation module)."""
    s = OrderedDict()
    # gSDE: one row per unit of the policy's last hidden layer (one column without full_std)
    s["log_std"] = (pi[-1], act_dim if full_std else 1) if use_sde else (act_dim,)
    for net, widths in (("policy_net", pi), ("value_net", vf)):
        prev = obs_dim
        for i, w in enumerate(widths):
            s[f"mlp_extractor.{net}.{2 * i}.weight"] = (w, prev)
            s[f"mlp_extractor.{net}.{2 * i}.bias"] = (w,)
            prev = w
    s["action_net.weight"] = (act_dim, pi[-1])
    s["action_net.bias"] = (act_dim,)
    s["value_net.weight"] = (1, vf[-1])
    s["value_net.bias"] = (1,)
    return s


class _PinnedBlock:
    """Owner of one hipHostMalloc allocation.  NumPy arrays made from it keep it alive through `.base`, and the memory
    is returned to the driver when the LAST such array dies -- not when the engine closes: the rollout collector, the
    environment (`use_buffers`) and `PPO._last_obs` all hold views that outlive `PPOEngine.close()`."""

    def __init__(self, lib, nbytes, shape, dtype):
        self._lib, self.ptr = lib, lib.mobrob_ppo_host_alloc(max(int(nbytes), 1))
        if not self.ptr:
            raise MemoryError("hipHostMalloc failed")
        C.memset(self.ptr, 0, max(int(nbytes), 1))
        self.__array_interface__ = {"shape": tuple(shape), "typestr": np.dtype(dtype).str, "data": (self.ptr, False),
                                    "version": 3}

    def __del__(self):
        ptr, self.ptr = getattr(self, "ptr", None), None
        if ptr:
            try:
                self._lib.mobrob_ppo_host_free(C.c_void_p(ptr))
            except Exception:  # noqa: BLE001 - interpreter shutdown
                pass


class PartPipeline:
    """The N envs cut into `nparts` contiguous row ranges: while the host simulator steps range p, the GPU runs the
    policy for the other ranges.  Results equal act()/store() over all rows (same noise per env and step).  The
    ctypes pointers of the pinned arrays are taken once: the loop below is the per-step hot path of a host-env rollout.

        pipe = engine.part_pipeline(2, obs, clip, rew, done, trunc, term); engine.rollout_begin()
        for p in range(2): pipe.act(p)
        for t in range(T):
            for p in range(2):
                pipe.wait(p); n_trunc = env.step_range(*pipe.bounds[p], clip); pipe.store(p, n_trunc > 0)
                if t + 1 < T: pipe.act(p)
        engine.finish_rollout(obs, done)

    or, for an environment stepped by a C function, the same loop without Python: pipe.collect(fn, handle).
    """

    def __init__(self, engine, nparts, obs, clipped, rewards, dones, truncated, terminal_obs):
        e, n = engine, int(nparts)
        want = dict(obs=((e.N, e.D), F32), clipped=((e.N, e.A), F32), rewards=((e.N,), F32), dones=((e.N,), np.uint8),
                    truncated=((e.N,), np.uint8), terminal_obs=((e.N, e.D), F32))
        for name, arr in dict(obs=obs, clipped=clipped, rewards=rewards, dones=dones, truncated=truncated,
                              terminal_obs=terminal_obs).items():
            shape, dt = want[name]
            if arr.shape != shape or arr.dtype != dt or not arr.flags.c_contiguous:
                raise ValueError(f"{name} must be a C-contiguous {np.dtype(dt).name}{shape} array from engine.pinned()")
        self._keep = (obs, clipped, rewards, dones, truncated, terminal_obs)
        self.nparts, self._h, self._lib = n, engine._h, engine.lib
        self.bounds = [(e.N * p // n, e.N * (p + 1) // n) for p in range(n)]
        self._obs, self._clip, self._rew = _fp(obs), _fp(clipped), _fp(rewards)
        self._done, self._trunc, self._term = _u8(dones), _u8(truncated), _fp(terminal_obs)

    def act(self, part):
        check(self._lib.mobrob_ppo_act_part(self._h, part, self.nparts, self._obs, self._clip))

    def wait(self, part):
        check(self._lib.mobrob_ppo_wait_part(self._h, part))

    def store(self, part, any_truncated=True, pull_next_obs=True):
        """pull_next_obs: the same launch also moves the part's next observations (already in `obs`: the env wrote
        them) into the next rollout slot, so the following act() launches the policy kernel only."""
        check(self._lib.mobrob_ppo_store_part(self._h, part, self.nparts, self._rew, self._done,
                                              self._trunc if any_truncated else None,
                                              self._term if any_truncated else None,
                                              self._obs if pull_next_obs else None))

    def collect(self, step_range_fn, env_handle):
        """The whole rollout (rollout_begin ... finish_rollout) in one native call: mobrob_ppo_collect_host drives the
        environment through `step_range_fn` (address of a C function with mobrob_env_step_range_fn's signature, e.g.
        NativeGoalVecEnv.step_range_fn) -- no Python frame per step."""
        check(self._lib.mobrob_ppo_collect_host(self._h, step_range_fn, env_handle, self.nparts, self._obs, self._clip,
                                                self._rew, self._done, self._trunc, self._term))


class PPOEngine:
    @staticmethod
    def make_config(obs_dim, act_dim, n_envs, n_steps, batch_size=64, n_epochs=10, pi=(64, 64), vf=(64, 64),
                    gamma=0.99, gae_lambda=0.95, clip_range=0.2, ent_coef=0.0, vf_coef=0.5, max_grad_norm=0.5,
                    learning_rate=3e-4, adam_betas=(0.9, 0.999), adam_eps=1e-5, normalize_advantage=True,
                    action_low=-1.0, action_high=1.0, seed=0, device_id=0, rank=0, world_size=1, fast_kernels=True,
                    rollout_graph=True, rollout_persistent=True, activation="tanh", forward_x3=True, use_sde=False,
                    sde_sample_freq=-1, sde_full_std=True, sde_use_expln=False) -> Config:
        """PPO(...) keyword arguments -> `mobrob_ppo_config_t` (SB3 defaults, Appendix A.1)."""
        if not (1 <= len(pi) <= MAX_HIDDEN and 1 <= len(vf) <= MAX_HIDDEN):
            raise ValueError(f"net_arch: one to {MAX_HIDDEN} hidden layers per network (pi=[h1, ...], vf=[h1, ...])")
        cfg = Config()
        _lib.load().mobrob_ppo_default_config(C.byref(cfg))
        cfg.obs_dim, cfg.act_dim = int(obs_dim), int(act_dim)
        pw, vw = (list(map(int, pi)) + [0] * MAX_HIDDEN)[:MAX_HIDDEN], (list(map(int, vf)) + [0] * MAX_HIDDEN)[:MAX_HIDDEN]   # 0 ends the list
        cfg.pi_hidden[0], cfg.pi_hidden[1], cfg.pi_hidden3 = pw[:3]
        cfg.vf_hidden[0], cfg.vf_hidden[1], cfg.vf_hidden3 = vw[:3]
        for i in range(3, MAX_HIDDEN):
            cfg.pi_hidden_ext[i - 3], cfg.vf_hidden_ext[i - 3] = pw[i], vw[i]
        cfg.n_envs, cfg.n_steps, cfg.batch_size, cfg.n_epochs = int(n_envs), int(n_steps), int(batch_size), int(n_epochs)
        cfg.gamma, cfg.gae_lambda, cfg.clip_range = float(gamma), float(gae_lambda), float(clip_range)
        cfg.ent_coef, cfg.vf_coef, cfg.max_grad_norm = float(ent_coef), float(vf_coef), float(max_grad_norm)
        cfg.learning_rate = float(learning_rate)
        cfg.adam_beta1, cfg.adam_beta2, cfg.adam_eps = float(adam_betas[0]), float(adam_betas[1]), float(adam_eps)
        cfg.action_low, cfg.action_high = float(action_low), float(action_high)
        cfg.normalize_advantage = int(bool(normalize_advantage))
        cfg.seed, cfg.device_id, cfg.rank, cfg.world_size = int(seed), int(device_id), int(rank), int(world_size)
        cfg.fast_kernels = int(bool(fast_kernels))
        cfg.rollout_graph = int(bool(rollout_graph))
        cfg.rollout_persistent = int(bool(rollout_persistent))
        cfg.activation = ACTIVATIONS[activation_name(activation)][0]
        cfg.forward_x3 = int(bool(forward_x3))
        cfg.use_sde, cfg.sde_sample_freq = int(bool(use_sde)), int(sde_sample_freq)
        cfg.sde_full_std, cfg.sde_use_expln = int(bool(sde_full_std)), int(bool(sde_use_expln))
        return cfg

    @staticmethod
    def device_bytes(**kwargs) -> int:
        """Device bytes an engine with these arguments occupies (host-only sizing pass, no GPU needed)."""
        cfg = PPOEngine.make_config(**kwargs)
        n = C.c_size_t(0)
        check(_lib.load().mobrob_ppo_device_bytes(C.byref(cfg), C.byref(n)))
        return int(n.value)

    def __init__(self, obs_dim, act_dim, n_envs, n_steps, *, arena=None, **kwargs):
        """arena: optional (device_pointer, bytes) -- build the engine inside caller-owned device memory
        (mobrob_ppo_create_in_arena; the fleet packs several engines into one allocation this way)."""
        self.lib = _lib.load()
        cfg = self.make_config(obs_dim, act_dim, n_envs, n_steps, **kwargs)
        pi = tuple(w for w in (cfg.pi_hidden[0], cfg.pi_hidden[1], cfg.pi_hidden3, *cfg.pi_hidden_ext) if w > 0)
        vf = tuple(w for w in (cfg.vf_hidden[0], cfg.vf_hidden[1], cfg.vf_hidden3, *cfg.vf_hidden_ext) if w > 0)
        self.cfg = cfg
        self.D, self.A, self.N, self.T = int(obs_dim), int(act_dim), int(n_envs), int(n_steps)
        self.shapes = param_shapes(self.D, self.A, pi, vf, bool(cfg.use_sde), bool(cfg.sde_full_std))
        self.use_sde, self.HL = bool(cfg.use_sde), int(pi[-1])
        self._h = C.c_void_p()
        if arena is None:
            check(self.lib.mobrob_ppo_create(C.byref(cfg), C.byref(self._h)))
        else:
            check(self.lib.mobrob_ppo_create_in_arena(C.byref(cfg), C.c_void_p(int(arena[0])), C.c_size_t(int(arena[1])),
                                                      C.byref(self._h)))
        self.P = int(self.lib.mobrob_ppo_param_count(self._h))
        assert self.P == sum(int(np.prod(s)) for s in self.shapes.values())
        self.n_minibatches = int(self.lib.mobrob_ppo_num_minibatches(self._h))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.mobrob_ppo_destroy(self._h)
            self._h = C.c_void_p()
            for addr in list(getattr(self, "_registered", [])):
                self.unregister_host(addr)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- parameters ---------------------------------------------------------------------------
    def get_flat_params(self):
        out = np.empty(self.P, F32)
        check(self.lib.mobrob_ppo_get_params(self._h, _fp(out), self.P))
        return out

    def set_flat_params(self, flat):
        flat = _f32c(flat, (self.P,))
        check(self.lib.mobrob_ppo_set_params(self._h, _fp(flat), self.P))

    def unflatten(self, flat):
        out, o = OrderedDict(), 0
        for k, s in self.shapes.items():
            n = int(np.prod(s))
            out[k] = np.array(flat[o:o + n], F32).reshape(s)
            o += n
        return out

    def flatten(self, d):
        parts = []
        for k, s in self.shapes.items():
            a = np.asarray(d[k], F32)
            if a.shape != tuple(s):
                raise ValueError(f"size mismatch for {k}: expected {tuple(s)}, got {a.shape}")
            parts.append(a.ravel())
        return np.concatenate(parts)

    def get_params(self):
        return self.unflatten(self.get_flat_params())

    def set_params(self, d):
        self.set_flat_params(self.flatten(d))

    def get_optimizer_state(self):
        m, v, step = np.empty(self.P, F32), np.empty(self.P, F32), C.c_int64()
        check(self.lib.mobrob_ppo_get_optimizer_state(self._h, _fp(m), _fp(v), self.P, C.byref(step)))
        return self.unflatten(m), self.unflatten(v), int(step.value)

    def set_optimizer_state(self, exp_avg, exp_avg_sq, step):
        m, v = _f32c(self.flatten(exp_avg)), _f32c(self.flatten(exp_avg_sq))
        check(self.lib.mobrob_ppo_set_optimizer_state(self._h, _fp(m), _fp(v), self.P, int(step)))

    # ---- rollout ------------------------------------------------------------------------------
    # ---- gSDE ---------------------------------------------------------------------------------
    def sde_reset_noise(self):
        """policy.reset_noise(n_envs): new exploration matrices from the current log_std."""
        check(self.lib.mobrob_ppo_sde_reset_noise(self._h))

    def sde_set_noise(self, z):
        """Exploration matrices as an input: z ~ N(0, 1) of shape [n_envs, HL, A] (None: the engine's own draws again)."""
        if z is None:
            check(self.lib.mobrob_ppo_sde_set_noise(self._h, None))
            return
        z = _f32c(z, (self.N, self.HL, self.A))
        check(self.lib.mobrob_ppo_sde_set_noise(self._h, _fp(z)))

    def rollout_begin(self):
        check(self.lib.mobrob_ppo_rollout_begin(self._h))

    def act(self, obs, eps=None, out_clipped=None, want_all=True):
        """-> (raw actions, clipped actions, values, log_probs).  With want_all=False only the clipped actions the
        env needs are copied back (raw actions / values / log-probs stay in the device rollout buffer);
        out_clipped may be a pinned array (engine.pinned) to skip the staging copy."""
        obs = _f32c(obs, (self.N, self.D))
        eps = None if eps is None else _f32c(eps, (self.N, self.A))
        a_clip = np.empty((self.N, self.A), F32) if out_clipped is None else out_clipped
        if not want_all:
            check(self.lib.mobrob_ppo_act(self._h, _fp(obs), _fp(eps), None, _fp(a_clip), None, None))
            return None, a_clip, None, None
        a_raw = np.empty((self.N, self.A), F32)
        val, lp = np.empty(self.N, F32), np.empty(self.N, F32)
        check(self.lib.mobrob_ppo_act(self._h, _fp(obs), _fp(eps), _fp(a_raw), _fp(a_clip), _fp(val), _fp(lp)))
        return a_raw, a_clip, val, lp

    def store(self, rewards, dones, truncated=None, terminal_obs=None):
        rewards = _f32c(rewards, (self.N,))
        dones = np.ascontiguousarray(dones, dtype=np.uint8)
        tr = None if truncated is None else np.ascontiguousarray(truncated, dtype=np.uint8)
        to = None if terminal_obs is None else _f32c(terminal_obs, (self.N, self.D))
        check(self.lib.mobrob_ppo_store(self._h, _fp(rewards), _u8(dones), _u8(tr), _fp(to)))

    def part_pipeline(self, nparts, obs, clipped, rewards, dones, truncated, terminal_obs):
        """Driver of the pipelined host-env rollout (mobrob_ppo_act_part / wait_part / store_part) over FULL [N][...]
        pinned arrays (`engine.pinned`); see PartPipeline."""
        return PartPipeline(self, nparts, obs, clipped, rewards, dones, truncated, terminal_obs)

    def finish_rollout(self, last_obs, dones):
        last_obs = _f32c(last_obs, (self.N, self.D))
        dones = np.ascontiguousarray(dones, dtype=np.uint8)
        check(self.lib.mobrob_ppo_finish_rollout(self._h, _fp(last_obs), _u8(dones)))

    def collect_synthetic(self, p_term=1.0 / 107.0, time_limit=1000):
        check(self.lib.mobrob_ppo_collect_synthetic(self._h, float(p_term), int(time_limit)))

    def collect_goal_env(self, pos_dim, mix, time_limit, terminate_on_goal=True, dt=0.05, extent=3.0, reach_radius=0.3,
                         goal_bonus=5.0, extra_bonus=0.0, obs_noise=0.1):
        """Whole rollout + GAE against the device-resident goal environment (mobrob_ppo_collect_goal_env).
        mix: [pos_dim, act_dim] action -> velocity command read-out."""
        g = GoalEnv()
        g.pos_dim, g.terminate_on_goal, g.time_limit = int(pos_dim), int(bool(terminate_on_goal)), int(time_limit)
        g.dt, g.extent, g.reach_radius = float(dt), float(extent), float(reach_radius)
        g.goal_bonus, g.extra_bonus, g.obs_noise = float(goal_bonus), float(extra_bonus), float(obs_noise)
        mix = np.asarray(mix, F32)
        if mix.shape != (int(pos_dim), self.A):
            raise ValueError(f"mix must be [{int(pos_dim)}, {self.A}], got {mix.shape}")
        for j in range(mix.shape[0]):
            for k in range(mix.shape[1]):
                g.mix[j][k] = float(mix[j, k])
        check(self.lib.mobrob_ppo_collect_goal_env(self._h, C.byref(g)))

    def episode_stats(self, reset=True):
        """Episodes finished by the goal environment since the counters were last reset."""
        st = EpisodeStats()
        check(self.lib.mobrob_ppo_episode_stats(self._h, C.byref(st), int(bool(reset))))
        n = int(st.episodes)
        return {"episodes": n, "goals": int(st.goals), "ep_rew_mean": st.return_sum / n if n else float("nan"),
                "ep_len_mean": st.length_sum / n if n else float("nan")}

    def episode_records(self, max_records=100):
        """Monitor records [{r, l}] of the episodes the device goal environment finished since the last call
        (oldest first, the newest `max_records` at most)."""
        out = np.zeros((int(max_records), 2), F32)
        n = check(self.lib.mobrob_ppo_episode_records(self._h, _fp(out), int(max_records)))
        return [{"r": float(r), "l": int(l)} for r, l in out[:n]]

    # ---- env-side controllers (batched on the device, csrc/robot_ctrl.h) ---------------------------
    def ctrl_turtlebot3(self, pos, theta, goal, gain_changes):
        """Turtlebot3's proportional controller for n robots: -> twist [n, 2] = (v, w)."""
        pos, goal, gc = _f32c(pos), _f32c(goal), _f32c(gain_changes)
        theta = _f32c(theta)
        n = theta.shape[0]
        if pos.shape != (n, 2) or goal.shape != (n, 2) or gc.shape != (n, 2):
            raise ValueError("pos, goal, gain_changes must be [n, 2] and theta [n]")
        twist = np.empty((n, 2), F32)
        check(self.lib.mobrob_ctrl_turtlebot3(self._h, n, 0, _fp(pos), _fp(theta), _fp(goal), _fp(gc), _fp(twist)))
        return twist

    def ctrl_drone_pid(self, pos, rpy, goal, action, ctrl_state, mass, max_thrust, max_xy_torque, max_z_torque, g=9.8,
                       dt=1.0 / 50.0, max_roll_pitch=np.pi / 6, tune_fac=0.3):
        """The drone's cascaded PID for n robots; `ctrl_state` [n, 12] is updated in place -> [n, 4] thrust, torques."""
        pos, rpy, goal, action = _f32c(pos), _f32c(rpy), _f32c(goal), _f32c(action)
        n = pos.shape[0]
        if pos.shape != (n, 3) or rpy.shape != (n, 3) or goal.shape != (n, 3) or action.shape != (n, 18):
            raise ValueError("pos, rpy, goal must be [n, 3] and action [n, 18]")
        if ctrl_state.shape != (n, 12) or ctrl_state.dtype != F32 or not ctrl_state.flags.c_contiguous:
            raise ValueError("ctrl_state must be a C-contiguous float32 [n, 12] array")
        prm = DroneParams(mass, g, dt, max_thrust, max_xy_torque, max_z_torque, max_roll_pitch, tune_fac)
        out = np.empty((n, 4), F32)
        check(self.lib.mobrob_ctrl_drone_pid(self._h, n, 0, C.byref(prm), _fp(pos), _fp(rpy), _fp(goal), _fp(action),
                                             _fp(ctrl_state), _fp(out)))
        return out

    def compute_gae(self):
        check(self.lib.mobrob_ppo_compute_gae(self._h))

    def x3_mode(self):
        """Bit 0: rollout / value forward on the bf16 pipe with split float32 operands; bit 1: the gradient kernel's forward too."""
        return int(self.lib.mobrob_ppo_x3_mode(self._h))

    def update_mode(self):
        """Bit 0: the latest train() ran every epoch as one co-operative launch (k_epoch64) instead of three launches per step."""
        return int(self.lib.mobrob_ppo_update_mode(self._h))

    def explained_variance(self):
        """1 - Var[returns - values] / Var[returns] over the rollout in the buffer (SB3's train/explained_variance)."""
        out = C.c_double()
        check(self.lib.mobrob_ppo_explained_variance(self._h, C.byref(out)))
        return float(out.value)

    def mark_rollout_ready(self):
        check(self.lib.mobrob_ppo_mark_rollout_ready(self._h))

    # ---- update -------------------------------------------------------------------------------
    def train(self, perms=None):
        """Whole PPO.train().  perms: [n_epochs, T*N] int64 env-major permutations or None."""
        p = None
        if perms is not None:
            perms = np.ascontiguousarray(perms, dtype=np.int64)
            if perms.shape != (self.cfg.n_epochs, self.N * self.T):
                raise ValueError(f"perms must be [{self.cfg.n_epochs}, {self.N * self.T}]")
            p = perms.ctypes.data_as(C.POINTER(C.c_int64))
        st = TrainStats()
        check(self.lib.mobrob_ppo_train(self._h, p, C.byref(st)))
        return {k: float(getattr(st, k)) for k in STAT_KEYS} | {"n_minibatches": int(st.n_minibatches)}

    def comm_prepare(self):
        """Local, non-collective half of comm_init (RCCL loadable, device selectable, no communicator yet): ranks agree
        on its outcome before any of them enters the blocking collective."""
        check(self.lib.mobrob_ppo_comm_prepare(self._h))

    def comm_init(self, unique_id=None, rank=None, nranks=None):
        """Collective over the ranks of the job: build the engine's RCCL communicator.  Rank 0 calls
        `PPOEngine.comm_unique_id()` and ships the 128 bytes to the others (parallel.py uses torch.distributed).
        rank / nranks: position in the process (sub)group the communicator spans (default: the engine's config)."""
        buf = (C.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        if rank is None and nranks is None:
            check(self.lib.mobrob_ppo_comm_init(self._h, buf))
        else:
            check(self.lib.mobrob_ppo_comm_init_rank(self._h, buf, int(rank), int(nranks)))

    def comm_info(self):
        """(ncclCommCount, ncclCommUserRank) of the engine's communicator; (0, -1) without one."""
        n, r = C.c_int32(), C.c_int32()
        check(self.lib.mobrob_ppo_comm_info(self._h, C.byref(n), C.byref(r)))
        return int(n.value), int(r.value)

    def oneshot_export(self) -> bytes:
        """IPC handle of this rank's exchange buffer of the one-shot all-reduce (mobrob_ppo_oneshot_export)."""
        buf = (C.c_uint8 * 64)()
        check(self.lib.mobrob_ppo_oneshot_export(self._h, buf))
        return bytes(buf)

    def oneshot_open(self, handles, rank, nranks):
        """handles: the ranks' export handles in rank order; afterwards train_dp() exchanges through peer-mapped memory."""
        blob = b"".join(bytes(h) for h in handles)
        if len(blob) != 64 * int(nranks):
            raise ValueError(f"expected {nranks} handles of 64 bytes")
        buf = (C.c_uint8 * len(blob)).from_buffer_copy(blob)
        check(self.lib.mobrob_ppo_oneshot_open(self._h, buf, int(rank), int(nranks)))

    def oneshot_close(self):
        check(self.lib.mobrob_ppo_oneshot_close(self._h))

    def exchange_selfcheck(self, which):
        """A known vector through the engine's RCCL communicator (which = "rccl") or its one-shot exchange ("oneshot"), compared
        with the rank-ordered sum (collective).  -> number of message elements that are not bit-equal to it (0 = sound)."""
        bad = C.c_int32(-1)
        check(self.lib.mobrob_ppo_exchange_selfcheck(self._h, {"rccl": 0, "oneshot": 1}[which], C.byref(bad)))
        return int(bad.value)

    def allreduce_counters(self, reset=False):
        """(calls, payload bytes) of the all-reduces train_dp issued since the last reset."""
        c, b = C.c_int64(), C.c_int64()
        check(self.lib.mobrob_ppo_allreduce_counters(self._h, C.byref(c), C.byref(b), int(bool(reset))))
        return int(c.value), int(b.value)

    @staticmethod
    def comm_unique_id() -> bytes:
        buf = (C.c_uint8 * 128)()
        check(_lib.load().mobrob_ppo_comm_unique_id(buf))
        return bytes(buf)

    def train_dp(self, perms=None, allreduce=None):
        """PPO.train() across the ranks, the whole loop in C (mobrob_ppo_train_dp): one all-reduce of the gradient per
        optimizer step on the engine's stream.  allreduce: None -> RCCL on the communicator of comm_init; or a Python
        callable (device_ptr, count, dtype_code, hip_stream) -> None that sums in place (tests)."""
        p = None
        if perms is not None:
            perms = np.ascontiguousarray(perms, dtype=np.int64)
            if perms.shape != (self.cfg.n_epochs, self.N * self.T):
                raise ValueError(f"perms must be [{self.cfg.n_epochs}, {self.N * self.T}]")
            p = perms.ctypes.data_as(C.POINTER(C.c_int64))
        if allreduce is None:
            check(self.lib.mobrob_ppo_train_dp(self._h, p, None, None))
            return
        failure = []

        def trampoline(_ctx, buf, count, dtype, stream):
            try:
                allreduce(int(buf), int(count), int(dtype), int(stream or 0))
                return 0
            except BaseException as ex:  # noqa: BLE001 - must not propagate through the C frame
                failure.append(ex)
                return 1

        cb = _lib.ALLREDUCE_FN(trampoline)
        rc = self.lib.mobrob_ppo_train_dp(self._h, p, C.cast(cb, C.c_void_p), None)
        if failure:
            raise failure[0]
        check(rc)

    def set_hyper(self, **values):
        """learning_rate / clip_range (schedule values for the coming train()), clip_range_vf (None = off), target_kl
        (None = off), ent_coef, vf_coef -> mobrob_ppo_set_hyper."""
        for name, v in values.items():
            if name not in _lib.HYPER:
                raise ValueError(f"unknown hyper-parameter {name!r} (known: {sorted(_lib.HYPER)})")
            check(self.lib.mobrob_ppo_set_hyper(self._h, _lib.HYPER[name], -1.0 if v is None else float(v)))

    def last_train_info(self):
        """(epochs started, stopped early by target_kl, optimizer steps applied) of the latest train()."""
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        check(self.lib.mobrob_ppo_last_train_info(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return int(a.value), bool(b.value), int(c.value)

    def train_enqueue(self):
        """PPO.train() enqueued on the engine's stream without waiting (device-drawn permutations)."""
        check(self.lib.mobrob_ppo_train_enqueue(self._h, None))

    def train_stats(self):
        """Means over the last epoch's minibatches of the most recent update (waits for the stream)."""
        rows = self.fetch_step_stats(self.n_minibatches)
        m = np.zeros(7)
        if len(rows):
            r = rows.astype(np.float64)
            m = r.mean(axis=0)
            ok = ~np.isnan(r[:, 6])  # a step dropped by target_kl has no gradient norm
            m[6] = r[ok, 6].mean() if ok.any() else 0.0
        return {k: float(m[i]) for i, k in enumerate(STAT_KEYS)}

    def epoch_begin(self, perm=None):
        p = None
        if perm is not None:
            perm = np.ascontiguousarray(perm, dtype=np.int64)
            if perm.shape != (self.N * self.T,):
                raise ValueError("perm must have T*N entries")
            p = perm.ctypes.data_as(C.POINTER(C.c_int64))
        check(self.lib.mobrob_ppo_epoch_begin(self._h, p))

    def minibatch_grad(self, mb):
        check(self.lib.mobrob_ppo_minibatch_grad(self._h, int(mb)))

    def minibatch_apply(self):
        check(self.lib.mobrob_ppo_minibatch_apply(self._h))

    def minibatch_apply_checked(self):
        """minibatch_apply behind SB3's target_kl check -> True when the step was DROPPED (the driver ends train())."""
        stopped = C.c_int32()
        check(self.lib.mobrob_ppo_minibatch_apply_checked(self._h, C.byref(stopped)))
        return bool(stopped.value)

    def fetch_step_stats(self, max_rows=None):
        max_rows = int(max_rows or (self.n_minibatches * self.cfg.n_epochs))
        out = np.zeros((max_rows, 8), F32)
        n = check(self.lib.mobrob_ppo_fetch_step_stats(self._h, _fp(out), max_rows))
        return out[:n, :7]

    # ---- inference ----------------------------------------------------------------------------
    def predict(self, obs, deterministic=True, eps=None, want_values=False):
        obs = np.asarray(obs)
        single = obs.ndim == 1
        x = _f32c(obs[None] if single else obs)
        if x.ndim != 2 or x.shape[1] != self.D:
            raise ValueError(f"Unexpected observation shape {obs.shape} for Box environment with shape ({self.D},)")
        n = x.shape[0]
        act = np.empty((n, self.A), F32)
        val = np.empty(n, F32) if want_values else None
        eps = None if eps is None else _f32c(eps, (n, self.A))
        check(self.lib.mobrob_ppo_predict(self._h, _fp(x), n, int(bool(deterministic)), _fp(eps), _fp(act), _fp(val)))
        act = act[0] if single else act
        return (act, val) if want_values else act

    # ---- buffers ------------------------------------------------------------------------------
    def _buf_shape(self, name):
        T, N = self.T, self.N
        return {"obs": ((T + 1, N, self.D), F32), "actions": ((T, N, self.A), F32), "rewards": ((T, N), F32),
                "episode_starts": ((T, N), F32), "values": ((T, N), F32), "log_probs": ((T, N), F32),
                "advantages": ((T, N), F32), "returns": ((T, N), F32), "params": ((self.P,), F32),
                "grads": ((self.P,), F32), "advstat": ((self.n_minibatches, 4), np.float64),
                "last_values": ((N,), F32), "last_dones": ((N,), F32), "clipped_actions": ((N, self.A), F32), "episode_start_state": ((N,), F32),
                "terminal_obs": ((N, 16 * ((self.D + 15) // 16) if self.D <= 64 else 8 * ((self.D + 7) // 8)), F32), "terminal_values": ((N,), F32),
                "truncated": ((N,), np.uint8), "env_state": ((N, 12), F32), "grad_exchange": ((self.P + 8,), F32),
                "sde_noise": ((N, getattr(self, "HL", 0), self.A), F32)}[name]

    def read(self, name):
        shape, dt = self._buf_shape(name)
        out = np.empty(shape, dt)
        check(self.lib.mobrob_ppo_read_buffer(self._h, BUF[name], out.ctypes.data_as(C.c_void_p), out.nbytes))
        return out

    def write(self, name, arr):
        shape, dt = self._buf_shape(name)
        a = np.ascontiguousarray(arr, dtype=dt)
        if a.shape != shape:
            raise ValueError(f"{name}: expected {shape}, got {a.shape}")
        check(self.lib.mobrob_ppo_write_buffer(self._h, BUF[name], a.ctypes.data_as(C.c_void_p), a.nbytes))

    def device_buffer(self, name):
        """(device pointer, bytes) -- e.g. to wrap grads/advstat for a collective."""
        p, b = C.c_void_p(), C.c_size_t()
        check(self.lib.mobrob_ppo_buffer_info(self._h, BUF[name], C.byref(p), C.byref(b)))
        return int(p.value), int(b.value)

    def load_rollout(self, buf, last_values, dones):
        """Inject a complete rollout (tests): dict of [T,N,..] arrays as in the oracle."""
        obs = np.zeros((self.T + 1, self.N, self.D), F32)
        obs[:self.T] = buf["obs"]
        self.write("obs", obs)
        for k in ("actions", "rewards", "episode_starts", "values", "log_probs"):
            self.write(k, buf[k])
        self.write("last_values", last_values)
        self.write("last_dones", np.asarray(dones, F32))
        if "advantages" in buf:
            self.write("advantages", buf["advantages"])
            self.write("returns", buf["returns"])
            self.mark_rollout_ready()

    def feistel_permutation(self, n, key):
        out = np.empty(int(n), np.int64)
        check(self.lib.mobrob_ppo_feistel_permutation(self._h, int(n), int(key) & (2 ** 64 - 1),
                                                      out.ctypes.data_as(C.POINTER(C.c_int64))))
        return out

    def pinned(self, shape, dtype=np.float32):
        """Zero-filled NumPy array backed by pinned, device-visible host memory (mobrob_ppo_host_alloc): the kernels
        read / write such buffers in place (no staging copy).  The allocation lives as long as any view of the
        array does (see _PinnedBlock), also past close()."""
        dtype = np.dtype(dtype)
        shape = tuple(int(s) for s in np.atleast_1d(shape)) if not isinstance(shape, tuple) else shape
        block = _PinnedBlock(self.lib, int(np.prod(shape)) * dtype.itemsize, shape, dtype)
        return np.asarray(block)

    def register_host(self, address, nbytes):
        """Pin + map caller-owned host memory in place (mobrob_ppo_host_register), e.g. the shared block of
        `ShmVecEnv`; unregistered by unregister_host() or close()."""
        check(self.lib.mobrob_ppo_host_register(C.c_void_p(int(address)), C.c_size_t(int(nbytes))))
        self._registered = getattr(self, "_registered", [])
        self._registered.append(int(address))

    def unregister_host(self, address):
        if int(address) in getattr(self, "_registered", []):
            self._registered.remove(int(address))
            check(self.lib.mobrob_ppo_host_unregister(C.c_void_p(int(address))))

    def set_stream(self, stream_handle):
        check(self.lib.mobrob_ppo_set_stream(self._h, C.c_void_p(stream_handle)))

    def synchronize(self):
        check(self.lib.mobrob_ppo_synchronize(self._h))

    def profile(self, on=True, only=None):
        """HIP-event bracketing of the engine's phases.  only: names from _lib.KERNEL_IDS to restrict it to (each
        bracketed launch costs a few microseconds of GPU time)."""
        v = int(bool(on))
        if on and only is not None:
            v = sum(1 << (_lib.KERNEL_IDS[k] + 1) for k in only)
        check(self.lib.mobrob_ppo_profile_enable(self._h, v))

    def profile_read(self):
        n = len(_lib.KERNEL_IDS)  # MOBROB_K_COUNT
        ms, calls = (C.c_double * n)(), (C.c_int64 * n)()
        check(self.lib.mobrob_ppo_profile_read(self._h, ms, calls))
        return {k: (float(ms[i]), int(calls[i])) for k, i in _lib.KERNEL_IDS.items()}
