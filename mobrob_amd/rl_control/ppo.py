"""PPO controller -- mirror of the reference's `PPOCtrl` (/root/reference/src/mobrob/rl_control/ppo.py:14-77)
and of the slice of `stable_baselines3.PPO` that the reference touches:

    PPO(env=..., seed=..., tensorboard_log=..., **ppo_kwargs)         ppo.py:50-59
    .learn(total_timesteps, callback, progress_bar)                   ppo.py:73-74, examples/train.py:42-46
    .save(path) / PPO.load(path)                                      ppo.py:76-77, utils.py:15-16, train.py:30-33
    .policy.state_dict() / .policy.load_state_dict(...)               train.py:31-33
    .predict(obs, deterministic=True) -> (action, None)               examples/control.py:39
    CheckpointCallback(save_freq, save_path, name_prefix, verbose)    train.py:36-41

All arithmetic runs in the HIP engine (libmobrob_ppo.so) on an MI355X; there is no CPU fallback.  The YAML's
`device: cpu` is accepted (schema compatibility) and ignored.
"""
from __future__ import annotations

import os
import sys
import time
from collections import OrderedDict, deque

import numpy as np

from ..checkpoint import load_zip, save_zip
from ..engine import MAX_HIDDEN, PPOEngine, activation_name
from ..envs.vec_env import DeviceGoalVecEnv, DeviceSyntheticVecEnv, HostVecEnv, SyntheticVecEnv, make_vec_env
from ..envs.native_env import NativeGoalVecEnv
from ..envs.shm_vec_env import ShmVecEnv
from ..envs.wrapper import ROBOT_DIMS, get_env
from ..utils import DATA_DIR
from .init import orthogonal_policy_init, policy_init

try:
    import tensorboard  # noqa: F401
except ImportError:
    tensorboard = None

DummyVecEnv = HostVecEnv   # reference ppo.py:32-33: environments stepped in the learner process
SubprocVecEnv = ShmVecEnv  # reference ppo.py:30-31: environments stepped by worker processes on the host cores


# --------------------------------------------------------------------------------------------------
# callbacks (SB3 BaseCallback protocol, the part CheckpointCallback needs)
# --------------------------------------------------------------------------------------------------
class BaseCallback:
    def __init__(self, verbose: int = 0):
        self.verbose, self.model, self.n_calls, self.num_timesteps = verbose, None, 0, 0

    def init_callback(self, model):
        self.model = model

    def on_training_start(self, locals_=None, globals_=None):
        pass

    def on_rollout_start(self):
        pass

    def on_step(self) -> bool:
        self.n_calls += 1
        self.num_timesteps = self.model.num_timesteps
        return self._on_step()

    def _on_step(self) -> bool:
        return True

    def on_rollout_end(self):
        pass

    def on_training_end(self):
        pass


class CheckpointCallback(BaseCallback):
    """Saves `<save_path>/<name_prefix>_<num_timesteps>_steps.zip` every `save_freq` calls (Appendix A.9)."""

    def __init__(self, save_freq: int, save_path: str, name_prefix: str = "rl_model", verbose: int = 0):
        super().__init__(verbose)
        self.save_freq, self.save_path, self.name_prefix = max(int(save_freq), 1), save_path, name_prefix

    def init_callback(self, model):
        super().init_callback(model)
        if self.save_path is not None:
            os.makedirs(self.save_path, exist_ok=True)

    def _on_step(self) -> bool:
        if self.n_calls % self.save_freq == 0:
            path = os.path.join(self.save_path, f"{self.name_prefix}_{self.num_timesteps}_steps.zip")
            self.model.save(path)
            if self.verbose >= 2:
                print(f"Saving model checkpoint to {path}")
        return True


class _CallbackList(BaseCallback):
    def __init__(self, cbs):
        super().__init__()
        self.cbs = cbs

    def init_callback(self, model):
        super().init_callback(model)
        for c in self.cbs:
            c.init_callback(model)

    def on_training_start(self, l=None, g=None):
        for c in self.cbs:
            c.on_training_start(l, g)

    def on_rollout_start(self):
        for c in self.cbs:
            c.on_rollout_start()

    def _on_step(self):
        ok = True
        for c in self.cbs:
            ok = c.on_step() and ok
        return ok

    def on_rollout_end(self):
        for c in self.cbs:
            c.on_rollout_end()

    def on_training_end(self):
        for c in self.cbs:
            c.on_training_end()


# --------------------------------------------------------------------------------------------------
# policy handle
# --------------------------------------------------------------------------------------------------
class ActorCriticPolicyHandle:
    """`ppo.policy`: state_dict()/load_state_dict() with SB3's 13 keys; tensors are torch CPU tensors."""

    def __init__(self, model):
        self._m = model

    def state_dict(self):
        import torch
        return OrderedDict((k, torch.from_numpy(v.copy())) for k, v in self._m.engine.get_params().items())

    def load_state_dict(self, state_dict, strict: bool = True):
        cur = self._m.engine.get_params()
        missing = [k for k in cur if k not in state_dict]
        unexpected = [k for k in state_dict if k not in cur]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict: missing keys {missing}, unexpected keys {unexpected}")
        for k in cur:
            if k in state_dict:
                v = state_dict[k]
                v = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
                if v.shape != cur[k].shape:
                    raise RuntimeError(f"size mismatch for {k}: copying a param with shape {tuple(v.shape)} from "
                                       f"checkpoint, the shape in current model is {tuple(cur[k].shape)}")
                cur[k] = v.astype(np.float32)
        self._m.engine.set_params(cur)

    def predict(self, observation, state=None, episode_start=None, deterministic: bool = False):
        return self._m.predict(observation, state, episode_start, deterministic)

    def set_training_mode(self, mode: bool):
        pass


_SB3_DEFAULTS = dict(learning_rate=3e-4, n_steps=2048, batch_size=64, n_epochs=10, gamma=0.99, gae_lambda=0.95,
                     clip_range=0.2, clip_range_vf=None, normalize_advantage=True, ent_coef=0.0, vf_coef=0.5,
                     max_grad_norm=0.5, use_sde=False, sde_sample_freq=-1, target_kl=None, stats_window_size=100)


class PPO:
    def __init__(self, policy="MlpPolicy", env=None, learning_rate=3e-4, n_steps=2048, batch_size=64, n_epochs=10,
                 gamma=0.99, gae_lambda=0.95, clip_range=0.2, clip_range_vf=None, normalize_advantage=True,
                 ent_coef=0.0, vf_coef=0.5, max_grad_norm=0.5, use_sde=False, sde_sample_freq=-1, target_kl=None,
                 stats_window_size=100, tensorboard_log=None, policy_kwargs=None, verbose=0, seed=None, device="auto",
                 _init_setup_model=True, _dims=None, _engine_kwargs=None):
        if policy not in ("MlpPolicy",) and getattr(policy, "__name__", "") != "ActorCriticPolicy":
            raise ValueError(f"Policy {policy} unknown")
        self.policy_class = "MlpPolicy"
        self.env = env
        # learning_rate / clip_range / clip_range_vf may be floats or SB3 schedules: callables of progress_remaining
        # (1 at the start of learn(), 0 at the end), evaluated once per train() like SB3's _update_learning_rate
        self.lr_schedule = learning_rate if callable(learning_rate) else None
        self.clip_schedule = clip_range if callable(clip_range) else None
        self.clip_vf_schedule = clip_range_vf if callable(clip_range_vf) else None
        self.learning_rate = float(learning_rate(1.0)) if callable(learning_rate) else float(learning_rate)
        self.n_steps, self.batch_size, self.n_epochs = int(n_steps), int(batch_size), int(n_epochs)
        self.gamma, self.gae_lambda = float(gamma), float(gae_lambda)
        self.clip_range = float(clip_range(1.0)) if callable(clip_range) else float(clip_range)
        if clip_range_vf is not None and not callable(clip_range_vf) and not float(clip_range_vf) > 0:
            raise ValueError("`clip_range_vf` must be positive, pass `None` to deactivate vf clipping")
        self.clip_range_vf = (float(clip_range_vf(1.0)) if callable(clip_range_vf)
                              else (None if clip_range_vf is None else float(clip_range_vf)))
        self.normalize_advantage = bool(normalize_advantage)
        self.ent_coef, self.vf_coef, self.max_grad_norm = float(ent_coef), float(vf_coef), float(max_grad_norm)
        # generalised state-dependent exploration (SB3 StateDependentNoiseDistribution with its defaults): log_std is a
        # [last policy width, act_dim] matrix, the exploration matrices are redrawn every sde_sample_freq rollout steps
        self.use_sde, self.sde_sample_freq = bool(use_sde), int(sde_sample_freq)
        self.target_kl = None if target_kl is None else float(target_kl)
        self.tensorboard_log, self.verbose, self.seed, self.device = tensorboard_log, int(verbose), seed, device
        self.policy_kwargs = dict(policy_kwargs or {})
        # keys some SB3 writers leave inside policy_kwargs: `use_sde` (the off-policy algorithms store it there; it must agree with the
        # argument) and the deprecated `sde_net_arch=None`
        if "use_sde" in self.policy_kwargs and bool(self.policy_kwargs.pop("use_sde")) != bool(use_sde):
            raise ValueError("policy_kwargs['use_sde'] contradicts PPO(use_sde=...)")
        if self.policy_kwargs.get("sde_net_arch", None) is None:
            self.policy_kwargs.pop("sde_net_arch", None)
        net_arch = self.policy_kwargs.get("net_arch", dict(pi=[64, 64], vf=[64, 64]))
        if isinstance(net_arch, (list, tuple)):
            net_arch = dict(pi=list(net_arch), vf=list(net_arch))
        self.net_arch = (tuple(net_arch.get("pi", [64, 64])), tuple(net_arch.get("vf", [64, 64])))
        # SB3 ActorCriticPolicy keyword arguments beyond the reference YAMLs' `net_arch` (the reference splats
        # `ppo_kwargs` into PPO verbatim, /root/reference/src/mobrob/rl_control/ppo.py:58):
        #   log_std_init     initial value of the state-independent log standard deviation (default 0.0)
        #   ortho_init       True (default): orthogonal weights with SB3's gains; False: torch's nn.Linear default
        #   optimizer_kwargs Adam's `eps` / `betas` (SB3 passes eps=1e-5 itself); `optimizer_class` must stay Adam
        #   activation_fn    nn.Tanh (default), nn.ReLU, nn.ELU, nn.LeakyReLU, nn.Sigmoid, nn.Softplus, nn.Softsign, nn.Hardtanh,
        #                    nn.ReLU6, as a class or by name; all but tanh run the generic GEMM chain (the fused kernels'
        #                    epilogues are tanh)
        #   share_features_extractor  accepted: MlpPolicy's extractor is nn.Flatten (no parameters), shared or not is the same network
        self.log_std_init = float(self.policy_kwargs.get("log_std_init", 0.0))
        self.ortho_init = bool(self.policy_kwargs.get("ortho_init", True))
        opt_kw = dict(self.policy_kwargs.get("optimizer_kwargs") or {})
        self.adam_eps = float(opt_kw.pop("eps", 1e-5))
        self.adam_betas = tuple(float(b) for b in opt_kw.pop("betas", (0.9, 0.999)))
        if opt_kw.pop("weight_decay", 0.0) or opt_kw.pop("amsgrad", False) or opt_kw:
            raise NotImplementedError("optimizer_kwargs other than `eps` and `betas` are not supported (Adam without weight "
                                      "decay or amsgrad, SB3's default optimiser)")
        self.activation = activation_name(self.policy_kwargs.get("activation_fn"))   # NotImplementedError by name for the others
        oc = self.policy_kwargs.get("optimizer_class")
        if oc is not None and getattr(oc, "__name__", str(oc)) != "Adam":
            raise NotImplementedError(f"optimizer_class {oc!r}: only Adam is implemented")
        # MlpPolicy's features extractor is nn.Flatten (no parameters): sharing it or not is the same network; `normalize_images`
        # concerns image spaces only -- both are accepted and travel with the checkpoint
        unknown = set(self.policy_kwargs) - {"net_arch", "log_std_init", "ortho_init", "optimizer_kwargs", "activation_fn",
                                             "optimizer_class", "share_features_extractor", "normalize_images", "full_std",
                                             "use_expln", "squash_output"}
        # gSDE options of ActorCriticPolicy (they only matter with use_sde): full_std / use_expln are served, squashing is not
        self.sde_full_std = bool(self.policy_kwargs.get("full_std", True))
        self.sde_use_expln = bool(self.policy_kwargs.get("use_expln", False))
        if self.policy_kwargs.get("squash_output", False):
            raise NotImplementedError("policy_kwargs squash_output=True (tanh-squashed gSDE actions) is not implemented")
        if unknown:
            raise NotImplementedError(f"policy_kwargs {sorted(unknown)} are not supported (MlpPolicy, one to {MAX_HIDDEN} hidden "
                                      "layers per network)")
        self.num_timesteps = 0
        self._total_timesteps = 0
        self._num_timesteps_at_start = 0
        self._n_updates = 0
        self._episode_num = 0
        self._current_progress_remaining = 1.0
        self._last_obs = None
        self._last_episode_starts = None
        self.start_time = None
        self.ep_info_buffer = deque(maxlen=stats_window_size)
        self._stats_window_size = stats_window_size
        self._engine_kwargs = dict(_engine_kwargs or {})
        if env is not None:
            self.n_envs, self.obs_dim, self.act_dim = env.num_envs, env.obs_dim, env.act_dim
        elif _dims is not None:
            self.n_envs, self.obs_dim, self.act_dim = _dims
        else:
            raise ValueError("PPO needs an environment (or load a checkpoint with PPO.load)")
        # observation bounds travel with the model like SB3's `self.observation_space` (PPO.save writes them,
        # PPO.load(path, env=...) of real SB3 checks them against the env): from the env, else from the checkpoint
        sp = getattr(env, "observation_space", None)
        self.obs_bounds = None if sp is None else (np.asarray(sp.low, np.float32), np.asarray(sp.high, np.float32))
        self.engine = None
        self.policy = None
        self.world_size, self.rank, self._backend = 1, 0, None
        self._host_bufs = None
        # row ranges of the pipelined host-env rollout (engine.part_pipeline); 1 = whole batch per step
        self.host_parts = int(self._engine_kwargs.pop("host_parts", 2))
        if _init_setup_model:
            self._setup_model()

    # ---------------------------------------------------------------------------------------------
    def _setup_model(self):
        if self.n_steps * self.n_envs <= 1:
            raise AssertionError("`n_steps * n_envs` must be greater than 1")
        kw = dict(obs_dim=self.obs_dim, act_dim=self.act_dim, n_envs=self.n_envs, n_steps=self.n_steps,
                  batch_size=self.batch_size, n_epochs=self.n_epochs, pi=self.net_arch[0], vf=self.net_arch[1],
                  gamma=self.gamma, gae_lambda=self.gae_lambda, clip_range=self.clip_range, ent_coef=self.ent_coef,
                  vf_coef=self.vf_coef, max_grad_norm=self.max_grad_norm, learning_rate=self.learning_rate,
                  normalize_advantage=self.normalize_advantage, seed=0 if self.seed is None else int(self.seed),
                  adam_betas=self.adam_betas, adam_eps=self.adam_eps, activation=self.activation, use_sde=self.use_sde,
                  sde_sample_freq=self.sde_sample_freq, sde_full_std=self.sde_full_std, sde_use_expln=self.sde_use_expln)
        # data parallel: under torchrun / an initialised process group every rank owns its n_envs
        # environments and rollout shard; batch_size stays SB3's GLOBAL minibatch and must divide by the world size
        from ..parallel import distributed_context
        self.world_size, self.rank, local_rank = distributed_context()
        if self.world_size > 1:
            kw.update(rank=self.rank, world_size=self.world_size, device_id=local_rank)
        kw.update(self._engine_kwargs)
        self.engine = PPOEngine(**kw)
        self.engine.set_hyper(clip_range_vf=self.clip_range_vf, target_kl=self.target_kl)
        self._backend = None
        self.engine.set_params(policy_init(self.obs_dim, self.act_dim, self.net_arch[0], self.net_arch[1],
                                           seed=0 if self.seed is None else int(self.seed),
                                           log_std_init=self.log_std_init, ortho_init=self.ortho_init, use_sde=self.use_sde,
                                           full_std=self.sde_full_std))
        self.policy = ActorCriticPolicyHandle(self)

    def get_env(self):
        return self.env

    def set_env(self, env):
        if (env.num_envs, env.obs_dim, env.act_dim) != (self.n_envs, self.obs_dim, self.act_dim):
            raise ValueError("environment does not match the model's (n_envs, obs_dim, act_dim)")
        if getattr(self.env, "_registered_with", None) is self.engine and env is not self.env:
            self.env._registered_with = None
            self.engine.unregister_host(self.env.shared_block()[0])
        self.env = env
        sp = getattr(env, "observation_space", None)
        if sp is not None:
            self.obs_bounds = (np.asarray(sp.low, np.float32), np.asarray(sp.high, np.float32))
        self._last_obs = None
        self._host_bufs = None  # staging is bound to an environment (use_buffers / shared block): rebuilt in learn()

    # ---------------------------------------------------------------------------------------------
    def predict(self, observation, state=None, episode_start=None, deterministic: bool = False):
        """-> (actions clipped to the Box, None) like SB3's BasePolicy.predict (Appendix A.10)."""
        act = self.engine.predict(np.asarray(observation, dtype=np.float32), deterministic=deterministic)
        return act, None

    # ---------------------------------------------------------------------------------------------
    def _collect_rollouts(self, callback) -> bool:
        e, env, N = self.engine, self.env, self.n_envs
        callback.on_rollout_start()
        if isinstance(env, (DeviceSyntheticVecEnv, DeviceGoalVecEnv)):
            if isinstance(env, DeviceGoalVecEnv):
                env.collect(e)
                st = e.episode_stats(reset=True)  # Monitor statistics of the episodes this rollout finished
                self.device_episode_stats = st
                self._episode_num += st["episodes"]
                # real Monitor records: the device keeps (return, length) of the last 128 finished episodes
                now = round((time.time_ns() - (self.start_time or time.time_ns())) / 1e9, 6)
                self.ep_info_buffer.extend({"r": r["r"], "l": r["l"], "t": now}
                                           for r in e.episode_records(self.ep_info_buffer.maxlen or 100))
            else:
                e.collect_synthetic(env.p_term, env.time_limit)
            # The whole rollout ran in one launch; the per-step callbacks are replayed afterwards.  A checkpoint taken at
            # "step k of the rollout" holds what SB3's would: weights, optimizer state and update counters only change in
            # train(), and num_timesteps is the replayed value -- only `_last_obs` is the rollout's last observation
            # rather than step k's.
            for _ in range(self.n_steps):
                self.num_timesteps += N * self.world_size  # time/total_timesteps counts the whole job
                if not callback.on_step():
                    return False
            callback.on_rollout_end()
            return True
        if hasattr(env, "step_arrays"):
            return self._collect_rollouts_arrays(callback)
        e.rollout_begin()
        dones = np.zeros(N, bool)
        for _ in range(self.n_steps):
            _, clipped, _, _ = e.act(self._last_obs, want_all=False)
            new_obs, rewards, dones, infos = env.step(clipped)
            self.num_timesteps += N * self.world_size
            if not callback.on_step():
                return False
            trunc, term_obs = None, None
            for i, info in enumerate(infos):
                ep = info.get("episode")
                if ep is not None:
                    self.ep_info_buffer.append(ep)
                if dones[i] and info.get("terminal_observation") is not None and info.get("TimeLimit.truncated", False):
                    if trunc is None:
                        trunc, term_obs = np.zeros(N, np.uint8), np.zeros((N, self.obs_dim), np.float32)
                    trunc[i], term_obs[i] = 1, info["terminal_observation"]
            e.store(rewards, dones, trunc, term_obs)
            self._last_obs, self._last_episode_starts = new_obs, dones
        e.finish_rollout(self._last_obs, dones)
        callback.on_rollout_end()
        return True

    def _setup_host_buffers(self):
        """Pinned staging shared by the env and the engine: the env writes step results where the GPU reads them.
        `ShmVecEnv` owns its block (worker processes have it mapped): the engine pins and maps it in place;
        environments with `use_buffers` are pointed at fresh pinned arrays instead."""
        e, N = self.engine, self.n_envs
        if hasattr(self.env, "shared_block"):
            if getattr(self.env, "_registered_with", None) is not e:
                e.register_host(*self.env.shared_block())
                self.env._registered_with = e
            self._host_bufs = self.env.buffers()
            return
        self._host_bufs = dict(obs=e.pinned((N, self.obs_dim)), clip=e.pinned((N, self.act_dim)), rew=e.pinned((N,)),
                               done=e.pinned((N,), np.uint8), trunc=e.pinned((N,), np.uint8),
                               term=e.pinned((N, self.obs_dim)))
        b = self._host_bufs
        self.env.use_buffers(obs=b["obs"], rewards=b["rew"], dones=b["done"], truncated=b["trunc"], terminal_obs=b["term"])

    def _collect_rollouts_arrays(self, callback) -> bool:
        """Host env with the array protocol (NativeGoalVecEnv): O(1) Python work per step; observations, rewards and
        flags live in pinned staging that the GPU reads and writes in place.  With `host_parts` > 1 (default 2 for
        envs that can step a row range) the rollout is pipelined: the simulator steps one range of robots while the
        GPU runs the policy for the others -- same results as the whole-batch loop."""
        e, env, N, b = self.engine, self.env, self.n_envs, self._host_bufs
        # a handful of environments (the reference YAMLs: 2-16) leave nothing to overlap: one range, half the launches,
        # event waits and worker round trips per step
        parts = self.host_parts if hasattr(env, "step_range") and N >= 64 else 1
        finished = False
        e.rollout_begin()
        if type(callback) is BaseCallback and hasattr(env, "step_range_fn"):
            # nothing to call per step and a natively stepped env: the whole collector loop runs in C (mobrob_ppo_collect_host),
            # served by the persistent rollout kernel where the engine can -- also ONE range of a handful of environments (the
            # reference YAMLs' 2 - 16): no launch, no synchronisation and no Python frame per step
            pipe = e.part_pipeline(parts, b["obs"], b["clip"], b["rew"], b["done"], b["trunc"], b["term"])
            pipe.collect(env.step_range_fn, env.handle)  # includes finish_rollout
            finished = True
            self.num_timesteps += N * self.world_size * self.n_steps
            callback.n_calls += self.n_steps
            callback.num_timesteps = self.num_timesteps
        elif parts > 1:
            pipe = e.part_pipeline(parts, b["obs"], b["clip"], b["rew"], b["done"], b["trunc"], b["term"])
            for p in range(parts):
                pipe.act(p)
            for t in range(self.n_steps):
                for p in range(parts):
                    pipe.wait(p)
                    ntrunc = env.step_range(*pipe.bounds[p], b["clip"])
                    pipe.store(p, ntrunc > 0)
                    if t + 1 < self.n_steps:
                        pipe.act(p)
                self.num_timesteps += N * self.world_size
                if not callback.on_step():
                    return False
        else:
            for _ in range(self.n_steps):
                e.act(b["obs"], out_clipped=b["clip"], want_all=False)
                _, _, _, _, _, ntrunc = env.step_arrays(b["clip"])
                self.num_timesteps += N * self.world_size
                if not callback.on_step():
                    return False
                e.store(b["rew"], b["done"], b["trunc"] if ntrunc else None, b["term"] if ntrunc else None)
        if not finished:
            e.finish_rollout(b["obs"], b["done"])
        st = env.episode_stats(reset=True)
        self.device_episode_stats = st
        self._episode_num += st["episodes"]
        self.ep_info_buffer.extend(env.pop_episodes())  # real Monitor records {r, l, t}, in completion order
        callback.on_rollout_end()
        return True

    def train(self):
        p = self._current_progress_remaining
        if self.lr_schedule is not None or self.clip_schedule is not None or self.clip_vf_schedule is not None:
            if self.lr_schedule is not None:
                self.learning_rate = float(self.lr_schedule(p))
            if self.clip_schedule is not None:
                self.clip_range = float(self.clip_schedule(p))
            if self.clip_vf_schedule is not None:
                self.clip_range_vf = float(self.clip_vf_schedule(p))
            self.engine.set_hyper(learning_rate=self.learning_rate, clip_range=self.clip_range, clip_range_vf=self.clip_range_vf)
        if self.world_size > 1:
            from ..parallel import EngineBackend, train_data_parallel
            if self._backend is None:
                self._backend = EngineBackend(self.engine)
            epochs, stopped, applied = train_data_parallel(self._backend)
            # the loss sums travel with the gradient in the per-step all-reduce: every rank logs the global means
            stats = self.engine.train_stats()
            stats["n_minibatches"] = applied
            self._n_updates += epochs
            if stopped and self.verbose >= 1 and self.rank == 0:
                print(f"Early stopping at step {epochs - 1} due to reaching max kl: {stats['approx_kl']:.2f}")
        else:
            stats = self.engine.train(None)
            epochs, stopped, _ = self.engine.last_train_info()
            self._n_updates += epochs       # SB3 counts the epochs it started (target_kl may cut the last ones)
            if stopped and self.verbose >= 1:
                print(f"Early stopping at step {epochs - 1} due to reaching max kl: {stats['approx_kl']:.2f}")
        return stats

    _tb = None   # event-file writer of the learn() call in progress (tensorboard_log)

    def learn(self, total_timesteps, callback=None, log_interval=1, tb_log_name="PPO", reset_num_timesteps=True,
              progress_bar=False):
        if self.env is None:
            raise ValueError("learn() needs an environment: PPO.load(path, env=...) or set_env()")
        total_timesteps = int(total_timesteps)
        self.start_time = time.time_ns()
        if reset_num_timesteps:
            self.num_timesteps, self._episode_num = 0, 0
        else:
            total_timesteps += self.num_timesteps
        self._total_timesteps, self._num_timesteps_at_start = total_timesteps, self.num_timesteps
        host_env = not isinstance(self.env, (DeviceSyntheticVecEnv, DeviceGoalVecEnv))
        fresh_staging = host_env and hasattr(self.env, "step_arrays") and getattr(self, "_host_bufs", None) is None
        if fresh_staging:
            self._setup_host_buffers()
        # an array-protocol env reads its first observations from the staging: a `_last_obs` restored from a checkpoint
        # (PPO.load(force_reset=False)) describes environments that no longer exist, so new staging always starts with
        # a reset
        if host_env and (reset_num_timesteps or self._last_obs is None or fresh_staging):
            self._last_obs = self.env.reset()
            self._last_episode_starts = np.ones(self.n_envs, bool)
            self.engine.write("episode_start_state", np.ones(self.n_envs, np.float32))
        if callback is None:
            callback = BaseCallback()
        elif isinstance(callback, (list, tuple)):
            callback = _CallbackList(list(callback))
        callback.init_callback(self)
        callback.on_training_start(locals(), globals())
        bar = None
        if progress_bar:
            try:
                from tqdm import tqdm
                bar = tqdm(total=total_timesteps - self.num_timesteps, file=sys.stderr)
            except ImportError:
                bar = None
        iteration = 0
        self._tb = None
        if self.tensorboard_log is not None and self.rank == 0:   # SB3 configure_logger: <tensorboard_log>/<tb_log_name>_<run id>
            from ..tb_events import EventFileWriter, next_run_dir
            self._tb = EventFileWriter(next_run_dir(self.tensorboard_log, tb_log_name, continue_latest=not reset_num_timesteps))
        while self.num_timesteps < total_timesteps:
            before = self.num_timesteps
            go_on = self._collect_rollouts(callback)
            if self.world_size > 1:
                # a callback may stop SOME ranks only (it sees its own shard): the stop flag is agreed across the ranks
                # before anybody leaves the loop, so no rank enters train()'s collectives -- or the closing barrier -- alone
                from ..parallel import agree_all
                go_on = agree_all(go_on, group=getattr(self._backend, "_group", None), device_id=int(self.engine.cfg.device_id))
            if not go_on:
                break
            iteration += 1
            self._current_progress_remaining = 1.0 - float(self.num_timesteps) / float(total_timesteps)
            stats = self.train()
            if bar is not None:
                bar.update(self.num_timesteps - before)
            # SB3 dumps its logger every `log_interval` iterations whatever `verbose` is: stdout for verbose >= 1, the
            # event file whenever tensorboard_log is set
            if log_interval is not None and iteration % log_interval == 0 and (self.verbose >= 1 or self._tb is not None):
                self._log(iteration, stats)
        if bar is not None:
            bar.close()
        if self._tb is not None:
            self._tb.close()
            self._tb = None
        try:
            callback.on_training_end()
        finally:
            # EVERY rank reaches the collective closing handshake (parallel.EngineBackend.close: drain, barrier, unmap the one-shot
            # exchange), also the rank whose callback raised: its peers would otherwise wait in the barrier for it
            if self._backend is not None:
                self._backend.close()
        return self

    def _log(self, iteration, stats):
        """SB3's logger dump.  Data parallel: only rank 0 writes (stdout and event file alike), the other ranks return before
        any device read.  The loss statistics are GLOBAL means (the C loop all-reduces their sums with the gradient);
        `rollout/ep_*`, `train/explained_variance` and `time/fps` describe rank 0's shard of the environments (every shard
        draws from the same distribution: an unbiased estimate of the global figure from 1/world of the data)."""
        if self.rank != 0:
            return
        elapsed = max((time.time_ns() - self.start_time) / 1e9, sys.float_info.epsilon)
        fps = int((self.num_timesteps - self._num_timesteps_at_start) / elapsed)
        rows = []
        if len(self.ep_info_buffer) > 0:
            rows += [("rollout/ep_len_mean", float(np.mean([e["l"] for e in self.ep_info_buffer]))),
                     ("rollout/ep_rew_mean", float(np.mean([e["r"] for e in self.ep_info_buffer])))]
        rows += [("time/fps", fps), ("time/iterations", iteration), ("time/time_elapsed", int(elapsed)),
                 ("time/total_timesteps", self.num_timesteps), ("train/approx_kl", stats["approx_kl"]),
                 ("train/clip_fraction", stats["clip_fraction"]), ("train/clip_range", self.clip_range),
                 ("train/entropy_loss", stats["entropy_loss"]), ("train/learning_rate", self.learning_rate),
                 ("train/explained_variance", self.engine.explained_variance()),
                 ("train/loss", stats["loss"]), ("train/n_updates", self._n_updates),
                 ("train/policy_gradient_loss", stats["policy_loss"]),
                 ("train/std", float(np.mean(np.exp(self.engine.get_params()["log_std"])))),
                 ("train/value_loss", stats["value_loss"])]
        if self.clip_range_vf is not None and not callable(self.clip_range_vf):
            rows.append(("train/clip_range_vf", float(self.clip_range_vf)))
        if self._tb is not None:   # SB3 keeps these three out of the event file (logger.record(..., exclude="tensorboard"))
            skip = ("time/iterations", "time/time_elapsed", "time/total_timesteps")
            self._tb.add_scalars([(k, v) for k, v in rows if k not in skip], self.num_timesteps)
        if self.verbose < 1:
            return
        rows.sort()
        w = max(len(k) for k, _ in rows)
        print("-" * (w + 20))
        for k, v in rows:
            print(f"| {k:<{w}} | {v:<13.6g} |" if isinstance(v, float) else f"| {k:<{w}} | {v:<13} |")
        print("-" * (w + 20), flush=True)

    # ---------------------------------------------------------------------------------------------
    def _hyper(self):
        return dict(n_steps=self.n_steps, batch_size=self.batch_size, n_epochs=self.n_epochs, gamma=self.gamma,
                    gae_lambda=self.gae_lambda, ent_coef=self.ent_coef, vf_coef=self.vf_coef,
                    max_grad_norm=self.max_grad_norm, learning_rate=self.learning_rate, clip_range=self.clip_range,
                    normalize_advantage=self.normalize_advantage, n_envs=self.n_envs, clip_range_vf=self.clip_range_vf,
                    target_kl=self.target_kl, use_sde=self.use_sde, sde_sample_freq=self.sde_sample_freq)

    def save(self, path):
        """SB3-layout zip (checkpoint.py); appends .zip like SB3 when the suffix is missing.  Data-parallel replicas
        are identical: only rank 0 writes."""
        if getattr(self, "rank", 0) != 0:
            return
        m, v, step = self.engine.get_optimizer_state()
        d = os.path.dirname(str(path))
        if d:
            os.makedirs(d, exist_ok=True)
        explicit_arch = "net_arch" in self.policy_kwargs
        save_zip(path, params=self.engine.get_params(),
                 optimizer=dict(exp_avg=m, exp_avg_sq=v, step=step, lr=self.learning_rate, betas=self.adam_betas, eps=self.adam_eps),
                 hyper=self._hyper(), obs_dim=self.obs_dim, act_dim=self.act_dim,
                 net_arch=self.net_arch if explicit_arch else None,
                 counters=dict(num_timesteps=self.num_timesteps, _total_timesteps=self._total_timesteps,
                               _num_timesteps_at_start=self._num_timesteps_at_start, _n_updates=self._n_updates,
                               _episode_num=self._episode_num, start_time=self.start_time or time.time_ns(),
                               _current_progress_remaining=self._current_progress_remaining),
                 last_obs=self._last_obs, last_episode_starts=self._last_episode_starts,
                 ep_info_buffer=list(self.ep_info_buffer), verbose=self.verbose, seed=self.seed,
                 tensorboard_log=self.tensorboard_log,
                 obs_low=None if self.obs_bounds is None else self.obs_bounds[0],
                 obs_high=None if self.obs_bounds is None else self.obs_bounds[1],
                 extra_policy_kwargs={k: (list(v) if isinstance(v, tuple) else v) for k, v in self.policy_kwargs.items()
                                      if k in ("log_std_init", "ortho_init", "optimizer_kwargs", "share_features_extractor",
                                               "normalize_images", "full_std", "use_expln", "squash_output")}
                 | ({"activation_fn": self.activation} if self.activation != "tanh" else {}))

    @classmethod
    def load(cls, path, env=None, device="auto", custom_objects=None, print_system_info=False, force_reset=True,
             **kwargs):
        ck = load_zip(path)
        d, params = ck["data"], ck["params"]
        D = params["mlp_extractor.policy_net.0.weight"].shape[1]
        A = params["action_net.weight"].shape[0]   # (log_std is [HL, A] with use_sde)
        def widths(net):   # hidden widths from the state dict itself (nn.Sequential indices 0, 2, 4 ...)
            out, i = [], 0
            while f"mlp_extractor.{net}.{2 * i}.weight" in params:
                out.append(params[f"mlp_extractor.{net}.{2 * i}.weight"].shape[0])
                i += 1
            return tuple(out)
        pi, vf = widths("policy_net"), widths("value_net")
        pk = dict(d.get("policy_kwargs") or {})
        pk.setdefault("net_arch", dict(pi=list(pi), vf=list(vf)))
        explicit_arch = "net_arch" in (d.get("policy_kwargs") or {})
        n_envs = int(d.get("n_envs", 1)) if env is None else env.num_envs
        if env is not None and (env.obs_dim, env.act_dim) != (D, A):
            raise ValueError(f"Observation/action spaces do not match: checkpoint ({D},{A}) vs env ({env.obs_dim},{env.act_dim})")
        clip = d.get("clip_range")
        model = cls(policy="MlpPolicy", env=env, learning_rate=float(d.get("learning_rate", 3e-4)),
                    n_steps=int(d.get("n_steps", 2048)), batch_size=int(d.get("batch_size", 64)),
                    n_epochs=int(d.get("n_epochs", 10)), gamma=float(d.get("gamma", 0.99)),
                    gae_lambda=float(d.get("gae_lambda", 0.95)), clip_range=float(clip) if isinstance(clip, (int, float)) else 0.2,
                    clip_range_vf=d.get("clip_range_vf") if isinstance(d.get("clip_range_vf"), (int, float)) else None,
                    target_kl=d.get("target_kl"), use_sde=bool(d.get("use_sde", False)),
                    sde_sample_freq=int(d.get("sde_sample_freq", -1)),
                    normalize_advantage=bool(d.get("normalize_advantage", True)), ent_coef=float(d.get("ent_coef", 0.0)),
                    vf_coef=float(d.get("vf_coef", 0.5)), max_grad_norm=float(d.get("max_grad_norm", 0.5)),
                    tensorboard_log=d.get("tensorboard_log"), policy_kwargs=pk, verbose=int(d.get("verbose", 0)),
                    seed=d.get("seed"), device=device, _dims=(n_envs, D, A), **kwargs)
        if not explicit_arch:
            model.policy_kwargs.pop("net_arch", None)
        model.engine.set_params(params)
        opt = ck["optimizer"]
        if opt is not None and opt.get("exp_avg") is not None:
            model.engine.set_optimizer_state(opt["exp_avg"], opt["exp_avg_sq"], opt["step"])
        for k in ("num_timesteps", "_total_timesteps", "_num_timesteps_at_start", "_n_updates", "_episode_num",
                  "_current_progress_remaining", "start_time"):
            if d.get(k) is not None:
                setattr(model, k, d[k])
        if d.get("ep_info_buffer") is not None:
            model.ep_info_buffer = deque(d["ep_info_buffer"], maxlen=model._stats_window_size)
        sp = d.get("observation_space") or {}
        if model.obs_bounds is None and sp.get("low") is not None:
            model.obs_bounds = (np.asarray(sp["low"], np.float32), np.asarray(sp["high"], np.float32))
        if not force_reset and d.get("_last_obs") is not None:
            model._last_obs = np.asarray(d["_last_obs"], np.float32)
        return model


class PPOCtrl:
    """Same constructor, `from_config`, `learn`, `save_model` and `.ppo` attribute as the reference class
    (src/mobrob/rl_control/ppo.py:14-77).  `vec_env_type` accepts the reference values "subproc" and "dummy"
    (ValueError otherwise, ppo.py:35): "subproc" steps the environments in worker processes on all host cores
    (`ShmVecEnv`: shared GPU-visible block instead of pipes), "dummy" in the learner process (`HostVecEnv`) -- plus
    the build's extensions: "synthetic" (host NumPy env source), "native" (the goal task stepped by the multi-threaded C host environment,
    csrc/host_env.c), "device" (device-resident synthetic source) and "device_goal" (the goal task stepped on the GPU)."""

    def __init__(self, ppo_kwargs: dict, env_name: str, time_limit: int, n_env: int, vec_env_type: str = "dummy",
                 enable_gui: bool = False, seed: int = 0) -> None:
        self.ppo_kwargs = ppo_kwargs
        self.env_name = env_name
        self.time_limit = time_limit
        self.n_env = n_env
        from ..parallel import distributed_context
        _, rank, _ = distributed_context()
        env_seed = seed + 1000 * rank  # data-parallel ranks own different environments (the model seed stays shared)
        if vec_env_type in ("subproc", "dummy"):
            if env_name not in ROBOT_DIMS:
                raise ValueError(f"Env {env_name} not found")  # what get_env would raise inside a worker
            cls = DummyVecEnv
            if vec_env_type == "subproc":  # one rank of a data-parallel job gets its share of the usable host cores
                import functools
                from ..envs.shm_vec_env import usable_cores
                world, _, _ = distributed_context(init=False)
                # MOBROB_ENV_WORKERS overrides the one-worker-per-usable-core default; either way the
                # ranks of a data-parallel job on one node share the host cores
                workers = int(os.environ.get("MOBROB_ENV_WORKERS", 0) or usable_cores())
                cls = functools.partial(SubprocVecEnv, n_workers=max(1, workers // max(1, world)))
            vec_env = make_vec_env(get_env, n_envs=n_env,
                                   env_kwargs={"env_name": env_name, "enable_gui": enable_gui,
                                               "terminate_on_goal": True, "time_limit": time_limit},
                                   vec_env_cls=cls, seed=env_seed)
        elif vec_env_type == "synthetic":
            vec_env = SyntheticVecEnv.for_robot(env_name, n_env, time_limit, env_seed)
        elif vec_env_type == "device":
            vec_env = DeviceSyntheticVecEnv.for_robot(env_name, n_env, time_limit, seed)
        elif vec_env_type == "native":
            vec_env = NativeGoalVecEnv.for_robot(env_name, n_env, time_limit, env_seed, terminate_on_goal=True)
        elif vec_env_type == "device_goal":
            vec_env = DeviceGoalVecEnv.for_robot(env_name, n_env, time_limit, seed, terminate_on_goal=True)
        else:
            raise ValueError(f"Unknown vec_env_type: {vec_env_type}")
        self.ppo = PPO(env=vec_env, seed=seed,
                       tensorboard_log=(f"{DATA_DIR}/policies/tmp/{env_name}-ppo/tensorboard" if tensorboard is not None else None),
                       **ppo_kwargs)

    @classmethod
    def from_config(cls, config: dict) -> "PPOCtrl":
        return cls(ppo_kwargs=config["ppo_kwargs"], env_name=config["env_name"], time_limit=config["time_limit"],
                   n_env=config["n_envs"], vec_env_type=config["vec_env_type"], enable_gui=config["enable_gui"],
                   seed=config["seed"])

    def learn(self, *args, **kwargs) -> None:
        self.ppo.learn(*args, **kwargs)

    def save_model(self, save_path: str) -> None:
        self.ppo.save(save_path)
