/*
 * mobrob_ppo.h -- C ABI of libmobrob_ppo.so, the MI355X (gfx950) PPO training engine.
 *
 * Drop-in boundary for the ONE hot path of ZikangXiong/mobrob: everything that
 * `PPOCtrl.__init__ -> stable_baselines3.PPO(...)`  (reference src/mobrob/rl_control/ppo.py:50-59),
 * `PPOCtrl.learn -> PPO.learn`                       (reference src/mobrob/rl_control/ppo.py:73-74),
 * `PPOCtrl.save_model -> PPO.save`                   (reference src/mobrob/rl_control/ppo.py:76-77),
 * `load_policy -> PPO.load`                          (reference src/mobrob/utils.py:15-16) and
 * `policy.predict(obs, deterministic=True)`          (reference examples/control.py:39)
 * execute inside stable-baselines3 2.0.0 / torch-CPU (reference requirements.txt:9).
 * Each entry point below cites the reference call site / SB3 routine it replaces.
 *
 * Conventions
 *   - plain C types only; every pointer is a HOST pointer unless its name ends in `_dev`.
 *   - all arithmetic is IEEE float32 ("f32"); GAE carries its accumulator in f64 exactly like
 *     SB3's NumPy loop does (oracle/ppo_oracle.py:gae).
 *   - parameters travel as ONE flat f32 vector in SB3 `policy.state_dict()` order:
 *       log_std[A], pi.0.weight[H1,D], pi.0.bias[H1], pi.2.weight[H2,H1], pi.2.bias[H2],
 *       vf.0.weight[G1,D], vf.0.bias[G1], vf.2.weight[G2,G1], vf.2.bias[G2],
 *       action_net.weight[A,H2], action_net.bias[A], value_net.weight[1,G2], value_net.bias[1]
 *     (weights row-major [out,in]; verified against data/policies/<env>-ppo.zip:policy.pth).
 *     Other depths (one to eight hidden layers per network) continue nn.Sequential's numbering: pi.0, pi.2, pi.4 ... then vf.0 ...;
 *     with use_sde log_std is a matrix [HL,A] (HL = last policy width; [HL,1] without full_std), row-major, still first.
 *   - rollout storage is [T][N][...] on the device; minibatch indices are SB3's ENV-MAJOR flat
 *     index  flat = n*T + t  (SB3 RolloutBuffer.swap_and_flatten).
 *   - every function returns MOBROB_OK (0) or a negative error code; mobrob_ppo_last_error()
 *     returns a thread-local description.  Nothing here ever falls back to a CPU implementation.
 *   - ownership: the caller owns host buffers (pin them with mobrob_ppo_host_alloc for truly
 *     asynchronous H2D/D2H); the engine owns all device memory.
 */
#ifndef MOBROB_PPO_H
#define MOBROB_PPO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: config slot `persistent_train` became `activation`, `forward_x3` added, MOBROB_K_COUNT 6 -> 7 (profile_read arrays),
 *    MOBROB_BUF_COUNT / reserved[] resized -- a binding built against 1 must not load this library
 * 3: pi_hidden_ext / vf_hidden_ext (net_arch depths 4 .. 8), use_sde, sde_sample_freq appended to the config, activation codes 2 .. 11 */
#define MOBROB_PPO_ABI_VERSION 3
/* policy_kwargs.activation_fn (SB3 ActorCriticPolicy; the reference splats ppo_kwargs into PPO verbatim, ppo.py:58): the
 * parameter-free element-wise torch.nn modules with torch's default arguments (ELU alpha 1, LeakyReLU slope 0.01, Softplus beta 1 /
 * threshold 20, Hardtanh [-1, 1], GELU approximate='none') */
enum { MOBROB_ACT_TANH = 0, MOBROB_ACT_RELU = 1, MOBROB_ACT_ELU = 2, MOBROB_ACT_LEAKY_RELU = 3, MOBROB_ACT_SIGMOID = 4,
       MOBROB_ACT_SOFTPLUS = 5, MOBROB_ACT_SOFTSIGN = 6, MOBROB_ACT_HARDTANH = 7, MOBROB_ACT_RELU6 = 8, MOBROB_ACT_SILU = 9,
       MOBROB_ACT_GELU = 10, MOBROB_ACT_MISH = 11, MOBROB_ACT_COUNT = 12 };

enum {
  MOBROB_OK = 0,
  MOBROB_ERR_INVALID = -1,     /* bad argument / unsupported shape (-> ValueError)            */
  MOBROB_ERR_HIP = -2,         /* a HIP runtime call failed (-> RuntimeError)                  */
  MOBROB_ERR_STATE = -3,       /* call sequence violated, e.g. train before finish_rollout     */
  MOBROB_ERR_NO_DEVICE = -4    /* no gfx950 device visible                                     */
};

typedef struct mobrob_ppo_engine mobrob_ppo_engine_t;

/* Hyper-parameters = the kwargs the reference splats into PPO(...) (ppo.py:58; data/configs/
 * <env>-ppo.yaml `ppo_kwargs`) plus SB3 2.0.0 defaults for the rest. */
typedef struct mobrob_ppo_config {
  int32_t abi_version;        /* = MOBROB_PPO_ABI_VERSION                                        */
  int32_t obs_dim;            /* D: 14 point, 26 car, 58 doggo, 12 drone, 43 turtlebot3          */
  int32_t act_dim;            /* A:  2,        2,      12,       18,       2                     */
  int32_t pi_hidden[2];       /* policy_kwargs.net_arch.pi  (default [64,64]); one hidden layer: [h, 0]; a third: pi_hidden3 below */
  int32_t vf_hidden[2];       /* policy_kwargs.net_arch.vf  (default [64,64]); likewise with vf_hidden3                         */
  int32_t n_envs;             /* N: vectorised envs owned by THIS rank (yaml n_envs)             */
  int32_t n_steps;            /* T: rollout horizon (ppo_kwargs.n_steps, default 2048)           */
  int32_t batch_size;         /* GLOBAL minibatch size (ppo_kwargs.batch_size, default 64)       */
  int32_t n_epochs;           /* ppo_kwargs.n_epochs (default 10)                                */
  /* python floats are doubles: the engine rounds to f32 exactly where SB3/torch would */
  double gamma;               /* default 0.99                                                    */
  double gae_lambda;          /* default 0.95                                                    */
  double clip_range;          /* constant schedule, default 0.2                                  */
  double ent_coef;            /* default 0.0                                                     */
  double vf_coef;             /* default 0.5                                                     */
  double max_grad_norm;       /* default 0.5                                                     */
  double learning_rate;       /* constant schedule, default 3e-4                                 */
  double adam_beta1, adam_beta2, adam_eps; /* 0.9, 0.999, 1e-5 (SB3 passes eps=1e-5 to Adam)     */
  double action_low, action_high; /* Box bounds used for the clipped action copy (-1, 1)         */
  int32_t normalize_advantage;/* default 1                                                       */
  uint64_t seed;              /* Philox key for eps / synthetic env / Feistel permutations       */
  int32_t device_id;          /* HIP device ordinal                                              */
  int32_t rank, world_size;   /* data-parallel position; batch_size is split batch_size/world    */
  int32_t fast_kernels;       /* 1: use the fused MFMA kernels when the shape allows; 0: generic */
  int32_t rollout_graph;      /* 1: the per-step device rollout (2 launches per step) is replayed as one
                                 captured hipGraph; the persistent rollout needs no graph          */
  int32_t rollout_persistent; /* 1: run the device-resident rollout as one persistent kernel (fused widths) */
  int32_t activation;         /* hidden activation of both networks: MOBROB_ACT_TANH (0; SB3's default for MlpPolicy, every
                                 reference YAML) or another MOBROB_ACT_* (policy_kwargs activation_fn; generic GEMM chain) */
  int32_t forward_x3;         /* 1 (default): the hidden-layer matrix products of 256-wide tanh nets -- rollout policy forward, batched value
                                 pass, and inside the gradient kernel the forward, dh1, dW2, dW1 -- run on the bf16 matrix pipe with every
                                 float32 operand split into three bf16 pieces and six piece products kept, float32 accumulation: float32
                                 RESULTS (error against float64 not larger than v_mfma_f32's, not bit-equal to it) at up to
                                 16/6 of the f32 matrix rate.  0: v_mfma_f32 everywhere.  Heads, loss, GAE, clip and Adam are plain float32
                                 either way; so are the 64-wide kernel families and the generic GEMM chain. */
  int32_t pi_hidden3;         /* width of a THIRD policy hidden layer (0: none).  net_arch depths 1 .. 8 are accepted per network (SB3 takes any
                                 list; the reference's YAMLs use two layers); depths other than two run the generic GEMM chain */
  int32_t vf_hidden3;         /* likewise for the value network */
  int32_t reserved[1];
  int32_t pi_hidden_ext[5];   /* widths of policy hidden layers 4 .. 8 (a width of 0 ends the list) */
  int32_t vf_hidden_ext[5];   /* likewise for the value network */
  int32_t use_sde;            /* PPO(use_sde=True): generalised state-dependent exploration (SB3 StateDependentNoiseDistribution with its
                                 defaults; full_std / use_expln below; no squashing, learn_features=False).  log_std becomes a [HL][A] matrix (HL =
                                 width of the last policy hidden layer); generic GEMM chain.  Default 0 */
  int32_t sde_sample_freq;    /* PPO(sde_sample_freq): new exploration matrices every this many rollout steps (-1: only at the start of a
                                 rollout, SB3's default) */
  int32_t sde_full_std;       /* policy_kwargs full_std (default 1): 0 = one standard deviation per latent unit, log_std is [HL][1] */
  int32_t sde_use_expln;      /* policy_kwargs use_expln (default 0): std = exp(ls) for ls <= 0, log1p(ls + 1e-6) + 1 above       */
} mobrob_ppo_config_t;

/* Fill `cfg` with SB3 2.0.0 defaults (Appendix A.1).  Replaces PPO.__init__'s default kwargs. */
void mobrob_ppo_default_config(mobrob_ppo_config_t* cfg);

/* PPO(...)/_setup_model (ppo.py:50-59): allocate rollout buffer, policy, Adam(eps) on the device;
 * parameters are initialised to zero -- the caller uploads weights (set_params). */
int mobrob_ppo_create(const mobrob_ppo_config_t* cfg, mobrob_ppo_engine_t** out);
void mobrob_ppo_destroy(mobrob_ppo_engine_t* e);

/* Mixed fleet (several robot types = several PPO(...) objects with different obs/act dims trained side by
 * side, the loop of examples/train.py:42-46 run once per env name): every device buffer of an engine is
 * carved out of ONE arena, so the ragged segments of a fleet pack back to back into one rollout allocation,
 * each with its own (obs_dim, act_dim) strides.  device_bytes is a host-only sizing pass (no device needed);
 * create_in_arena builds the engine inside caller-owned device memory (256-byte aligned, >= device_bytes),
 * which must outlive the engine.  mobrob_ppo_create == device_bytes + one private arena. */
int mobrob_ppo_device_bytes(const mobrob_ppo_config_t* cfg, size_t* bytes);
int mobrob_ppo_create_in_arena(const mobrob_ppo_config_t* cfg, void* arena, size_t arena_bytes,
                               mobrob_ppo_engine_t** out);
void* mobrob_ppo_device_alloc(int32_t device_id, size_t bytes);
void mobrob_ppo_device_free(void* p);
const char* mobrob_ppo_last_error(void);
int mobrob_ppo_abi_version(void);

/* Use an externally owned hipStream_t (e.g. torch's current stream, so that RCCL collectives
 * issued through torch.distributed are ordered with the engine's kernels).  NULL -> own stream. */
int mobrob_ppo_set_stream(mobrob_ppo_engine_t* e, void* hip_stream);
int mobrob_ppo_synchronize(mobrob_ppo_engine_t* e);

/* Pinned host memory for the rollout streamer (hipHostMalloc / hipHostFree). */
void* mobrob_ppo_host_alloc(size_t bytes);
void mobrob_ppo_host_free(void* p);
/* Pin and map caller-owned host memory in place (hipHostRegister, mapped + portable): afterwards the range is
 * accepted wherever host_alloc memory is (zero-copy act/store, act_part/store_part, collect_host).  Meant for the
 * POSIX shared-memory block that environment worker PROCESSES write their step results into -- the replacement of
 * the pipes between SubprocVecEnv workers and the learner (/root/reference/src/mobrob/rl_control/ppo.py:30-33 with
 * make_vec_env :37-48): `p` must stay mapped until host_unregister.  Fails (MOBROB_ERR_HIP) if the range cannot be
 * pinned or is not device visible at its host address. */
int mobrob_ppo_host_register(void* p, size_t bytes);
int mobrob_ppo_host_unregister(void* p);

/* policy.state_dict() / load_state_dict() (examples/train.py:31-33) and the optimizer state of
 * policy.optimizer.pth.  n must equal mobrob_ppo_param_count(). */
int64_t mobrob_ppo_param_count(const mobrob_ppo_engine_t* e);
int mobrob_ppo_get_params(mobrob_ppo_engine_t* e, float* out, int64_t n);
int mobrob_ppo_set_params(mobrob_ppo_engine_t* e, const float* in, int64_t n);
int mobrob_ppo_get_optimizer_state(mobrob_ppo_engine_t* e, float* exp_avg, float* exp_avg_sq, int64_t n,
                                   int64_t* step);
int mobrob_ppo_set_optimizer_state(mobrob_ppo_engine_t* e, const float* exp_avg, const float* exp_avg_sq,
                                   int64_t n, int64_t step);

/* ---- rollout: OnPolicyAlgorithm.collect_rollouts (SB3; reached via ppo.py:73-74) ------------- */

/* rollout_buffer.reset(): position -> 0. */
int mobrob_ppo_rollout_begin(mobrob_ppo_engine_t* e);

/* `actions, values, log_probs = policy(obs)` + np.clip for the env.  obs[N*D] is streamed H2D into
 * rollout slot t; eps[N*A] ~ N(0,I) may be NULL -> drawn on device (Philox4x32-10 + Box-Muller).
 * Outputs (any may be NULL): raw actions (what the buffer stores), clipped actions (what the env
 * sees), values, log_probs.  Blocks until the outputs are on the host. */
int mobrob_ppo_act(mobrob_ppo_engine_t* e, const float* obs, const float* eps, float* actions_raw,
                   float* actions_clipped, float* values, float* log_probs);

/* rollout_buffer.add(...) for the step just acted: rewards[N]; dones[N] (terminated|truncated,
 * becomes the NEXT step's episode_start); truncated[N] = infos["TimeLimit.truncated"] (may be
 * NULL); terminal_obs[N*D] = infos["terminal_observation"] rows (only rows with truncated!=0 are
 * read; may be NULL when nothing was truncated).  Applies rewards += gamma * V(terminal_obs). */
int mobrob_ppo_store(mobrob_ppo_engine_t* e, const float* rewards, const uint8_t* dones,
                     const uint8_t* truncated, const float* terminal_obs);

/* ---- pipelined host-environment rollout -------------------------------------------------------------------
 * Same results as act()/store() over all N rows, but the rows are cut into `nparts` contiguous ranges
 * [N*part/nparts, N*(part+1)/nparts) so that the host simulator steps one range while the GPU runs the policy for
 * another (the reference alternates strictly: SubprocVecEnv.step_async/step_wait after policy.forward,
 * ppo.py:30-33 + SB3 collect_rollouts).  Every pointer is the FULL [N][...] array in device-visible pinned memory
 * (mobrob_ppo_host_alloc); a call touches only the rows of its part.
 *   act_part    enqueue: pull the part's obs rows -> policy/value forward + sample -> push its clipped actions;
 *               returns without waiting
 *   wait_part   block until the clipped actions of the part's latest act_part are in actions_clipped
 *   store_part  enqueue rollout_buffer.add for the part (rewards, episode starts, time-limit bootstrap); with
 *               next_obs != NULL the same launch also pulls the part's next observations into slot t+1, so the
 *               following act_part launches the policy kernel only
 * Each part keeps its own step index (part p may act on step t+1 before part q has stored step t);
 * finish_rollout() requires every part to have stored n_steps steps.  nparts is fixed between rollout_begin()
 * and finish_rollout(), 1 <= nparts <= MOBROB_MAX_PARTS. */
#define MOBROB_MAX_PARTS 8
int mobrob_ppo_act_part(mobrob_ppo_engine_t* e, int32_t part, int32_t nparts, const float* obs,
                        float* actions_clipped);
int mobrob_ppo_wait_part(mobrob_ppo_engine_t* e, int32_t part);
int mobrob_ppo_store_part(mobrob_ppo_engine_t* e, int32_t part, int32_t nparts, const float* rewards,
                          const uint8_t* dones, const uint8_t* truncated, const float* terminal_obs,
                          const float* next_obs);

/* The whole pipelined rollout as ONE native call -- SB3's collect_rollouts loop (on_policy_algorithm.py, reached via
 * ppo.py:73-74) without a Python frame per step: rollout_begin, n_steps x nparts x (wait_part, step_range of the
 * part's envs, store_part + act_part), finish_rollout.  `step_range(env, i0, i1, actions, obs, rewards, dones,
 * truncated, terminal_obs)` steps the envs [i0, i1) of a vectorised environment over the FULL pinned arrays (VecEnv
 * semantics: auto-reset, terminal observation of truncated rows) and returns how many of them were truncated (< 0:
 * error).  csrc/host_env.c's mobrob_hostenv_step_range has exactly this signature.  `obs` must hold the current
 * observations (VecEnv.reset() or the previous rollout's last step) on entry and holds the last ones on return.
 *
 * SERVED form (the default where it applies): the step loop contains NO launch, event or other HIP call -- the persistent rollout kernel
 * of the device environments serves the host one (csrc/kernels_rollout.h KIND 3: k_rollout_persistent<.., 3, S8> for 256-wide engines
 * with the split-bf16 forward, k_rollout64_tile<.., 3> for the 64-wide networks of the reference YAMLs).  A workgroup writes its rows'
 * clipped actions into `actions_clipped`, raises a flag word in pinned memory and polls the host's word for its row range; this function
 * waits for the flags of a range, calls step_range on it and raises the range's word; V(obs) comes from the batched value pass.
 * Conditions: a fused engine of one of the two widths; row ranges of whole 32-row tiles (n_envs % (32 nparts) == 0) or ONE range of any
 * size (nparts == 1: the reference's 2 - 16 environments); at most one tile per compute unit; all six buffers in COHERENT pinned host
 * memory over their whole length (mobrob_ppo_host_alloc, or mobrob_ppo_host_register while HIP_HOST_COHERENT is not 0; a non-coherent
 * allocation is refused by name); no announced co-tenant of the device (MOBROB_DP_SAME_DEVICE ranks, a CU mask); every workgroup
 * resident within MOBROB_SERVER_RESIDENCY_S (default 2 s) -- otherwise, and with MOBROB_COLLECT_SERVER=0, the launch-per-step form runs
 * (=2: fail instead, naming the reason).  256-wide: numbers agree with the launch-per-step form to float32 rounding (policy forward on
 * the other matrix pipe); 64-wide: bit for bit.  Every wait on either side is bounded (MOBROB_SERVER_TIMEOUT_S, default 60: the call
 * fails, the queued launches return at once). */
typedef int32_t (*mobrob_env_step_range_fn)(void* env, int32_t i0, int32_t i1, const float* actions, float* obs,
                                            float* rewards, uint8_t* dones, uint8_t* truncated, float* terminal_obs);
int mobrob_ppo_collect_host(mobrob_ppo_engine_t* e, mobrob_env_step_range_fn step_range, void* env, int32_t nparts,
                            float* obs, float* actions_clipped, float* rewards, uint8_t* dones, uint8_t* truncated,
                            float* terminal_obs);

/* `values = policy.predict_values(new_obs)` + RolloutBuffer.compute_returns_and_advantage
 * (GAE(lambda) reverse scan).  last_obs[N*D], dones[N] = dones of the final step. */
int mobrob_ppo_finish_rollout(mobrob_ppo_engine_t* e, const float* last_obs, const uint8_t* dones);

/* Device-resident synthetic env source (SURVEY.md §8d / BASELINE.md §3): the whole T-step rollout
 * (env draw -> act -> store -> bootstrap) runs on the GPU without host round trips, then GAE.
 * obs ~ N(0,1), reward ~ N(0.03,0.1^2) + 5*terminated, terminated ~ Bernoulli(p_term),
 * truncation at time_limit with a terminal observation.  State persists across calls. */
int mobrob_ppo_collect_synthetic(mobrob_ppo_engine_t* e, float p_term, int32_t time_limit);

/* Device-resident GOAL environment: the reference's env-side rules evaluated on the GPU inside the rollout
 * loop -- reward_fn (src/mobrob/envs/wrapper.py:137-154, drone bonus :491-496), step/terminate_on_goal
 * (:156-171), lazy reset + new goal (:173-201), reached (:203-207), TimeLimit truncation as get_env wraps it
 * (:549-571), VecEnv auto-reset with terminal observation and the time-limit bootstrap.  The robot itself is
 * the kinematic stand-in of mobrob_amd/envs/wrapper.py (the reference's MuJoCo / Bullet physics is out of
 * scope): vel <- 0.8 vel + 0.2 mix.clip(a); pos <- clip(pos + dt vel, +-extent); observation =
 * [unit vector to the goal, vel, pos, N(0, obs_noise^2) padding].  init_space = +-extent/2, goal_space =
 * +-extent.  State persists across calls; switching between env kinds restarts the environments. */
typedef struct mobrob_goal_env {
  int32_t pos_dim;            /* 1..3, 3*pos_dim <= obs_dim                                              */
  int32_t terminate_on_goal;  /* PPOCtrl passes True (ppo.py:37-48)                                      */
  int32_t time_limit;         /* max_episode_steps                                                       */
  float dt, extent;
  float reach_radius;         /* 0.3 in the reference                                                    */
  float goal_bonus;           /* +5 on reach                                                             */
  float extra_bonus;          /* +10 more for the drone                                                  */
  float obs_noise;            /* std of the padding features                                             */
  float mix[3][32];           /* [pos_dim][act_dim] action -> velocity command read-out                  */
} mobrob_goal_env_t;
int mobrob_ppo_collect_goal_env(mobrob_ppo_engine_t* e, const mobrob_goal_env_t* env);

/* Monitor-style statistics of the episodes the goal environment finished since the last reset of the
 * counters (SB3's rollout/ep_rew_mean, ep_len_mean).  Waits for the engine's stream. */
typedef struct mobrob_episode_stats {
  int64_t episodes;
  double return_sum, length_sum;
  int64_t goals;              /* episodes that ended inside the reach radius */
} mobrob_episode_stats_t;
int mobrob_ppo_episode_stats(mobrob_ppo_engine_t* e, mobrob_episode_stats_t* out, int32_t reset);
/* The Monitor records SB3 keeps in `ep_info_buffer` (what `rollout/ep_rew_mean` averages and PPO.save stores):
 * (return, length) of the episodes the device goal environment finished since the previous call, oldest first,
 * newest `max_records` at most (the device keeps the last 128).  out = [max_records][2] floats; returns the count. */
int mobrob_ppo_episode_records(mobrob_ppo_engine_t* e, float* out, int32_t max_records);

/* ---- update: PPO.train (SB3 ppo/ppo.py) ----------------------------------------------------- */

typedef struct mobrob_ppo_train_stats {
  /* means over the minibatches of the LAST epoch run by the call (SB3 logs the same way) */
  float policy_loss, value_loss, entropy_loss, loss, approx_kl, clip_fraction, grad_norm;
  int32_t n_minibatches; /* optimizer steps taken by the call */
} mobrob_ppo_train_stats_t;

/* ---- data-parallel update: one process per GPU, rank r owns its envs and rollout shard ------------
 * PPO.train() of the reference (reached through PPOCtrl.learn, /root/reference/src/mobrob/rl_control/ppo.py:73-74)
 * across `cfg.world_size` ranks: per epoch ONE all-reduce of the [n_minibatches][4] float64 advantage statistics, per
 * optimizer step ONE all-reduce (sum) of the flat [P + 8] float32 gradient + loss sums, both enqueued on the engine's stream between
 * the kernels -- no host synchronisation and no interpreter in the loop.  Every rank then applies the identical
 * clip + Adam, so replicas stay bit-identical.  batch_size in the config is the GLOBAL minibatch.
 *   comm_unique_id  rank 0 makes the 128-byte RCCL id; the caller ships it to the other ranks (any side channel)
 *   comm_init       collective: ncclCommInitRank(world_size, id, rank) -> the engine's communicator (RCCL over xGMI)
 *   train_dp        the loop.  fn == NULL: RCCL on the communicator.  fn != NULL: the caller's all-reduce, called with
 *                   (ctx, device pointer, element count, dtype 0 = f32 / 1 = f64, hipStream_t of the engine); it must
 *                   leave the SUM over ranks in place, ordered after prior work on that stream, and return 0
 *                   (tests: gloo between two ranks that share one GPU). */
typedef int (*mobrob_allreduce_fn)(void* ctx, void* buf_dev, size_t count, int32_t dtype, void* hip_stream);
int mobrob_ppo_comm_unique_id(uint8_t* out128);
/* comm_prepare  LOCAL, non-collective half of comm_init: RCCL loadable, device selectable, no communicator yet.  Ranks
 *               agree on its outcome (any side channel) BEFORE any of them enters the blocking ncclCommInitRank, so a
 *               rank that cannot take part never leaves the others waiting for it.
 * comm_init_rank  comm_init with the rank / size of the process (sub)group the communicator spans; nranks must equal
 *               cfg.world_size (the minibatch split).  comm_init == comm_init_rank(cfg.rank, cfg.world_size).
 * comm_info     ncclCommCount / ncclCommUserRank of the engine's communicator (0 / -1 without one): what a bench line
 *               reports as the number of ranks that really took part in the collectives. */
int mobrob_ppo_comm_prepare(mobrob_ppo_engine_t* e);
int mobrob_ppo_comm_init(mobrob_ppo_engine_t* e, const uint8_t* id128);
int mobrob_ppo_comm_init_rank(mobrob_ppo_engine_t* e, const uint8_t* id128, int32_t rank, int32_t nranks);
int mobrob_ppo_comm_info(mobrob_ppo_engine_t* e, int32_t* nranks, int32_t* rank);
int mobrob_ppo_comm_destroy(mobrob_ppo_engine_t* e);
/* SB3's target_kl works under data parallel too: the minibatch's approx_kl sum travels with the gradient (the message
 * is [P + 8] floats: gradient + loss sums), every rank reads the same global value and stops at the same step. */
int mobrob_ppo_train_dp(mobrob_ppo_engine_t* e, const int64_t* perms, mobrob_allreduce_fn fn, void* ctx);
/* ---- one-shot all-reduce over peer-mapped memory (opt-in; RCCL stays the default) ------------------------------------
 * The gradient message of a PPO step is 645 KB: latency-bound on the xGMI mesh, where a direct exchange (every rank
 * reads every peer's contribution and adds them up itself, IN RANK ORDER -> the same bits on every rank, deterministic
 * run to run) beats a ring.  Every rank exports an exchange buffer with hipIpcGetMemHandle; the handles travel over any
 * side channel; every rank opens the others' (hipIpcOpenMemHandle: over xGMI between GPUs, the same physical memory for
 * two ranks that share a device).  Afterwards train_dp(fn == NULL) exchanges through it instead of RCCL: one launch per
 * message, per-16-KB-chunk sequence flags with system-scope release / acquire (csrc/oneshot_allreduce.h).  A peer that
 * never publishes raises an error at the next synchronising call (MOBROB_ONESHOT_TIMEOUT_MS, default 20 s) instead of
 * hanging the device.
 *   oneshot_export  allocate the exchange buffer (once) and write its IPC handle (MOBROB_IPC_HANDLE_BYTES bytes)
 *   oneshot_open    handles = [nranks][MOBROB_IPC_HANDLE_BYTES] in rank order (the own entry is ignored)
 *   oneshot_close   unmap the peers, free the buffer (also done by destroy) */
#define MOBROB_IPC_HANDLE_BYTES 64
int mobrob_ppo_oneshot_export(mobrob_ppo_engine_t* e, uint8_t* handle64);
int mobrob_ppo_oneshot_open(mobrob_ppo_engine_t* e, const uint8_t* handles, int32_t rank, int32_t nranks);
int mobrob_ppo_oneshot_close(mobrob_ppo_engine_t* e);
/* Known vectors through the exchange train_dp would use, compared with the rank-ordered sums: run at communicator set-up (collective),
 * before any gradient depends on the exchange.  which: 0 = the engine's RCCL communicator, 1 = the one-shot exchange.  Four
 * messages: the [P + 8]-float gradient message twice and the [n_minibatches][4]-double advantage message twice (both payload slots
 * of the one-shot exchange, slot reuse, both element types), on a scratch buffer of the call's own; MOBROB_ERR_STATE while an
 * epoch is open or a gradient awaits its apply.
 * *mismatches = elements over the four messages that are not bit-equal to the expected sum; 0 = sound.  The reference has one
 * exchange path and trusts it (SubprocVecEnv pipes, /root/reference/src/mobrob/rl_control/ppo.py:30-33); here a peer-mapped
 * exchange is used on real peers only after it has passed, RCCL otherwise (mobrob_amd/parallel.py). */
int mobrob_ppo_exchange_selfcheck(mobrob_ppo_engine_t* e, int32_t which, int32_t* mismatches);
/* all-reduces issued by train_dp since the last reset: how many, and their payload bytes (bench: allreduces_per_step) */
int mobrob_ppo_allreduce_counters(mobrob_ppo_engine_t* e, int64_t* calls, int64_t* bytes, int32_t reset);

/* ---- env-side controllers of the Bullet robots, batched over n robots on the device (csrc/robot_ctrl.h) ----------
 * In the reference the RL action of these two robots corrects controller GAINS and the controller runs inside
 * env.step on the host, one robot at a time.  dev_ptrs != 0: every array pointer is a DEVICE pointer and the call
 * only enqueues the kernel on the engine's stream (a device-resident simulator calls it between physics steps);
 * dev_ptrs == 0: host arrays, copied in and out, the call returns when the results are in place.
 *   turtlebot3  `Turtlebot3.prop_ctrl` (robots/turtlebot3.py:214-238, from Turtlebot3Env.step, envs/wrapper.py:540-546):
 *               pos[n][2], theta[n], goal[n][2], gain_changes[n][2] (the action) -> twist[n][2] = (v, w)
 *   drone       `DronePIDController.control` with `finetune_*_pid_coef` (robots/drone.py:58-159, 175-193, from
 *               DroneEnv.step, envs/wrapper.py:481-489): pos[n][3], rpy[n][3], goal[n][3], action[n][18] (6 x 3 gain
 *               corrections: force P I D, torque P I D), ctrl_state[n][12] in/out (last position error, its integral,
 *               last attitude error, its integral) -> out[n][4] = thrust, torque x y z.  The rotor mixing
 *               (`_compute_rpm`) belongs to Bullet's actuator model and is not part of it. */
typedef struct mobrob_drone_params {
  float mass, g, dt;                                   /* kg, m/s^2, controller period (world.timestep = 1/50) */
  float max_thrust, max_xy_torque, max_z_torque;       /* actuator limits (drone.py:260-267) */
  float max_roll_pitch;                                /* attitude limit, pi/6 in the reference (drone.py:50) */
  float tune_fac;                                      /* gain radius = tune_fac * default gain, 0.3 (drone.py:29) */
} mobrob_drone_params_t;
int mobrob_ctrl_turtlebot3(mobrob_ppo_engine_t* e, int32_t n, int32_t dev_ptrs, const float* pos, const float* theta,
                           const float* goal, const float* gain_changes, float* twist);
int mobrob_ctrl_drone_pid(mobrob_ppo_engine_t* e, int32_t n, int32_t dev_ptrs, const mobrob_drone_params_t* prm,
                          const float* pos, const float* rpy, const float* goal, const float* action, float* ctrl_state,
                          float* out);

/* Hyper-parameters that SB3 lets change or that the reference YAMLs never set, without growing the config struct.
 * `ppo_kwargs` are splatted into stable_baselines3.PPO verbatim (/root/reference/src/mobrob/rl_control/ppo.py:58, README.md:49):
 *   LEARNING_RATE / CLIP_RANGE  the value of a schedule for the coming PPO.train() (SB3 evaluates callables of
 *                               `progress_remaining` once per train())
 *   CLIP_RANGE_VF               value-function clipping: the loss uses old_value + clamp(value - old_value, +-c);
 *                               negative = None (default)
 *   TARGET_KL                   early stop: when a minibatch's approx_kl > 1.5 * target, its optimizer step and the rest
 *                               of train() are dropped (train and train_dp alike); <= 0 = None (default)
 *   ENT_COEF / VF_COEF          loss coefficients */
enum {
  MOBROB_HYPER_LEARNING_RATE = 0, MOBROB_HYPER_CLIP_RANGE = 1, MOBROB_HYPER_CLIP_RANGE_VF = 2, MOBROB_HYPER_TARGET_KL = 3,
  MOBROB_HYPER_ENT_COEF = 4, MOBROB_HYPER_VF_COEF = 5,
  MOBROB_HYPER_EPOCH_KERNEL = 6   /* not an SB3 keyword: 0 keeps three launches per optimizer step where mobrob_ppo_train would run an
                                     epoch as one co-operative launch (csrc/kernels_epoch64.h; the env MOBROB_EPOCH_KERNEL=0 does the
                                     same for every engine of the process).  Engines that update CONCURRENTLY on one device (a fleet's
                                     segments) set 0: co-resident spinning launches must fit the device together. */
};
int mobrob_ppo_set_hyper(mobrob_ppo_engine_t* e, int32_t which, double value);
/* Of the latest mobrob_ppo_train / train_enqueue: epochs started (SB3's `_n_updates` increment), whether target_kl
 * stopped it, optimizer steps applied. */
int mobrob_ppo_last_train_info(const mobrob_ppo_engine_t* e, int32_t* epochs_started, int32_t* stopped_early,
                               int32_t* steps_applied);

/* Whole PPO.train(): n_epochs x ceil(T*N / batch) optimizer steps.  perms = n_epochs concatenated
 * env-major permutations of range(T*N) (what np.random.permutation would have produced), or NULL ->
 * counter-based Feistel permutations keyed by (seed, rank, update counter).  world_size must be 1. */
int mobrob_ppo_train(mobrob_ppo_engine_t* e, const int64_t* perms, mobrob_ppo_train_stats_t* stats);
/* Same update, enqueued on the engine's stream without waiting for it (mobrob_ppo_train == train_enqueue +
 * statistics read-back).  Lets the engines of a mixed fleet overlap their updates on one GPU; `perms`, when
 * given, must stay valid until mobrob_ppo_synchronize. */
int mobrob_ppo_train_enqueue(mobrob_ppo_engine_t* e, const int64_t* perms);

/* The same update split at the two points where data-parallel ranks exchange data:
 *   epoch_begin   -> local (sum adv, sum adv^2, count) per minibatch into advstat_dev
 *   [all-reduce advstat_dev, 3 doubles per minibatch]
 *   minibatch_grad(mb) -> local gradient of the GLOBAL-mean loss into grad_dev (P floats)
 *   [all-reduce grad_dev (sum)]
 *   minibatch_apply -> clip_grad_norm_ + Adam.step on every rank (replicas stay identical)
 * `perm` (also every row of `perms` of mobrob_ppo_train / _train_dp / _train_enqueue) must be a PERMUTATION of [0, T*N) in SB3's env-major
 * flat order (index = n * T + t): it is validated on the host -- every index in range, none twice -- and MOBROB_ERR_INVALID is returned
 * before any kernel scatters through it.  NULL: the engine draws its own (a keyed Feistel permutation on the device). */
int mobrob_ppo_epoch_begin(mobrob_ppo_engine_t* e, const int64_t* perm /* T*N or NULL */);
int mobrob_ppo_num_minibatches(const mobrob_ppo_engine_t* e);
int mobrob_ppo_minibatch_grad(mobrob_ppo_engine_t* e, int32_t mb);
int mobrob_ppo_minibatch_apply(mobrob_ppo_engine_t* e);
/* minibatch_apply with SB3's target_kl check in front of the optimizer step (the approx_kl of the -- all-reduced --
 * loss sums is read back, 4 bytes): *stopped = 1 means the step was NOT taken and the driver must end its train().
 * minibatch_apply itself refuses (MOBROB_ERR_STATE) while target_kl is set, so that no step-wise driver ignores it. */
int mobrob_ppo_minibatch_apply_checked(mobrob_ppo_engine_t* e, int32_t* stopped);
/* per-minibatch stats of every optimizer step since the last call to this function:
 * rows of 8 floats [policy_loss, value_loss, entropy_loss, loss, approx_kl, clip_fraction,
 * grad_norm, 0]; returns rows written (<= max_rows) or a negative error. */
int mobrob_ppo_fetch_step_stats(mobrob_ppo_engine_t* e, float* out, int32_t max_rows);

/* ---- inference: policy.predict (examples/control.py:39) -------------------------------------- */
/* (use_sde engines, deterministic = 0: `eps` is not used -- the noise is latent . exploration matrix, SB3's get_noise: the
 *  environments' own matrices when n == n_envs, the single exploration_mat otherwise) */
int mobrob_ppo_predict(mobrob_ppo_engine_t* e, const float* obs, int32_t n, int32_t deterministic,
                       const float* eps /* n*A or NULL */, float* actions_clipped, float* values);

/* ---- gSDE (use_sde = 1) ------------------------------------------------------------------------
 * policy.reset_noise(n_envs) (SB3 ActorCriticPolicy.reset_noise -> sample_weights): new exploration matrices for every environment
 * and the single matrix predict() uses for batches of another size, from the CURRENT log_std.  The rollout collectors call it
 * themselves at the start of a rollout and every sde_sample_freq steps (OnPolicyAlgorithm.collect_rollouts). */
int mobrob_ppo_sde_reset_noise(mobrob_ppo_engine_t* e);
/* The exploration matrices as an INPUT (tests, the oracle; like `eps` of mobrob_ppo_act): z = standard normals [n_envs][HL][A] ->
 * matrices z * exp(log_std), kept until the next call -- the collectors stop resampling on their own.  z = NULL: back to the
 * engine's own draws. */
int mobrob_ppo_sde_set_noise(mobrob_ppo_engine_t* e, const float* z);

/* ---- device buffers (tests, DP collectives, profiling) --------------------------------------- */
enum {
  MOBROB_BUF_OBS = 0,        /* f32 [T+1][N][Dp]  (Dp = D rounded up to 8; slot T = last_obs)     */
  MOBROB_BUF_ACTIONS = 1,    /* f32 [T][N][A]                                                    */
  MOBROB_BUF_REWARDS = 2,    /* f32 [T][N]                                                       */
  MOBROB_BUF_EPISODE_STARTS = 3, /* f32 [T][N]                                                   */
  MOBROB_BUF_VALUES = 4,     /* f32 [T][N]                                                       */
  MOBROB_BUF_LOG_PROBS = 5,  /* f32 [T][N]                                                       */
  MOBROB_BUF_ADVANTAGES = 6, /* f32 [T][N]                                                       */
  MOBROB_BUF_RETURNS = 7,    /* f32 [T][N]                                                       */
  MOBROB_BUF_PARAMS = 8,     /* f32 [P]                                                          */
  MOBROB_BUF_GRADS = 9,      /* f32 [P]   -- all-reduce target                                   */
  MOBROB_BUF_ADVSTAT = 10,   /* f64 [n_minibatches][4] (sum, sumsq, count, pad) -- all-reduce    */
  MOBROB_BUF_LAST_VALUES = 11, /* f32 [N]                                                        */
  MOBROB_BUF_LAST_DONES = 12,  /* f32 [N] (0/1)                                                  */
  MOBROB_BUF_CLIPPED_ACTIONS = 13, /* f32 [N][A] of the most recent act                          */
  MOBROB_BUF_EPISODE_START_STATE = 14, /* f32 [N] `_last_episode_starts` carried between rollouts   */
  MOBROB_BUF_TERMINAL_OBS = 15,    /* f32 [N][Dp] terminal observations of the rows truncated in the latest step */
  MOBROB_BUF_TERMINAL_VALUES = 16, /* f32 [N] V(terminal_obs) of those rows (time-limit bootstrap)              */
  MOBROB_BUF_TRUNCATED = 17,       /* u8  [N] TimeLimit.truncated flags of the latest step                       */
  MOBROB_BUF_ENV_STATE = 18,       /* f32 [N][12] goal-env state: pos[3] vel[3] goal[3] return length pad         */
  MOBROB_BUF_GRAD_EXCHANGE = 19,   /* f32 [P + 8]: the gradient followed by the eight loss sums of the minibatch (policy, value,
                                      approx_kl, clip fraction, row count, ...) -- what a data-parallel step sums across ranks */
  MOBROB_BUF_SDE_NOISE = 20,       /* f32 [N][HL][A] the environments' gSDE exploration matrices (use_sde engines only)          */
  MOBROB_BUF_COUNT = 21
};
/* Device pointer and size of a buffer.  Asking for the POINTER of ACTIONS / VALUES / LOG_PROBS / ADVANTAGES / RETURNS tells the
 * engine that the caller may write those arrays behind its back: the packed per-row training records the 256-wide gradient
 * kernel reads are then re-packed before every gradient launch instead of once per rollout (ptr_dev == NULL
 * queries the size only and changes nothing).  write_buffer needs no such care: it invalidates the records itself. */
int mobrob_ppo_buffer_info(mobrob_ppo_engine_t* e, int32_t which, void** ptr_dev, size_t* bytes);
/* copy with host layout [..][D] <-> device layout [..][Dp] handled for MOBROB_BUF_OBS */
int mobrob_ppo_read_buffer(mobrob_ppo_engine_t* e, int32_t which, void* host_out, size_t bytes);
int mobrob_ppo_write_buffer(mobrob_ppo_engine_t* e, int32_t which, const void* host_in, size_t bytes);
/* mark the rollout as complete (tests that inject a rollout with write_buffer) */
int mobrob_ppo_mark_rollout_ready(mobrob_ppo_engine_t* e);

/* Which matrix products of this engine run on the bf16 pipe with three-way split float32 operands (config.forward_x3, 256-wide tanh
 * nets): bit 0 = rollout policy forward and batched value pass, bit 1 = the hidden-layer products inside the gradient kernel (forward,
 * dh1, dW2, dW1; heads <= 16 wide, observation rows of 16 / 32 / 64 padded columns), bit 2 = that gradient kernel is the register-chained
 * k_chain_train (csrc/kernels_chain.h; MOBROB_NO_CHAIN=1 keeps k_fused_train<.., X3>).  0: everything on v_mfma_f32.  Measurement code
 * prices the kernels against the matrix peak of the pipe each product ran on (bench.py). */
int mobrob_ppo_x3_mode(const mobrob_ppo_engine_t* e);
/* How the latest mobrob_ppo_train / train_enqueue ran SB3's PPO.train (/root/reference/src/mobrob/rl_control/ppo.py:73-74): bit 0 = every
 * epoch as ONE co-operative launch (k_epoch64: gradient -> grid barrier -> fixed-order slab reduction -> grid barrier -> clip + Adam + packs
 * -> grid barrier, per minibatch; single rank, 64-wide networks, minibatches of at most 64 tiles, no target_kl), 0 = three launches per
 * optimizer step.  Same bits either way.  Every wait of the co-operative form is bounded (MOBROB_EPOCH_TIMEOUT_S, default 10): if a launch
 * gives up (workgroups not resident together), mobrob_ppo_train restores the snapshot it took of parameters and moments and re-runs the
 * update as three launches per step -- the call succeeds, the engine keeps that form, this query returns 0 from then on;
 * mobrob_ppo_train_enqueue (no snapshot) fails at the next synchronising call instead. */
int mobrob_ppo_update_mode(const mobrob_ppo_engine_t* e);

/* train/explained_variance as SB3's PPO.train logs it (stable_baselines3 2.0.0 ppo.py: explained_variance(rollout_buffer.values.flatten(),
 * rollout_buffer.returns.flatten()) = 1 - Var[returns - values] / Var[returns], NaN when the returns do not vary), over the rollout
 * in the buffer; reached from the reference through PPOCtrl.learn (src/mobrob/rl_control/ppo.py:73-74) with verbose / tensorboard_log
 * (ppo.py:52-56).  Synchronises the engine's stream. */
int mobrob_ppo_explained_variance(mobrob_ppo_engine_t* e, double* out);

/* GAE on the buffers as they are (after write_buffer of rewards/values/episode_starts/
 * last_values/last_dones): RolloutBuffer.compute_returns_and_advantage in isolation. */
int mobrob_ppo_compute_gae(mobrob_ppo_engine_t* e);

/* Feistel permutation used when perm == NULL (bit-exact vs oracle/ppo_oracle.py:feistel_permutation) */
int mobrob_ppo_feistel_permutation(mobrob_ppo_engine_t* e, int64_t n, uint64_t key, int64_t* out);

/* ---- kernel timing with HIP events on the engine's stream (bench.py roofline) ---------------- */
enum {
  MOBROB_K_ACT = 0,          /* rollout policy/value forward + sample                           */
  MOBROB_K_GAE = 1,          /* GAE(lambda) scan                                                */
  MOBROB_K_TRAIN_GRAD = 2,   /* minibatch forward + loss + backward (dominant kernel)           */
  MOBROB_K_APPLY = 3,        /* grad-norm + clip + Adam                                         */
  MOBROB_K_ENV = 4,          /* synthetic env source (+ time-limit bootstrap)                   */
  MOBROB_K_GRAD_REDUCE = 5,  /* deterministic reduction of the per-workgroup gradient slabs     */
  MOBROB_K_ALLREDUCE = 6,    /* data parallel: the all-reduces of train_dp (gradient + statistics) */
  MOBROB_K_COUNT = 7
};
/* on: 0 = off, 1 = bracket every phase with HIP events, otherwise a mask with bit (MOBROB_K_x + 1) set for each phase
 * to bracket.  An event pair costs a few microseconds of GPU time per launch: bracketing all four launches of an
 * optimizer step slows the headline shape by 3.7 %, the dominant kernel alone by under 1 %. */
int mobrob_ppo_profile_enable(mobrob_ppo_engine_t* e, int32_t on);
/* accumulated since enable: total milliseconds and launch-group count per id */
int mobrob_ppo_profile_read(mobrob_ppo_engine_t* e, double* ms /*[K_COUNT]*/, int64_t* calls /*[K_COUNT]*/);

#ifdef __cplusplus
}
#endif
#endif /* MOBROB_PPO_H */
