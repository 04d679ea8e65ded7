// libmobrob_ppo.so -- host-side orchestration + C ABI (include/mobrob_ppo.h) of the MI355X PPO engine.
// Reference surface replaced: stable_baselines3.PPO as configured by the reference's
// src/mobrob/rl_control/ppo.py:50-59 and driven by :73-77 (see the header for per-entry citations).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types and prototypes only: the library is dlopen'ed by comm_init (no link-time dependency)
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <cstdlib>
#include <string>
#include <vector>

#ifdef MOBROB_POISON_LDS
// Diagnostic build (scratch/build_variant.sh poison -DMOBROB_POISON_LDS; never the product library): every kernel launch is
// preceded by one that fills the LDS of every CU with 0xFFFFFFFF (a NaN as float, -1 as an index).  LDS is not cleared
// between kernels, so a kernel that reads a word it did not write sees whatever the previous tenant of the CU left there:
// right on most runs, wrong once in a while.  Under this build it is wrong every time (scratch/README.md: GPU suite with
// MOBROB_PPO_LIB=scratch/lib_poison.so).
namespace mobrob {
__global__ __launch_bounds__(1024) void k_poison_lds() {
  extern __shared__ unsigned poison_words[];
  volatile unsigned* w = poison_words;
  for (int i = threadIdx.x; i < 160 * 256; i += 1024) w[i] = 0xFFFFFFFFu;
}
inline void poison_lds(hipStream_t st) {
  static bool once = false;
  if (!once) { (void)hipFuncSetAttribute((const void*)k_poison_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); once = true; }
  k_poison_lds<<<dim3(512), dim3(1024), 160 * 1024, st>>>();
}
}  // namespace mobrob
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernelName, nb, nt, mem, st, ...)                    \
  do {                                                                          \
    mobrob::poison_lds(st);                                                     \
    hipLaunchKernelGGLInternal((kernelName), nb, nt, mem, st, __VA_ARGS__);     \
  } while (0)
#endif

#include "../../include/mobrob_ppo.h"
#include "kernels_generic.h"
#include "kernels_fused.h"
#include "fused_dispatch.h"
#include "kernels_env.h"
#include "kernels_rollout.h"
#include "kernels_epoch64.h"
#include "robot_ctrl.h"
#include "oneshot_allreduce.h"

using namespace mobrob;

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIPC(expr)                                                                                            \
  do {                                                                                                        \
    hipError_t _e = (expr);                                                                                   \
    if (_e != hipSuccess)                                                                                     \
      return fail(MOBROB_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)
#define CHK(expr)                   \
  do {                              \
    int _r = (expr);                \
    if (_r != MOBROB_OK) return _r; \
  } while (0)

inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline int rup(int a, int b) { return cdiv(a, b) * b; }

struct ProfSpan {
  hipEvent_t a, b;
  int id;
};

// Canonical tensors in SB3's registration order: log_std, (W, b) of every policy hidden layer, (W, b) of every value hidden layer,
// action head, value head.  With one to eight hidden layers per network that is 9 .. 37 tensors; the ids are the engine's
// (engine_dims), the names below resolve through `e`.  Two hidden layers per network give the numbering 0 .. 12 the fused
// kernels' argument structs were written for.
// (kMaxHidden = 8: device_utils.h; kMaxTensors = 37: kernels_fused.h)
#define T_LOGSTD 0
#define T_PW1 (e->tPW[0])
#define T_PB1 (e->tPB[0])
#define T_PW2 (e->tPW[1])
#define T_PB2 (e->tPB[1])
#define T_VW1 (e->tVW[0])
#define T_VB1 (e->tVB[0])
#define T_VW2 (e->tVW[1])
#define T_VB2 (e->tVB[1])
#define T_AW (e->tAW)
#define T_AB (e->tAB)
#define T_VW (e->tVWh)
#define T_VB (e->tVBh)

}  // namespace

struct mobrob_ppo_engine {
  mobrob_ppo_config_t cfg;
  int D, Dp, A, Ap, H1, H2, G1, G2, N, T, P;   // H1, H2 / G1, G2: the first two hidden widths (the fused kernels' view; H2 = 0 at depth 1)
  int Lp = 2, Lv = 2;                          // hidden layers of the policy / value network (1 .. kMaxHidden)
  int Hp[kMaxHidden] = {0}, Hv[kMaxHidden] = {0};   // their widths
  int HL = 0, GL = 0;                          // width of the last hidden layer of each network (what the heads read)
  int tPW[kMaxHidden] = {0}, tPB[kMaxHidden] = {0}, tVW[kMaxHidden] = {0}, tVB[kMaxHidden] = {0}, tAW = 0, tAB = 0, tVWh = 0, tVBh = 0;
  int ntens = 13;
  int Bl;        // local minibatch rows (batch_size / world)
  int nmb;       // minibatches per epoch
  int rows_max;  // max rows any forward sees at once
  int offs[kMaxTensors + 1];  // canonical parameter offsets (SB3 order); offs[ntens] = P
  hipStream_t stream = nullptr;
  bool own_stream = false;
  // parameters / optimizer
  float *params = nullptr, *grads = nullptr /* [P] + 8 loss sums */, *m = nullptr, *v = nullptr;
  NormChunk* chunks_dev = nullptr;
  double* chunk_partial = nullptr;
  int nchunks = 0;
  float *pW1p = nullptr, *vW1p = nullptr, *aWp = nullptr, *vWp = nullptr;  // zero-padded compute copies
  int64_t adam_step = 0;
  // rollout storage
  float *obs = nullptr, *actions = nullptr, *rewards = nullptr, *es = nullptr, *values = nullptr, *logp = nullptr,
        *adv = nullptr, *ret = nullptr;
  float *last_values = nullptr, *last_dones = nullptr, *prev_dones = nullptr, *dones_tmp = nullptr;
  float *clip_act = nullptr, *rew_tmp = nullptr, *term_obs = nullptr, *term_val = nullptr, *eps_dev = nullptr;
  uint8_t *trunc_dev = nullptr, *dones_u8 = nullptr;
  int *ep_len = nullptr, *ep_len2 = nullptr;
  uint32_t* ctr_dev = nullptr;  // [0] eps draw base, [1] env step base (device-resident: graph replays and the persistent rollout kernels advance them)
  hipGraph_t ro_graph = nullptr;
  hipGraphExec_t ro_exec = nullptr;
  struct RolloutSpec {  // what the captured rollout graph was built for
    int kind = 0;       // 0 none, 1 synthetic source, 2 goal environment
    float p_term = 0.f;
    int time_limit = 0;
    GoalEnvParams goal{};
  } ro_spec;
  int env_started = 0;  // env kind whose state is live on the device (0 = none)
  hipStream_t vstream = nullptr;          // batched value pass of finished rollout chunks, concurrent with the rollout
  std::vector<hipEvent_t> ev_chunks;      // one event per rollout chunk (an event is recorded once per capture)
  hipEvent_t ev_vdone = nullptr;
  float* gstate[2] = {nullptr, nullptr};  // goal env state, double buffered [N][kGoalStateFloats]
  double* ep_stats = nullptr;             // [4] episode statistics of the goal env + the Monitor ring (kernels_env.h)
  uint64_t ep_ring_read = 0;              // records already handed out by mobrob_ppo_episode_records
  uint32_t draw_counter = 0;  // Philox draw index for eps
  // pipelined host-env rollout (act_part / wait_part / store_part): per-part step indices and completion events
  int nparts = 0;
  int part_act_t[MOBROB_MAX_PARTS] = {0}, part_store_t[MOBROB_MAX_PARTS] = {0};
  int part_obs_t[MOBROB_MAX_PARTS] = {0};  // rollout slot whose observations store_part already pulled (-1: none)
  hipEvent_t ev_part[MOBROB_MAX_PARTS] = {nullptr};
  const void* pinned_seen[8] = {nullptr};  // pointers validated by is_pinned_cached since rollout_begin
  unsigned pinned_seen_n = 0;
  uint32_t draw_ro0 = 0;  // draw_counter at rollout_begin: part p at its step t draws with draw_ro0 + t
  // host-env rollout SERVED by the persistent rollout kernel (collect_host_served): flags in pinned memory, abort word on the device
  unsigned* srv_flags = nullptr;   // [srv_blocks] workgroup flags | [MOBROB_MAX_PARTS][16] host words | [16] error word
  int srv_blocks = 0;
  int* srv_abort = nullptr;
  uint32_t env_step_counter = 0;
  int t = 0;
  bool rollout_ready = false;
  // training records of the H = 256 gradient kernel (kernels_fused.h, k_build_train_records): rebuilt before the next
  // gradient launch whenever one of the five arrays they are packed from may have changed
  bool train_rec_valid = false;
  bool train_rec_external = false;  // a device pointer to one of those arrays was handed out (buffer_info): never trusted again
  // update
  int* rows = nullptr;          // [T*N] device row index per permuted position
  int64_t* perm_dev = nullptr;  // [T*N]
  double* advstat = nullptr;    // [nmb][4]
  double* expvar_part = nullptr;  // [kEvBlocks][4] partial sums of mobrob_ppo_explained_variance
  unsigned long long* advbins = nullptr;  // [nmb][2] fixed-point sums of k_adv_stats_stream (+ 1 word: max |adv| bits)
  uint64_t perm_counter = 0;
  unsigned adv_pass = 0;        // epoch_begin calls so far: which of the two max-|adv| words is live
  bool epoch_open = false;
  float* stats = nullptr;  // [stats_cap][8]
  int stats_cap = 0, stats_n = 0;
  int cur_count = 0;
  bool grad_pending = false;
  double clip_vf = -1.0;      // SB3 clip_range_vf (< 0: None); mobrob_ppo_set_hyper
  double target_kl = -1.0;    // SB3 target_kl (< 0: None): early stop of PPO.train() when a minibatch's approx_kl > 1.5 target
  int last_epochs_started = 0, last_stopped_early = 0, last_steps_applied = 0;  // of the latest mobrob_ppo_train*
  ncclComm_t comm = nullptr;  // RCCL communicator of the data-parallel job (mobrob_ppo_comm_init)
  int64_t allreduce_calls = 0, allreduce_bytes = 0;  // since the last mobrob_ppo_allreduce_counters(reset)
  struct OneShot {  // peer-mapped exchange buffers of the one-shot all-reduce (oneshot_allreduce.h)
    char* xbuf = nullptr;                 // own: [2 slots][payload] + [kOneShotMaxChunks] u64 flags (a hipMalloc of its own: IPC exports whole allocations)
    size_t payload = 0;
    char* peer[kOneShotMaxRanks] = {nullptr};  // every rank's buffer as mapped here (peer[rank] == xbuf)
    int world = 0, rank = -1;
    bool ready = false;
    unsigned long long seq = 0;           // messages exchanged so far: the same number on every rank
    int* error = nullptr;                 // pinned host word the kernel raises when a peer never arrived
    long long timeout_ticks = 0;
  } oneshot;
  // norm records of the reduction kernels (kernels_fused.h: block_norm_records): used inside mobrob_ppo_train only
  double* norm_rec_sum = nullptr; int* norm_rec_t = nullptr; int* fold_idx_dev = nullptr;
  int fold_start[kMaxTensors + 1] = {0};
  bool use_norm_records = false;
  // one co-operative launch per epoch for small minibatches of 64-wide nets (kernels_epoch64.h)
  bool epoch_kernel_on = true;        // MOBROB_EPOCH_KERNEL=0 / mobrob_ppo_set_hyper(MOBROB_HYPER_EPOCH_KERNEL, 0): the three launches per step
  unsigned* epoch_bar = nullptr;      // [0] arrival counter, [1] abort word
  float* epoch_consts = nullptr;      // [nmb][2] per-step Adam constants of the epoch being launched
  int* epoch_idx = nullptr;           // [nmb] statistics rows of the epoch being launched
  char* epoch_stage = nullptr;        // pinned staging of the two (12 bytes per step), one slot per epoch of a train() call
  int* epoch_err_host = nullptr;      // pinned: raised by a launch that gave up at a barrier
  hipEvent_t epoch_ev = nullptr;      // behind the last staging copy of a train() call: the staging may be rewritten after it
  int last_update_mode = 0;           // bit 0: the latest train() ran its epochs as k_epoch64 launches
  float* epoch_snap = nullptr;        // [3][P] parameters and both moments as they were before a mobrob_ppo_train that uses k_epoch64 (restored if it aborts)
  int rollout64_tile_max = 256;  // rollouts of up to this many 32-env tiles use k_rollout64_tile (MOBROB_ROLLOUT64_TILE_MAX)
  int pair64_min_tiles = 65;     // minibatches of at least this many tiles use k_pair64_train (MOBROB_PAIR64_MIN_TILES; 0: never)
  int split64_max_tiles = 64;  // minibatches of up to this many 32-row tiles use k_split64_train (MOBROB_SPLIT64_MAX_TILES)
  // generic-path workspace
  float *Xg = nullptr, *actg = nullptr, *lpg = nullptr, *advg = nullptr, *retg = nullptr, *oldvg = nullptr;
  float *hp[kMaxHidden] = {nullptr}, *hv[kMaxHidden] = {nullptr}, *mu = nullptr, *vout = nullptr;     // activations per hidden layer
  float *dmu = nullptr, *dv = nullptr, *dzp[kMaxHidden] = {nullptr}, *dzv[kMaxHidden] = {nullptr};     // pre-activation gradients
  float *pred_obs = nullptr, *pred_act = nullptr;
  // gSDE (use_sde; generic chain only): per-env exploration matrices [N][HL][A], the single matrix [HL][A], staging for supplied noise,
  // the log_std-gradient GEMM's operands of a minibatch
  int gemm_tiles = 0;   // MOBROB_GEMM_TILES: 0 = by shape (launch_gemm)
  bool sde = false, sde_hold = false;   // hold: the caller supplies the noise (mobrob_ppo_sde_set_noise); no automatic resampling
  float *sde_E = nullptr, *sde_E1 = nullptr, *sde_lat2 = nullptr, *sde_gsig = nullptr, *sde_graw = nullptr;
  float *sde_S2 = nullptr, *sde_var = nullptr;   // std^2 table [HL][A] of the current log_std; variance [rows][Ap] of the rows in flight
  SdeMode sde_mode{1, 0};   // policy_kwargs full_std / use_expln
  // rollout streamer (host-env path): pinned staging + a side stream for the H2D/D2H copies
  hipStream_t cstream = nullptr;
  hipEvent_t ev_in = nullptr, ev_k = nullptr, ev_store = nullptr;
  struct Stage {
    float *obs = nullptr, *eps = nullptr, *rew = nullptr, *term = nullptr;
    uint8_t *dones = nullptr, *trunc = nullptr;
  } stage[2];
  float *o_raw = nullptr, *o_clip = nullptr, *o_val = nullptr, *o_lp = nullptr;  // pinned output staging
  int stage_i = 0;
  // profiling
  bool prof_on = false;
  uint32_t prof_mask = ~0u;  // which MOBROB_K_* phases are bracketed while prof_on
  std::vector<ProfSpan> spans;
  std::vector<hipEvent_t> ev_pool;
  double prof_ms[MOBROB_K_COUNT] = {0};
  int64_t prof_calls[MOBROB_K_COUNT] = {0};
  std::vector<void*> allocs;
  // device arena: every device buffer of the engine is carved out of ONE allocation (owned, or handed in by the
  // caller so that several engines -- the robot types of a mixed fleet -- pack into one rollout buffer)
  char* arena = nullptr;
  size_t arena_bytes = 0, arena_used = 0;
  bool plan_only = false;  // sizing pass: count bytes, touch no device
  std::vector<NormChunk> chunk_table;
  FusedState fused;
};

namespace {

int check_async_error(mobrob_ppo_engine* e);  // defined with the one-shot all-reduce

constexpr size_t kArenaAlign = 256;

template <typename Tp>
int dalloc(mobrob_ppo_engine* e, Tp** p, size_t count) {
  const size_t bytes = (std::max<size_t>(count, 1) * sizeof(Tp) + kArenaAlign - 1) / kArenaAlign * kArenaAlign;
  if (e->plan_only) {
    e->arena_used += bytes;
    *p = nullptr;
    return MOBROB_OK;
  }
  if (e->arena_used + bytes > e->arena_bytes)
    return fail(MOBROB_ERR_INVALID, "device arena too small: need more than %zu bytes", e->arena_bytes);
  void* q = e->arena + e->arena_used;
  e->arena_used += bytes;
  HIPC(hipMemsetAsync(q, 0, bytes, e->stream));
  *p = static_cast<Tp*>(q);
  return MOBROB_OK;
}

hipEvent_t get_event(mobrob_ppo_engine* e) {
  if (!e->ev_pool.empty()) {
    hipEvent_t ev = e->ev_pool.back();
    e->ev_pool.pop_back();
    return ev;
  }
  hipEvent_t ev;
  (void)hipEventCreate(&ev);
  return ev;
}

struct ProfScope {
  mobrob_ppo_engine* e;
  ProfSpan s;
  bool on;
  ProfScope(mobrob_ppo_engine* e_, int id) : e(e_), on(e_->prof_on && ((e_->prof_mask >> id) & 1u)) {
    if (on) {
      s.id = id;
      s.a = get_event(e);
      s.b = get_event(e);
      (void)hipEventRecord(s.a, e->stream);
    }
  }
  ~ProfScope() {
    if (on) {
      (void)hipEventRecord(s.b, e->stream);
      e->spans.push_back(s);
    }
  }
};

void prof_resolve(mobrob_ppo_engine* e) {
  for (auto& s : e->spans) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
      e->prof_ms[s.id] += ms;
      e->prof_calls[s.id] += 1;
    }
    e->ev_pool.push_back(s.a);
    e->ev_pool.push_back(s.b);
  }
  e->spans.clear();
}

// ---- GEMM launchers --------------------------------------------------------------------------------
// One GEMM of the generic chain as the launcher takes it: mode / epilogue (template parameters of k_gemm), arguments, batch split
struct GemmOp {
  int mode = 0, epi = 0, ksplit = 1;
  GemmArgs g{};
};
using GemmQueue = std::vector<GemmOp>;

template <int MODE, int EPI, int TM, int TN>
void launch_gemm_tiles(mobrob_ppo_engine* e, const GemmOp& a, const GemmOp* b) {
  // Orientation of a block's four waves, per problem.  Forward / input-gradient GEMMs stream a tall A ([rows][K]) against a small B
  // (weights): with the waves side by side along N the block fetches its A rows ONCE (L1 / one XCD's L2) where four column blocks
  // scattered over the chip fetched them four times -- at 65 536 x 256 x 256 that was 268 MB through L2 per launch against 67.
  // Only where the column tiles fill the four waves; the weight-gradient GEMM (both operands tall) keeps them stacked along M.
  static const bool wn_on = getenv("MOBROB_GEMM_WN") == nullptr || atoi(getenv("MOBROB_GEMM_WN")) != 0;
  GemmArgs ga = a.g, gb = b ? b->g : a.g;
  int gx = 1, gy = 1;
  for (GemmArgs* g : {&ga, &gb}) {
    const int mt = cdiv(g->M, 32 * TM), nt = cdiv(g->N, 32 * TN);
    g->wn = (wn_on && MODE != MODE_TN && nt % 4 == 0) ? 1 : 0;
    gx = std::max(gx, g->wn ? mt : cdiv(mt, 4));
    gy = std::max(gy, g->wn ? nt / 4 : nt);
  }
  const int ks = b ? std::max(a.ksplit, b->ksplit) : a.ksplit;
  hipLaunchKernelGGL((k_gemm<MODE, EPI, TM, TN>), dim3(gx, gy, b ? 2 * ks : ks), dim3(256), 0, e->stream, ga, gb, b ? 1 : 0);
}
// tiles per wave by shape: 2x2 wherever both extents leave room for it, 2x1 for narrow outputs (heads), 1x1 for tiny ones.
// MOBROB_GEMM_TILES (read when the engine is created): 1 keeps one tile per wave (A/B); 21 / 22 force 2x1 / 2x2 wherever the output
// has more than one tile in that direction, whatever the wave count (the parity tests drive small shapes through every form).
// b: a second problem of the same mode / epilogue for the same launch (the other network's GEMM at this point of the chain).
template <int MODE, int EPI>
void launch_gemm(mobrob_ppo_engine* e, const GemmOp& a, const GemmOp* b = nullptr) {
  const int f = e->gemm_tiles;
  const int M = b ? std::max(a.g.M, b->g.M) : a.g.M, N = b ? std::max(a.g.N, b->g.N) : a.g.N;
  if (f == 22 && M > 32 && N > 32) return launch_gemm_tiles<MODE, EPI, 2, 2>(e, a, b);
  if ((f == 21 || f == 22) && M > 32) return launch_gemm_tiles<MODE, EPI, 2, 1>(e, a, b);
  // fat wave tiles only while they still leave ~8 waves per CU: a rollout step (4096 rows) is 1024 one-tile waves, 256 as 2x2
  auto waves = [&](int tm, int tn) {
    long w = (long)cdiv(a.g.M, 32 * tm) * cdiv(a.g.N, 32 * tn) * a.ksplit;
    if (b) w += (long)cdiv(b->g.M, 32 * tm) * cdiv(b->g.N, 32 * tn) * b->ksplit;
    return w;
  };
  if (f != 0 || M <= 32 || waves(2, 1) < 2048) launch_gemm_tiles<MODE, EPI, 1, 1>(e, a, b);
  else if (N <= 32 || waves(2, 2) < 2048) launch_gemm_tiles<MODE, EPI, 2, 1>(e, a, b);
  else launch_gemm_tiles<MODE, EPI, 2, 2>(e, a, b);
}
void launch_op(mobrob_ppo_engine* e, const GemmOp& a, const GemmOp* b = nullptr) {
  if (a.mode == MODE_NT && a.epi == EPI_BIAS_TANH) launch_gemm<MODE_NT, EPI_BIAS_TANH>(e, a, b);
  else if (a.mode == MODE_NT) launch_gemm<MODE_NT, EPI_BIAS>(e, a, b);
  else if (a.mode == MODE_NN) launch_gemm<MODE_NN, EPI_DTANH_COLSUM>(e, a, b);
  else launch_gemm<MODE_TN, EPI_ATOMIC>(e, a, b);
}
// tile class launch_gemm would pick for one op on its own: 11, 21 or 22
int gemm_tile_class(const mobrob_ppo_engine* e, const GemmOp& a) {
  const int f = e->gemm_tiles, M = a.g.M, N = a.g.N;
  if (f == 22 && M > 32 && N > 32) return 22;
  if ((f == 21 || f == 22) && M > 32) return 21;
  const long w21 = (long)cdiv(M, 64) * cdiv(N, 32) * a.ksplit, w22 = (long)cdiv(M, 64) * cdiv(N, 64) * a.ksplit;
  if (f != 0 || M <= 32 || w21 < 2048) return 11;
  return (N <= 32 || w22 < 2048) ? 21 : 22;
}
// up to four one-tile problems of any kind in ONE launch (k_gemm_multi)
void launch_multi(mobrob_ppo_engine* e, const GemmOp* const* ops, int n) {
  GemmMulti m{};
  int gx = 1, gy = 1, z = 0;
  for (int p = 0; p < 4; ++p) {
    const GemmOp& o = *ops[p < n ? p : n - 1];   // (unused slots repeat the last problem; their z range is empty)
    m.g[p] = o.g; m.mode[p] = o.mode; m.epi[p] = o.epi;
    m.zbeg[p] = z;
    if (p < n) {
      z += o.ksplit;
      gx = std::max(gx, cdiv(cdiv(o.g.M, 32), 4)); gy = std::max(gy, cdiv(o.g.N, 32));
    }
  }
  m.zbeg[4] = z;
  hipLaunchKernelGGL(k_gemm_multi, dim3(gx, gy, z), dim3(256), 0, e->stream, m);
}
// The two networks' chains side by side, in STAGES of `width` consecutive GEMMs per network that depend on earlier stages only
// (forward: 1 -- a layer; backward: 2 -- a layer's weight-gradient and input-gradient GEMM).  A stage whose GEMMs are all of the
// one-tile class (small minibatches, rollout steps of few environments) goes out as ONE launch of up to four problems; otherwise
// the j-th GEMMs of both networks go out pairwise where they are of the same kind (they always are at equal depths; the tails of
// unequal depths, and a head opposite a hidden layer, go alone).  MOBROB_GEMM_PAIR=0: one launch per GEMM; =1: pairs only.
void run_queues(mobrob_ppo_engine* e, const GemmQueue& qa, const GemmQueue& qb, int width) {
  static const int pairing = getenv("MOBROB_GEMM_PAIR") == nullptr ? 2 : atoi(getenv("MOBROB_GEMM_PAIR"));
  const size_t n = std::max(qa.size(), qb.size());
  for (size_t j0 = 0; j0 < n; j0 += width) {
    const GemmOp* st[4];
    int ns = 0;
    bool small = pairing >= 2;
    for (size_t j = j0; j < j0 + width && j < n; ++j)
      for (const GemmQueue* q : {&qa, &qb})
        if (j < q->size()) {
          st[ns++] = &(*q)[j];
          small = small && gemm_tile_class(e, (*q)[j]) == 11;
        }
    if (small && ns >= 2) { launch_multi(e, st, ns); continue; }
    for (size_t j = j0; j < j0 + width && j < n; ++j) {
      const GemmOp* a = j < qa.size() ? &qa[j] : nullptr;
      const GemmOp* b = j < qb.size() ? &qb[j] : nullptr;
      if (a && b && pairing >= 1 && a->mode == b->mode && a->epi == b->epi) { launch_op(e, *a, b); continue; }
      if (a) launch_op(e, *a);
      if (b) launch_op(e, *b);
    }
  }
}
// q: non-null = the GEMM is queued (run_queues pairs it with the other network's), null = launched now

// Y[M x Nn] = act(X[M x K] . W[Nn x K]^T + bias)
void linear_fwd(mobrob_ppo_engine* e, const float* X, int ldx, const float* W, int ldw, const float* bias, float* Y,
                int ldy, int M, int Nn, int K, bool tanh_, float* Z = nullptr, GemmQueue* q = nullptr) {
  GemmOp o;
  GemmArgs& g = o.g;
  g.A = X; g.B = W; g.C = Y; g.M = M; g.N = Nn; g.K = K; g.lda = ldx; g.ldb = ldw; g.ldc = ldy; g.bias = bias;
  g.act = e->cfg.activation;
  g.Z = (tanh_ && act_needs_z(g.act)) ? Z : nullptr; g.ldz = ldy;   // pre-activations for SiLU / GELU / Mish backward (same shape as Y)
  o.mode = MODE_NT; o.epi = tanh_ ? EPI_BIAS_TANH : EPI_BIAS;
  if (q) q->push_back(o); else launch_op(e, o);
}
// dZ[M x Nn] = (dY[M x K] . W[K x Nn]) * act'(.) ; column sums -> bias grad
void linear_bwd_input(mobrob_ppo_engine* e, const float* dY, int ldd, const float* W, int ldw, const float* H, int ldh,
                      float* dZ, int ldz, float* bias_grad, int M, int Nn, int K, GemmQueue* q = nullptr) {
  GemmOp o;
  GemmArgs& g = o.g;
  g.A = dY; g.B = W; g.C = dZ; g.M = M; g.N = Nn; g.K = K; g.lda = ldd; g.ldb = ldw; g.ldc = ldz;
  g.Hact = H; g.ldh = ldh; g.colsum = bias_grad; g.act = e->cfg.activation;
  o.mode = MODE_NN; o.epi = EPI_DTANH_COLSUM;
  if (q) q->push_back(o); else launch_op(e, o);
}
// dW[M x Nn] += dY[rows x M]^T . X[rows x Nn]
void linear_bwd_weight(mobrob_ppo_engine* e, const float* dY, int ldd, const float* X, int ldx, float* dW, int ldw,
                       int M, int Nn, int rows, GemmQueue* q = nullptr) {
  GemmOp o;
  GemmArgs& g = o.g;
  g.A = dY; g.B = X; g.C = dW; g.M = M; g.N = Nn; g.K = rows; g.lda = ldd; g.ldb = ldx; g.ldc = ldw;
  // waves of one pass over the output (launch_gemm's tiling: 64-row / 64-column wave tiles where the extents allow); the batch rows
  // are split until ~2048 waves are in flight (8 per CU) -- every split adds M x Nn float atomics, so not further
  const int wtiles = cdiv(M, M <= 32 ? 32 : 64) * cdiv(Nn, (M <= 32 || Nn <= 32) ? 32 : 64);
  int ksplit = std::max(1, std::min(cdiv(rows, 64), cdiv(2048, wtiles)));   // (launch_gemm then finds >= 2048 waves wherever the rows allow)
  g.kchunk = rup(cdiv(rows, ksplit), 8);
  o.ksplit = cdiv(rows, g.kchunk);
  o.mode = MODE_TN; o.epi = EPI_ATOMIC;
  if (q) q->push_back(o); else launch_op(e, o);
}

float* Pp(mobrob_ppo_engine* e, int t) { return e->params + e->offs[t]; }
float* Gp(mobrob_ppo_engine* e, int t) { return e->grads + e->offs[t]; }

// x3 packs of the hidden layers of both networks from the canonical parameters (kernels_fused.h): one launch
void pack_x3_all(mobrob_ppo_engine* e) {
  if (e->fused.net[0].W2x == nullptr) return;
  PackX3Args a{};
  const int w1[2] = {T_PW1, T_VW1}, w2[2] = {T_PW2, T_VW2};
  int maxthreads = 0;
  for (int n = 0; n < 2; ++n) {
    const int i1 = 3 * n, i2 = 3 * n + 1, i3 = 3 * n + 2;
    a.W[i1] = Pp(e, w1[n]); a.N[i1] = FH; a.K[i1] = e->D; a.ld[i1] = e->D; a.NB[i1] = FH / 32; a.KS[i1] = e->Dp / 16; a.scale[i1] = kTanhScale;
    a.out[i1] = reinterpret_cast<unsigned short*>(const_cast<unsigned*>(e->fused.net[n].W1x));
    a.W[i2] = Pp(e, w2[n]); a.N[i2] = FH; a.K[i2] = FH; a.ld[i2] = FH; a.NB[i2] = FH / 32; a.KS[i2] = FH / 16; a.scale[i2] = kTanhScale;
    a.out[i2] = reinterpret_cast<unsigned short*>(const_cast<unsigned*>(e->fused.net[n].W2x));
    a.W[i3] = Pp(e, w2[n]); a.N[i3] = FH; a.K[i3] = FH; a.ld[i3] = FH; a.NB[i3] = FH / 32; a.KS[i3] = FH / 16; a.scale[i3] = 1.0f; a.trans[i3] = 1;
    a.out[i3] = reinterpret_cast<unsigned short*>(const_cast<unsigned*>(e->fused.net[n].W2bx));
    maxthreads = std::max(maxthreads, std::max(a.NB[i1] * a.KS[i1], a.NB[i2] * a.KS[i2]) * 512);
  }
  hipLaunchKernelGGL(k_pack_x3_multi, dim3(cdiv(maxthreads, 256), 6), dim3(256), 0, e->stream, a);
  if (e->fused.train_chain) {  // chain packs of k_chain_train (kernels_chain.h)
    ChainPackArgs c{};
    const int w3[2] = {T_AW, T_VW};
    for (int n = 0; n < 2; ++n) {
      c.W1[n] = Pp(e, w1[n]); c.W2[n] = Pp(e, w2[n]); c.W3[n] = Pp(e, w3[n]);
      c.w1c[n] = reinterpret_cast<unsigned short*>(const_cast<unsigned*>(e->fused.net[n].W1c));
      c.w2c[n] = reinterpret_cast<unsigned short*>(const_cast<unsigned*>(e->fused.net[n].W2c));
      c.w2bc[n] = reinterpret_cast<unsigned short*>(const_cast<unsigned*>(e->fused.net[n].W2bc));
      c.w3c[n] = const_cast<float*>(e->fused.net[n].W3c); c.w3bc[n] = const_cast<float*>(e->fused.net[n].W3bc);
      c.head[n] = n == 0 ? e->A : 1;
    }
    c.D = e->D; c.Dp = e->Dp;
    hipLaunchKernelGGL(k_pack_chain, dim3(FH * FH / 256, 2), dim3(256), 0, e->stream, c);
  }
}

void repack(mobrob_ppo_engine* e) {
  auto pad = [&](const float* src, float* dst, int r, int c, int rp, int cp) {
    hipLaunchKernelGGL(k_pad_rows, dim3(cdiv(rp * cp, 256)), dim3(256), 0, e->stream, src, dst, r, c, rp, cp);
  };
  pad(Pp(e, T_PW1), e->pW1p, e->H1, e->D, e->H1, e->Dp);
  pad(Pp(e, T_VW1), e->vW1p, e->G1, e->D, e->G1, e->Dp);
  pad(Pp(e, T_AW), e->aWp, e->A, e->HL, e->Ap, e->HL);
  pad(Pp(e, T_VW), e->vWp, 1, e->GL, 8, e->GL);
  fused_repack(e->fused, e->params, e->offs, e->stream);
  pack_x3_all(e);
}

// forward of both networks on `rows` device rows of padded observations (ld = Dp); mu ld = Ap, v ld = 1
void forward_generic(mobrob_ppo_engine* e, const float* X, int rows, bool want_pi, float* mu_out, bool want_v,
                     float* v_out, bool keep_z = false) {
  // layer 0 reads the zero-padded copy of its weights (observation rows are padded to Dp columns); one to eight hidden layers.
  // keep_z (the training forward of a minibatch, rows <= Bl): layer l's pre-activations are left in its dz buffer for the backward
  // epilogue of the activations that need them (act_needs_z).  The two networks' GEMMs are queued and go out pairwise (run_queues).
  GemmQueue qp, qv;
  if (want_pi) {
    linear_fwd(e, X, e->Dp, e->pW1p, e->Dp, Pp(e, e->tPB[0]), e->hp[0], e->Hp[0], rows, e->Hp[0], e->Dp, true, keep_z ? e->dzp[0] : nullptr, &qp);
    for (int l = 1; l < e->Lp; ++l)
      linear_fwd(e, e->hp[l - 1], e->Hp[l - 1], Pp(e, e->tPW[l]), e->Hp[l - 1], Pp(e, e->tPB[l]), e->hp[l], e->Hp[l], rows, e->Hp[l], e->Hp[l - 1], true,
                 keep_z ? e->dzp[l] : nullptr, &qp);
    linear_fwd(e, e->hp[e->Lp - 1], e->HL, e->aWp, e->HL, Pp(e, T_AB), mu_out, e->Ap, rows, e->A, e->HL, false, nullptr, &qp);
  }
  if (want_v) {
    linear_fwd(e, X, e->Dp, e->vW1p, e->Dp, Pp(e, e->tVB[0]), e->hv[0], e->Hv[0], rows, e->Hv[0], e->Dp, true, keep_z ? e->dzv[0] : nullptr, &qv);
    for (int l = 1; l < e->Lv; ++l)
      linear_fwd(e, e->hv[l - 1], e->Hv[l - 1], Pp(e, e->tVW[l]), e->Hv[l - 1], Pp(e, e->tVB[l]), e->hv[l], e->Hv[l], rows, e->Hv[l], e->Hv[l - 1], true,
                 keep_z ? e->dzv[l] : nullptr, &qv);
    linear_fwd(e, e->hv[e->Lv - 1], e->GL, e->vWp, e->GL, Pp(e, T_VB), v_out, 1, rows, 1, e->GL, false, nullptr, &qv);
  }
  run_queues(e, qp, qv, 1);
}

void forward(mobrob_ppo_engine* e, const float* X, int rows, bool want_pi, float* mu_out, bool want_v, float* v_out) {
  if (e->fused.enabled && fused_forward(e->fused, X, rows, want_pi, mu_out, e->Ap, want_v, v_out, e->stream)) return;
  forward_generic(e, X, rows, want_pi, mu_out, want_v, v_out);
}

int value_net_width_sum(const mobrob_ppo_engine* e) {
  int s = 0;
  for (int l = 0; l < e->Lv; ++l) s += e->Hv[l];
  return s;
}
// the value network as the per-row evaluators take it (canonical parameters; layers beyond the network's depth are null)
ValueNetArgs value_net_args(mobrob_ppo_engine* e) {
  ValueNetArgs vn{};
  for (int l = 0; l < e->Lv; ++l) { vn.W[l] = Pp(e, e->tVW[l]); vn.b[l] = Pp(e, e->tVB[l]); vn.G[l] = e->Hv[l]; }
  vn.Wv = Pp(e, T_VW); vn.bv = Pp(e, T_VB);
  vn.L = e->Lv; vn.act = e->cfg.activation; vn.width_sum = value_net_width_sum(e);
  return vn;
}
BootNetArgs boot_args(mobrob_ppo_engine* e) { return BootNetArgs{value_net_args(e), (float)e->cfg.gamma, e->term_val}; }

void value_flagged(mobrob_ppo_engine* e, const float* obs_rows, const uint8_t* flags, float* out,
                   float* bootstrap_rewards = nullptr) {
  const size_t sm = (size_t)(e->D + value_net_width_sum(e) + 16) * sizeof(float);
  hipLaunchKernelGGL(k_value_flagged, dim3(e->N), dim3(256), sm, e->stream, obs_rows, e->Dp, flags, value_net_args(e), e->D, out,
                     bootstrap_rewards, (float)e->cfg.gamma);
}

// Philox key of the action-noise stream: data-parallel ranks must not share it
uint64_t eps_seed(const mobrob_ppo_engine* e) { return e->cfg.seed ^ (0xD1B54A32D192ED03ull * (uint64_t)(e->cfg.rank + 1)); }

void run_gae(mobrob_ppo_engine* e) {
  ProfScope ps(e, MOBROB_K_GAE);
  const double gl = e->cfg.gamma * e->cfg.gae_lambda;  // python: self.gamma * self.gae_lambda (float64)
  hipLaunchKernelGGL(k_gae, dim3(cdiv(e->N, kGaeEnvs)), dim3(kGaeThreads), 0, e->stream, e->rewards, e->values, e->es, e->last_values,
                     e->last_dones, (float)e->cfg.gamma, gl, e->T, e->N, e->adv, e->ret);
}

// gSDE: new exploration matrices for env rows [r0, r0 + n) (and, with `single`, the one matrix predict() uses for foreign batches)
void sde_resample(mobrob_ppo_engine* e, int r0, int n, uint32_t draw, const uint32_t* draw_base, bool single) {
  const int HLA = e->HL * e->A, per = cdiv(HLA, 4);
  hipLaunchKernelGGL(k_sde_resample, dim3(cdiv((n + (single ? 1 : 0)) * per, 256)), dim3(256), 0, e->stream, Pp(e, T_LOGSTD), HLA, e->A, e->sde_mode, r0, n,
                     eps_seed(e) ^ 0x5DE5DE5DE5DE5DEull, draw, draw_base, e->sde_E, single ? e->sde_E1 : (float*)nullptr);
}

// gSDE: variance of `rows` rows of the policy's last hidden activations (the forward has just produced them) into sde_var.
// fresh_table: recompute the std^2 table from log_std first (parameters may have changed since it was last built)
void sde_variance(mobrob_ppo_engine* e, int rows, bool fresh_table, float* lat2 = nullptr) {
  if (fresh_table)
    hipLaunchKernelGGL(k_sde_std2, dim3(cdiv(e->HL * e->A, 256)), dim3(256), 0, e->stream, Pp(e, T_LOGSTD), e->HL, e->A, e->sde_mode, e->sde_S2);
  hipLaunchKernelGGL(k_sde_var, dim3(cdiv(rows, 4)), dim3(256), 0, e->stream, e->hp[e->Lp - 1], e->HL, e->sde_S2, rows, e->HL, e->A, e->sde_var, e->Ap,
                     lat2);
}

// policy forward + sample for rows [r0, r0 + n) of rollout slot t (observations already in the slot).
// draw = Philox draw index of the step (the whole-step callers pass the running counter and advance it).
void act_rows(mobrob_ppo_engine* e, int t, int r0, int n, const float* eps_dev_or_null, uint32_t draw,
              const uint32_t* draw_base, float* clip_out = nullptr) {
  // clip_out: where the clipped actions of the rows go (default: the device staging rows; the pipelined host path
  // passes the caller's pinned buffer -- the sampling epilogue writes them over PCIe itself, no copy kernel)
  ProfScope ps(e, MOBROB_K_ACT);
  if (!clip_out) clip_out = e->clip_act + (size_t)r0 * e->A;
  const size_t row = (size_t)t * e->N + r0;
  const float* X = e->obs + row * e->Dp;
  if (e->fused.enabled) {
    FusedActArgs a{};
    a.X = X; a.rows = n; a.row0 = r0; a.want_pi = 1; a.want_v = 1; a.mu = nullptr; a.ldmu = e->Ap;
    a.v = e->values + row; a.sample = 1; a.A = e->A; a.log_std = Pp(e, T_LOGSTD); a.eps = eps_dev_or_null;
    a.seed = eps_seed(e); a.draw = draw; a.draw_base = draw_base; a.lo = (float)e->cfg.action_low; a.hi = (float)e->cfg.action_high;
    a.act_raw = e->actions + row * e->A; a.act_clip = clip_out; a.logp = e->logp + row;
    fused_launch_act(e->fused, a, e->stream);
    return;
  }
  forward(e, X, n, true, e->mu, true, e->values + row);
  if (e->sde) {
    // collect_rollouts: reset_noise(n_envs) at the start of a rollout and every sde_sample_freq steps (SB3 on_policy_algorithm.py);
    // the rows' matrices at this step's draw index (a function of (env, draw), whichever launch draws them)
    const int freq = e->cfg.sde_sample_freq;
    if (!e->sde_hold && (t == 0 || (freq > 0 && t % freq == 0))) sde_resample(e, r0, n, draw, draw_base, r0 == 0);
    // the std^2 table at the first step of a rollout (every row range's: ranges are enqueued independently; the parameters are
    // fixed from there to the end of the rollout -- and a captured rollout graph must rebuild it on every replay)
    sde_variance(e, n, t == 0);
    hipLaunchKernelGGL(k_sample_sde, dim3(cdiv(n, 4)), dim3(256), 0, e->stream, e->mu, e->Ap, e->hp[e->Lp - 1], e->HL, e->sde_var, e->Ap,
                       e->sde_E, r0, 0, n, e->HL, e->A, (float)e->cfg.action_low, (float)e->cfg.action_high,
                       e->actions + row * e->A, clip_out, e->logp + row);
    return;
  }
  hipLaunchKernelGGL(k_sample, dim3(cdiv(n, 256)), dim3(256), 0, e->stream, e->mu, e->Ap, Pp(e, T_LOGSTD),
                     eps_dev_or_null, n, e->A, (float)e->cfg.action_low, (float)e->cfg.action_high, eps_seed(e),
                     draw, draw_base, e->actions + row * e->A, clip_out, e->logp + row, r0);
}
void act_slot(mobrob_ppo_engine* e, int t, const float* eps_dev_or_null, bool device_counter = false) {
  act_rows(e, t, 0, e->N, eps_dev_or_null, device_counter ? (uint32_t)t : e->draw_counter,
           device_counter ? e->ctr_dev : nullptr);
  if (!device_counter) e->draw_counter++;
}

int upload_obs(mobrob_ppo_engine* e, const float* host, float* dev_rows, int rows) {
  HIPC(hipMemcpy2DAsync(dev_rows, (size_t)e->Dp * 4, host, (size_t)e->D * 4, (size_t)e->D * 4, rows,
                        hipMemcpyHostToDevice, e->stream));
  return MOBROB_OK;
}

int fused_init(mobrob_ppo_engine* e) {
  FusedState& f = e->fused;
  // (every fused kernel's epilogue is tanh: ReLU networks run the generic GEMM chain)
  f.enabled = e->cfg.fast_kernels && e->cfg.activation == MOBROB_ACT_TANH && e->Lp == 2 && e->Lv == 2 && !e->sde &&
              fused_shape_ok(e->D, e->A, e->H1, e->H2, e->G1, e->G2);   // (other depths: generic GEMM chain)
  if (!f.enabled) return MOBROB_OK;
  f.D = e->D; f.Dp = e->Dp; f.A = e->A; f.H = e->H1;
  if ((uint64_t)(e->T + 1) * e->N * e->Dp * 4ull >= (1ull << 32) || (uint64_t)e->T * e->N * e->A * 4ull >= (1ull << 32) ||
      (uint64_t)e->T * e->N * train_rec_width(e->A) * 4ull >= (1ull << 32)) {
    f.enabled = false;  // the fused kernels address rollout rows and training records with 32-bit byte offsets
    return MOBROB_OK;
  }
  const int H = f.H;
  const size_t nW1 = (size_t)(H / 32) * (e->Dp / 8) * 256, nW2 = (size_t)(H / 32) * (H / 8) * 256;
  const size_t nW3f = (size_t)(H / 8) * 256, nW3b = (size_t)(H / 32) * 4 * 256;
  const size_t nW3h = (size_t)(H / 16) * 256;  // 16x16x4 pack of heads <= 16 wide (H = 256 train kernel)
  const size_t per_net = nW1 + 2 * nW2 + nW3f + nW3h + nW3b + 2 * H;
  f.packed_floats = 2 * per_net;
  CHK(dalloc(e, &f.packed, f.packed_floats));
  const int bias_ids[2][3] = {{T_PB1, T_PB2, T_AB}, {T_VB1, T_VB2, T_VB}};
  for (int n = 0; n < 2; ++n) {
    float* p = f.packed + n * per_net;
    f.net[n].W1f = reinterpret_cast<const f32x4*>(p); p += nW1;
    f.net[n].W2f = reinterpret_cast<const f32x4*>(p); p += nW2;
    f.net[n].W3f = reinterpret_cast<const f32x4*>(p); p += nW3f;
    f.net[n].W2b = reinterpret_cast<const f32x4*>(p); p += nW2;
    f.net[n].W3b = reinterpret_cast<const f32x4*>(p); p += nW3b;
    f.net[n].b1s = p; p += H;
    f.net[n].b2s = p; p += H;
    f.net[n].W3h = reinterpret_cast<const f32x4*>(p); p += nW3h;  // last: the H = 64 kernels mirror [W1f, b2s] as one block
    f.net[n].b3 = e->params + e->offs[bias_ids[n][2]];
    f.net[n].head = n == 0 ? e->A : 1;
    f.net[n].W1x = nullptr; f.net[n].W2x = nullptr; f.net[n].W2bx = nullptr;
    f.net[n].W1c = nullptr; f.net[n].W2c = nullptr; f.net[n].W2bc = nullptr; f.net[n].W3c = nullptr; f.net[n].W3bc = nullptr;
  }
  f.max_grid = 256;
  if (H == 64) {
    f.slab_floats = s64_size();
    // one slab per block (k_fused64_train), per tile (k_split64_train) or per two-wave workgroup (k_pair64_train)
    const int tiles_max = cdiv(std::min(e->Bl, e->N * e->T), GR);
    f.pair_nseq_max = std::min(kPairsPerCu * 256 / 2, tiles_max);
    CHK(dalloc(e, &f.slabs, (size_t)std::max(f.max_grid, 2 * f.pair_nseq_max) * f.slab_floats));
    f.lds_bytes = fused64_train_lds_bytes(e->Dp);
    f.lds_act_bytes = fused64_lds_bytes(e->Dp);
    CHK(dalloc(e, &e->epoch_bar, 1024));
    CHK(dalloc(e, &e->epoch_snap, (size_t)3 * e->P));
    CHK(dalloc(e, &e->epoch_consts, (size_t)2 * e->nmb));
    CHK(dalloc(e, &e->epoch_idx, (size_t)e->nmb));
  } else {
    f.slab_floats = slab_size(e->Dp);
    CHK(dalloc(e, &f.slabs, (size_t)f.max_grid * f.slab_floats));
    if (e->cfg.forward_x3 && e->cfg.activation == MOBROB_ACT_TANH && getenv("MOBROB_NO_X3") == nullptr) {
      // x3 packs of the hidden layers of both networks (kernels_fused.h, gemm_x3_r32): rebuilt at the start of every rollout
      for (int n = 0; n < 2; ++n) {
        unsigned* w1 = nullptr; unsigned* w2 = nullptr;
        CHK(dalloc(e, &w1, (size_t)(H / 32) * (e->Dp / 16) * 192 * 4));
        CHK(dalloc(e, &w2, (size_t)(H / 32) * (H / 16) * 192 * 4));
        f.net[n].W1x = w1; f.net[n].W2x = w2;
        unsigned* w2b = nullptr;
        CHK(dalloc(e, &w2b, (size_t)(H / 32) * (H / 16) * 192 * 4));
        f.net[n].W2bx = w2b;
      }
      f.train_x3 = e->A <= 16 && e->Dp != 48 && getenv("MOBROB_NO_TRAIN_X3") == nullptr;
      // the register-chained gradient kernel (kernels_chain.h) takes the same shapes; MOBROB_NO_CHAIN=1 keeps k_fused_train<.., X3>
      f.train_chain = f.train_x3 && getenv("MOBROB_NO_CHAIN") == nullptr;
      if (f.train_chain) {
        const int K1 = (e->Dp + 31) / 32;
        for (int n = 0; n < 2; ++n) {
          unsigned* w1 = nullptr; unsigned* w2 = nullptr; unsigned* w2b = nullptr; float* w3 = nullptr; float* w3b = nullptr;
          CHK(dalloc(e, &w1, (size_t)K1 * 16 * 768));       // [k step][tile 16][3 pieces][64 lanes] x 16 bytes
          CHK(dalloc(e, &w2, (size_t)8 * 16 * 768));
          CHK(dalloc(e, &w2b, (size_t)8 * 16 * 768));
          CHK(dalloc(e, &w3, (size_t)16 * 64 * 4));
          CHK(dalloc(e, &w3b, (size_t)16 * 64 * 4));
          f.net[n].W1c = w1; f.net[n].W2c = w2; f.net[n].W2bc = w2b; f.net[n].W3c = w3; f.net[n].W3bc = w3b;
        }
        f.lds_chain_bytes = chain_lds_bytes(e->Dp);
      }
    }
    CHK(dalloc(e, &f.train_rec, (size_t)e->N * e->T * train_rec_width(e->A)));
    f.lds_bytes = fused_lds_bytes(e->Dp);
    f.lds_act_bytes = fused_lds_act_bytes(e->Dp);
  }
  CHK(dalloc(e, &f.stamps, 32));
  if (!e->plan_only) HIPC(fused_set_lds_attr(f));
  {  // norm-record tables: which (network, block, slot) of the reduction kernel holds which tensor -- emulated here
    // exactly as block_norm_records forms them (per wave: tensors by first occurrence; adjacent equal ones merge)
    const int nb = cdiv(f.slab_floats, 256);
    CHK(dalloc(e, &e->norm_rec_sum, (size_t)2 * nb * kNormRec));
    CHK(dalloc(e, &e->norm_rec_t, (size_t)2 * nb * kNormRec));
    CHK(dalloc(e, &e->fold_idx_dev, (size_t)2 * nb * kNormRec));
    SlabReduceArgs s256{};
    Slab64ReduceArgs s64{};
    for (int i = 0; i < 14; ++i) { s256.offs[i] = e->offs[i]; s64.offs[i] = e->offs[i]; }
    s256.D = e->D; s256.Dp = e->Dp; s256.A = e->A; s256.h16 = e->A <= 16; s256.P = e->P; s256.slab_floats = f.slab_floats;
    s64.D = e->D; s64.A = e->A; s64.P = e->P;
    std::vector<std::vector<int>> per_tensor(13);
    for (int net = 0; net < 2; ++net)
      for (int b = 0; b < nb; ++b) {
        std::vector<int> recs;  // tensors of this block's records, in order
        for (int w = 0; w < 4; ++w) {
          std::vector<int> wave;
          for (int l = 0; l < 64; ++l) {
            const int p = b * 256 + w * 64 + l;
            const int dst = p < f.slab_floats ? (H == 64 ? slab64_to_canonical(s64, net, p) : slab_to_canonical(s256, net, p)) : -1;
            const int t = tensor_of_canonical(e->offs, e->P, dst);
            if (t >= 0 && std::find(wave.begin(), wave.end(), t) == wave.end()) wave.push_back(t);
          }
          for (int t : wave)
            if (recs.empty() || recs.back() != t) recs.push_back(t);
        }
        if ((int)recs.size() > kNormRec) return fail(MOBROB_ERR_INVALID, "norm record table: %zu tensors in one reduction block", recs.size());
        for (size_t k = 0; k < recs.size(); ++k) per_tensor[recs[k]].push_back((net * nb + b) * kNormRec + (int)k);
      }
    std::vector<int> fold;
    for (int t = 0; t < 13; ++t) {
      e->fold_start[t] = (int)fold.size();
      fold.insert(fold.end(), per_tensor[t].begin(), per_tensor[t].end());
    }
    e->fold_start[13] = (int)fold.size();
    if (fold.size() > 1024) return fail(MOBROB_ERR_INVALID, "norm record table too large (%zu)", fold.size());
    if (!e->plan_only)
      HIPC(hipMemcpyAsync(e->fold_idx_dev, fold.data(), fold.size() * sizeof(int), hipMemcpyHostToDevice, e->stream));
    if (!e->plan_only) HIPC(hipStreamSynchronize(e->stream));
  }
  return MOBROB_OK;
}

// 64-wide networks: one independent wave per 32-row tile (kernels_fused64.h)
void fused64_minibatch_grad(mobrob_ppo_engine* e, int mb, int start, int B, float inv_bg) {
  FusedState& f = e->fused;
  Fused64TrainArgs a{};
  a.net[0] = f.net[0]; a.net[1] = f.net[1];
  a.obs = e->obs; a.actions = e->actions; a.A = e->A; a.old_logp = e->logp; a.adv = e->adv; a.ret = e->ret;
  a.rows = e->rows + start; a.count = B; a.log_std = e->params + e->offs[T_LOGSTD];
  a.advstat = e->advstat + 4 * (size_t)mb; a.normalize = e->cfg.normalize_advantage;
  a.clip = (float)e->cfg.clip_range; a.vf_coef = (float)e->cfg.vf_coef; a.ent_coef = (float)e->cfg.ent_coef;
  a.clip_vf = (float)e->clip_vf; a.old_values = e->values;
  a.inv_bg = inv_bg; a.slabs = f.slabs; a.sums = e->grads + e->P; a.stamps = f.stamps;
  const int ntiles = cdiv(B, GR);
  const int grid = 2 * std::min(f.max_grid / 2, cdiv(ntiles, g_train_waves(e->Dp)));
  a.wpack[0] = reinterpret_cast<const float*>(f.net[0].W1f);
  a.wpack[1] = reinterpret_cast<const float*>(f.net[1].W1f);
  // Small minibatches: one workgroup per tile (kernels_split64.h); its per-tile slabs are folded in groups of
  // g_train_waves tiles, which reproduces the block kernel bit for bit as long as that kernel would have given every
  // wave at most one tile.
  const bool split = ntiles <= e->split64_max_tiles && 2 * ntiles <= f.max_grid && ntiles <= (grid / 2) * g_train_waves(e->Dp);
  // Large minibatches: persistent two-wave workgroups, four per CU (kernels_pair64.h); one slab per workgroup.
  const bool pair = !split && e->pair64_min_tiles > 0 && ntiles >= e->pair64_min_tiles;
  const int nseq = std::min(ntiles, f.pair_nseq_max);
  {
    ProfScope ps(e, MOBROB_K_TRAIN_GRAD);
    if (split) split64_launch_train(f, a, ntiles, e->stream);
    else if (pair) pair64_launch_train(f, a, nseq, e->stream);
    else fused64_launch_train(f, a, grid, e->stream);
  }
  ProfScope pr(e, MOBROB_K_GRAD_REDUCE);
  Slab64ReduceArgs s{};
  const int nbseq = (nseq + 1) / 2;  // k_pair64_train: one slab per workgroup of two pairs
  s.slabs = f.slabs; s.nblocks = split ? 2 * ntiles : (pair ? 2 * nbseq : grid); s.group = split ? g_train_waves(e->Dp) : 1; s.grads = e->grads; s.P = e->P;
  for (int i = 0; i < 14; ++i) s.offs[i] = e->offs[i];
  s.D = e->D; s.A = e->A; s.ent_coef = (float)e->cfg.ent_coef; s.b_local = (float)B; s.inv_bg = inv_bg;
  s.sums = e->grads + e->P;
  s.rec_sum = e->use_norm_records ? e->norm_rec_sum : nullptr; s.rec_t = e->norm_rec_t;
  if (pair && nbseq > 128) hipLaunchKernelGGL(k_slab64_reduce_wide, dim3(cdiv(s64_size(), 256), 2), dim3(1024), 0, e->stream, s);
  else hipLaunchKernelGGL(k_slab64_reduce, dim3(cdiv(s64_size(), 256), 2), dim3(256), 0, e->stream, s);
}

// fused minibatch gradient: one persistent kernel + the deterministic slab reduction
void fused_minibatch_grad(mobrob_ppo_engine* e, int mb, int start, int B, float inv_bg) {
  FusedState& f = e->fused;
  FusedTrainArgs a{};
  a.net[0] = f.net[0]; a.net[1] = f.net[1];
  a.obs = e->obs; a.Dp = e->Dp; a.A = e->A; a.rec = f.train_rec; a.RW = train_rec_width(e->A);
  a.rows = e->rows + start; a.count = B; a.log_std = e->params + e->offs[T_LOGSTD];
  a.advstat = e->advstat + 4 * (size_t)mb; a.normalize = e->cfg.normalize_advantage;
  a.clip = (float)e->cfg.clip_range; a.vf_coef = (float)e->cfg.vf_coef; a.ent_coef = (float)e->cfg.ent_coef;
  a.clip_vf = (float)e->clip_vf;
  a.inv_bg = inv_bg; a.slabs = f.slabs; a.slab_floats = f.slab_floats; a.sums = e->grads + e->P;
  a.stamps = f.stamps;
  const int ntiles = cdiv(B, FR);
  const int grid = 2 * std::min(f.max_grid / 2, ntiles);
  // the 8 loss accumulators behind the gradient vector were zeroed by the previous k_adam_pack (or at allocation)
  {
    ProfScope ps(e, MOBROB_K_TRAIN_GRAD);
    fused_launch_train(f, a, grid, e->stream);
  }
  ProfScope pr(e, MOBROB_K_GRAD_REDUCE);
  SlabReduceArgs s{};
  s.slabs = f.slabs; s.slab_floats = f.slab_floats; s.nslabs = grid; s.grads = e->grads; s.P = e->P;
  for (int i = 0; i < 14; ++i) s.offs[i] = e->offs[i];
  s.D = e->D; s.Dp = e->Dp; s.A = e->A; s.h16 = e->A <= 16; s.ent_coef = (float)e->cfg.ent_coef; s.b_local = (float)B; s.inv_bg = inv_bg;
  s.sums = e->grads + e->P;
  s.rec_sum = e->use_norm_records ? e->norm_rec_sum : nullptr; s.rec_t = e->norm_rec_t;
  hipLaunchKernelGGL(k_slab_reduce, dim3(cdiv(f.slab_floats, 256), 2), dim3(256), 0, e->stream, s);
}

// ---- rollout streamer --------------------------------------------------------------------------------
bool is_pinned(const void* p) {
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, p) != hipSuccess) {
    (void)hipGetLastError();  // pageable memory: clear the sticky error
    return false;
  }
  return at.type == hipMemoryTypeHost;
}
// What the SERVED host collector (collect_host_served) may be handed: the rollout kernel's hand-over is built on system-scope
// write-through stores, relaxed flag words and one acquire per step, which is sound for COHERENT (fine-grained) host memory --
// hipHostMalloc(Coherent) and hipHostRegister blocks while HIP_HOST_COHERENT is not 0.  A non-coherent allocation is refused by
// name, and BOTH ends of the range the kernel reads / writes must lie in device-visible host memory (the first byte alone says
// nothing about an [N][D] array).
const char* served_buffer_problem(const void* p, size_t bytes) {
  if (!p || bytes == 0) return "a null or empty buffer";
  const char* hc = getenv("HIP_HOST_COHERENT");
  const bool default_noncoherent = hc != nullptr && atoi(hc) == 0;
  const char* ends[2] = {static_cast<const char*>(p), static_cast<const char*>(p) + bytes - 1};
  for (const char* q : ends) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, q) != hipSuccess) {
      (void)hipGetLastError();
      return q == ends[0] ? "pageable buffers" : "a buffer whose end lies outside the pinned block";
    }
    if (at.type != hipMemoryTypeHost) return "buffers that are not pinned host memory";
    if (at.allocationFlags & hipHostMallocNonCoherent) return "a non-coherent host allocation (hipHostMallocNonCoherent)";
    if (default_noncoherent && !(at.allocationFlags & hipHostMallocCoherent))
      return "host memory that is non-coherent by default (HIP_HOST_COHERENT=0) and was not allocated hipHostMallocCoherent";
  }
  return nullptr;
}
// The per-step calls of the pipelined host path see the same few buffers for a whole rollout: a pointer that was
// found pinned is remembered until the next rollout_begin (the driver query costs about a microsecond each time).
bool is_pinned_cached(mobrob_ppo_engine* e, const void* p) {
  for (const void* q : e->pinned_seen)
    if (q == p) return true;
  if (!is_pinned(p)) return false;
  e->pinned_seen[e->pinned_seen_n++ % 8] = p;
  return true;
}
int streamer_init(mobrob_ppo_engine* e) {
  if (e->cstream) return MOBROB_OK;
  const size_t N = e->N, D = e->D, A = e->A;
  HIPC(hipStreamCreateWithFlags(&e->cstream, hipStreamNonBlocking));
  HIPC(hipEventCreateWithFlags(&e->ev_in, hipEventDisableTiming));
  HIPC(hipEventCreateWithFlags(&e->ev_k, hipEventDisableTiming));
  HIPC(hipEventCreateWithFlags(&e->ev_store, hipEventDisableTiming));
  for (auto& st : e->stage) {
    HIPC(hipHostMalloc((void**)&st.obs, N * D * 4, hipHostMallocDefault));
    HIPC(hipHostMalloc((void**)&st.eps, N * A * 4, hipHostMallocDefault));
    HIPC(hipHostMalloc((void**)&st.rew, N * 4, hipHostMallocDefault));
    HIPC(hipHostMalloc((void**)&st.term, N * D * 4, hipHostMallocDefault));
    HIPC(hipHostMalloc((void**)&st.dones, N, hipHostMallocDefault));
    HIPC(hipHostMalloc((void**)&st.trunc, N, hipHostMallocDefault));
  }
  HIPC(hipHostMalloc((void**)&e->o_raw, N * A * 4, hipHostMallocDefault));
  HIPC(hipHostMalloc((void**)&e->o_clip, N * A * 4, hipHostMallocDefault));
  HIPC(hipHostMalloc((void**)&e->o_val, N * 4, hipHostMallocDefault));
  HIPC(hipHostMalloc((void**)&e->o_lp, N * 4, hipHostMallocDefault));
  return MOBROB_OK;
}
// source pointer for an async H2D: the caller's buffer if it is pinned (zero copy), else a pinned staging copy
template <typename Tp>
const Tp* stage_in(const Tp* user, Tp* staging, size_t count) {
  if (is_pinned(user)) return user;
  memcpy(staging, user, count * sizeof(Tp));
  return staging;
}
int upload_obs_on(mobrob_ppo_engine* e, hipStream_t st, const float* host, float* dev_rows, int rows) {
  HIPC(hipMemcpy2DAsync(dev_rows, (size_t)e->Dp * 4, host, (size_t)e->D * 4, (size_t)e->D * 4, rows,
                        hipMemcpyHostToDevice, st));
  return MOBROB_OK;
}

// net_arch as the config carries it: pi_hidden[0 .. 1], pi_hidden3, pi_hidden_ext[0 .. 4] (likewise vf); a width of 0 ends the list
void cfg_widths(const mobrob_ppo_config_t* c, int* pw, int* vw) {
  pw[0] = c->pi_hidden[0]; pw[1] = c->pi_hidden[1]; pw[2] = c->pi_hidden3;
  vw[0] = c->vf_hidden[0]; vw[1] = c->vf_hidden[1]; vw[2] = c->vf_hidden3;
  for (int i = 3; i < kMaxHidden; ++i) { pw[i] = c->pi_hidden_ext[i - 3]; vw[i] = c->vf_hidden_ext[i - 3]; }
}

static_assert(ACT_TANH == MOBROB_ACT_TANH && ACT_RELU == MOBROB_ACT_RELU && ACT_ELU == MOBROB_ACT_ELU && ACT_LEAKY_RELU == MOBROB_ACT_LEAKY_RELU &&
              ACT_SIGMOID == MOBROB_ACT_SIGMOID && ACT_SOFTPLUS == MOBROB_ACT_SOFTPLUS && ACT_SOFTSIGN == MOBROB_ACT_SOFTSIGN &&
              ACT_HARDTANH == MOBROB_ACT_HARDTANH && ACT_RELU6 == MOBROB_ACT_RELU6 && ACT_SILU == MOBROB_ACT_SILU &&
              ACT_GELU == MOBROB_ACT_GELU && ACT_MISH == MOBROB_ACT_MISH && ACT_COUNT == MOBROB_ACT_COUNT,
              "kernels_generic.h activation codes are the header's");

int check_cfg(const mobrob_ppo_config_t* c) {
  if (c->abi_version != MOBROB_PPO_ABI_VERSION) return fail(MOBROB_ERR_INVALID, "abi_version %d != %d", c->abi_version, MOBROB_PPO_ABI_VERSION);
  if (c->obs_dim < 1 || c->act_dim < 1) return fail(MOBROB_ERR_INVALID, "obs_dim/act_dim must be >= 1");
  {  // one to eight hidden layers per network: [h1, h2, ...] with trailing zeros, every width a positive multiple of 8
    int pw[kMaxHidden], vw[kMaxHidden];
    cfg_widths(c, pw, vw);
    for (const int* w : {pw, vw}) {
      bool ended = false;
      for (int i = 0; i < kMaxHidden; ++i) {
        if (w[i] == 0 && i > 0) { ended = true; continue; }
        if (ended || w[i] < 8 || w[i] % 8)
          return fail(MOBROB_ERR_INVALID, "hidden widths must be positive multiples of 8, one to %d layers per network (a width of 0 ends the list)", kMaxHidden);
      }
    }
  }
  if (c->n_envs < 1 || c->n_steps < 1 || c->batch_size < 1 || c->n_epochs < 1)
    return fail(MOBROB_ERR_INVALID, "n_envs, n_steps, batch_size, n_epochs must be >= 1");
  if (c->world_size < 1 || c->rank < 0 || c->rank >= c->world_size) return fail(MOBROB_ERR_INVALID, "bad rank/world_size");
  if (c->batch_size % c->world_size) return fail(MOBROB_ERR_INVALID, "batch_size must be divisible by world_size");
  if ((int64_t)c->n_envs * c->n_steps > (int64_t)1 << 30) return fail(MOBROB_ERR_INVALID, "rollout too large");
  if (c->use_sde != 0 && c->use_sde != 1) return fail(MOBROB_ERR_INVALID, "use_sde must be 0 or 1");
  if ((c->sde_full_std != 0 && c->sde_full_std != 1) || (c->sde_use_expln != 0 && c->sde_use_expln != 1))
    return fail(MOBROB_ERR_INVALID, "sde_full_std / sde_use_expln must be 0 or 1");
  if (c->activation < 0 || c->activation >= MOBROB_ACT_COUNT) return fail(MOBROB_ERR_INVALID, "activation must be one of MOBROB_ACT_* (0 .. %d)", MOBROB_ACT_COUNT - 1);
  return MOBROB_OK;
}

}  // namespace

extern "C" {

int mobrob_ppo_abi_version(void) { return MOBROB_PPO_ABI_VERSION; }
const char* mobrob_ppo_last_error(void) { return g_err.c_str(); }

void mobrob_ppo_default_config(mobrob_ppo_config_t* c) {
  memset(c, 0, sizeof *c);
  c->abi_version = MOBROB_PPO_ABI_VERSION;
  c->obs_dim = 58; c->act_dim = 12;
  c->pi_hidden[0] = c->pi_hidden[1] = 64;
  c->vf_hidden[0] = c->vf_hidden[1] = 64;
  c->n_envs = 1; c->n_steps = 2048; c->batch_size = 64; c->n_epochs = 10;
  c->gamma = 0.99; c->gae_lambda = 0.95; c->clip_range = 0.2; c->ent_coef = 0.0; c->vf_coef = 0.5;
  c->max_grad_norm = 0.5; c->learning_rate = 3e-4; c->adam_beta1 = 0.9; c->adam_beta2 = 0.999; c->adam_eps = 1e-5;
  c->action_low = -1.0; c->action_high = 1.0;
  c->normalize_advantage = 1; c->seed = 0; c->device_id = 0; c->rank = 0; c->world_size = 1; c->fast_kernels = 1;
  c->rollout_graph = 1;
  c->rollout_persistent = 1;
  c->forward_x3 = 1;
  c->use_sde = 0; c->sde_sample_freq = -1; c->sde_full_std = 1; c->sde_use_expln = 0;
}

void* mobrob_ppo_host_alloc(size_t bytes) {
  void* p = nullptr;
  // coherent (fine-grained) whatever HIP_HOST_COHERENT says: the zero-copy and served collectors hand data over through it
  if (hipHostMalloc(&p, bytes, hipHostMallocCoherent | hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) {
    (void)hipGetLastError();
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
  }
  return p;
}
void mobrob_ppo_host_free(void* p) { if (p) (void)hipHostFree(p); }

int mobrob_ppo_host_register(void* p, size_t bytes) {
  if (!p || bytes == 0) return fail(MOBROB_ERR_INVALID, "host_register: null pointer or empty range");
  hipError_t r = hipHostRegister(p, bytes, hipHostRegisterMapped | hipHostRegisterPortable);
  if (r != hipSuccess) {
    (void)hipGetLastError();
    return fail(MOBROB_ERR_HIP, "hipHostRegister(%p, %zu): %s", p, bytes, hipGetErrorString(r));
  }
  // the zero-copy kernels dereference the HOST address: it must also be the device address of the mapping
  void* d = nullptr;
  r = hipHostGetDevicePointer(&d, p, 0);
  if (r != hipSuccess || d != p) {
    (void)hipGetLastError();
    (void)hipHostUnregister(p);
    return fail(MOBROB_ERR_HIP, "host_register: the registered range is not device visible at its host address (%p -> %p)", p, d);
  }
  return MOBROB_OK;
}
int mobrob_ppo_host_unregister(void* p) {
  if (!p) return fail(MOBROB_ERR_INVALID, "host_unregister: null pointer");
  hipError_t r = hipHostUnregister(p);
  if (r != hipSuccess) {
    (void)hipGetLastError();
    return fail(MOBROB_ERR_HIP, "hipHostUnregister(%p): %s", p, hipGetErrorString(r));
  }
  return MOBROB_OK;
}

}  // extern "C"

namespace {

// host-only: dimensions, parameter offsets, the norm chunk table
int engine_dims(mobrob_ppo_engine* e, const mobrob_ppo_config_t* cfg) {
  e->cfg = *cfg;
  e->D = cfg->obs_dim; e->Dp = padded_obs_dim(cfg->obs_dim); e->A = cfg->act_dim; e->Ap = rup(cfg->act_dim, 8);
  int pw[kMaxHidden], vw[kMaxHidden];   // (check_cfg has validated the pattern)
  cfg_widths(cfg, pw, vw);
  e->Lp = e->Lv = 0;
  for (int l = 0; l < kMaxHidden; ++l) { e->Hp[l] = e->Hv[l] = 0; }
  for (int l = 0; l < kMaxHidden && pw[l] > 0; ++l) e->Hp[e->Lp++] = pw[l];
  for (int l = 0; l < kMaxHidden && vw[l] > 0; ++l) e->Hv[e->Lv++] = vw[l];
  e->HL = e->Hp[e->Lp - 1]; e->GL = e->Hv[e->Lv - 1];
  e->H1 = e->Hp[0]; e->H2 = e->Hp[1]; e->G1 = e->Hv[0]; e->G2 = e->Hv[1];
  e->N = cfg->n_envs; e->T = cfg->n_steps;
  e->Bl = cfg->batch_size / cfg->world_size;
  const int total = e->N * e->T;
  e->nmb = cdiv(total, e->Bl);
  e->rows_max = std::max(e->N, std::min(e->Bl, total));
  int sizes[kMaxTensors] = {0};
  int nt = 0;
  e->sde = cfg->use_sde != 0;
  e->sde_mode = SdeMode{cfg->sde_full_std != 0, cfg->sde_use_expln != 0};
  sizes[nt++] = e->sde ? e->HL * (e->sde_mode.full ? e->A : 1) : e->A;   // log_std ([A]; gSDE: [HL][A], or [HL][1] without full_std)
  for (int l = 0; l < e->Lp; ++l) { e->tPW[l] = nt; sizes[nt++] = e->Hp[l] * (l ? e->Hp[l - 1] : e->D); e->tPB[l] = nt; sizes[nt++] = e->Hp[l]; }
  for (int l = 0; l < e->Lv; ++l) { e->tVW[l] = nt; sizes[nt++] = e->Hv[l] * (l ? e->Hv[l - 1] : e->D); e->tVB[l] = nt; sizes[nt++] = e->Hv[l]; }
  e->tAW = nt; sizes[nt++] = e->A * e->HL; e->tAB = nt; sizes[nt++] = e->A;
  e->tVWh = nt; sizes[nt++] = e->GL; e->tVBh = nt; sizes[nt++] = 1;
  e->ntens = nt;
  e->offs[0] = 0;
  for (int i = 0; i < nt; ++i) e->offs[i + 1] = e->offs[i] + sizes[i];
  for (int i = nt; i < kMaxTensors; ++i) e->offs[i + 1] = e->offs[nt];
  e->P = e->offs[nt];
  e->stats_cap = std::max(64, 4 * e->nmb * cfg->n_epochs);
  // chunk table for the gradient-norm reduction: <= 4096 elements per block, chunks sorted by tensor
  e->chunk_table.clear();
  for (int tt = 0; tt < e->ntens; ++tt)
    for (int s0 = e->offs[tt]; s0 < e->offs[tt + 1]; s0 += 4096)
      e->chunk_table.push_back(NormChunk{tt, s0, std::min(s0 + 4096, e->offs[tt + 1]), 0});
  e->nchunks = (int)e->chunk_table.size();
  if (e->nchunks > 256) return fail(MOBROB_ERR_INVALID, "parameter vector too large for the norm chunk table (%d chunks)", e->nchunks);
  return MOBROB_OK;
}

// carve every device buffer out of the arena (or, with plan_only, just add up the bytes)
int engine_alloc(mobrob_ppo_engine* e) {
  const size_t P = e->P, N = e->N, T = e->T, Dp = e->Dp, A = e->A, Bl = std::min(e->Bl, e->N * e->T), R = e->rows_max;
  CHK(dalloc(e, &e->params, P)); CHK(dalloc(e, &e->grads, P + 8)); CHK(dalloc(e, &e->m, P)); CHK(dalloc(e, &e->v, P));
  CHK(dalloc(e, &e->pW1p, (size_t)e->H1 * Dp)); CHK(dalloc(e, &e->vW1p, (size_t)e->G1 * Dp));
  CHK(dalloc(e, &e->aWp, (size_t)e->Ap * e->HL)); CHK(dalloc(e, &e->vWp, (size_t)8 * e->GL));
  CHK(dalloc(e, &e->obs, (T + 1) * N * Dp)); CHK(dalloc(e, &e->actions, T * N * A));
  CHK(dalloc(e, &e->rewards, T * N)); CHK(dalloc(e, &e->es, T * N)); CHK(dalloc(e, &e->values, (T + 1) * N));
  e->last_values = e->values + T * N;  // V(last_obs) sits behind the stored values: one batched pass covers both
  CHK(dalloc(e, &e->logp, T * N)); CHK(dalloc(e, &e->adv, T * N)); CHK(dalloc(e, &e->ret, T * N));
  CHK(dalloc(e, &e->last_dones, N)); CHK(dalloc(e, &e->prev_dones, N));
  CHK(dalloc(e, &e->dones_tmp, N)); CHK(dalloc(e, &e->clip_act, N * A)); CHK(dalloc(e, &e->rew_tmp, N));
  CHK(dalloc(e, &e->term_obs, N * Dp)); CHK(dalloc(e, &e->term_val, N)); CHK(dalloc(e, &e->eps_dev, R * A));
  CHK(dalloc(e, &e->trunc_dev, N)); CHK(dalloc(e, &e->dones_u8, N)); CHK(dalloc(e, &e->ep_len, N)); CHK(dalloc(e, &e->ep_len2, N)); CHK(dalloc(e, &e->ctr_dev, 2));
  CHK(dalloc(e, &e->rows, T * N)); CHK(dalloc(e, &e->perm_dev, T * N)); CHK(dalloc(e, &e->advstat, (size_t)e->nmb * 4)); CHK(dalloc(e, &e->advbins, (size_t)e->nmb * 2 + 1)); CHK(dalloc(e, &e->expvar_part, (size_t)kEvBlocks * 4));  // + two 32-bit max words
  CHK(dalloc(e, &e->stats, (size_t)e->stats_cap * 8));
  CHK(dalloc(e, &e->Xg, Bl * Dp)); CHK(dalloc(e, &e->actg, Bl * A)); CHK(dalloc(e, &e->lpg, Bl));
  CHK(dalloc(e, &e->advg, Bl)); CHK(dalloc(e, &e->retg, Bl)); CHK(dalloc(e, &e->oldvg, Bl));
  for (int l = 0; l < e->Lp; ++l) CHK(dalloc(e, &e->hp[l], R * e->Hp[l]));
  for (int l = 0; l < e->Lv; ++l) CHK(dalloc(e, &e->hv[l], R * e->Hv[l]));
  CHK(dalloc(e, &e->mu, R * e->Ap)); CHK(dalloc(e, &e->vout, R));
  CHK(dalloc(e, &e->dmu, Bl * e->Ap)); CHK(dalloc(e, &e->dv, Bl * 8));
  for (int l = 0; l < e->Lp; ++l) CHK(dalloc(e, &e->dzp[l], Bl * e->Hp[l]));
  for (int l = 0; l < e->Lv; ++l) CHK(dalloc(e, &e->dzv[l], Bl * e->Hv[l]));
  CHK(dalloc(e, &e->pred_obs, R * Dp)); CHK(dalloc(e, &e->pred_act, R * A));
  if (e->sde) {
    CHK(dalloc(e, &e->sde_E, N * (size_t)e->HL * A)); CHK(dalloc(e, &e->sde_E1, (size_t)e->HL * A));
    CHK(dalloc(e, &e->sde_lat2, Bl * e->HL)); CHK(dalloc(e, &e->sde_gsig, Bl * e->Ap)); CHK(dalloc(e, &e->sde_graw, (size_t)e->HL * A));
    CHK(dalloc(e, &e->sde_S2, (size_t)e->HL * A)); CHK(dalloc(e, &e->sde_var, R * e->Ap));
  }
  CHK(dalloc(e, &e->gstate[0], N * kGoalStateFloats)); CHK(dalloc(e, &e->gstate[1], N * kGoalStateFloats));
  CHK(dalloc(e, &e->ep_stats, kEpStatsDoubles));
  CHK(dalloc(e, &e->chunks_dev, e->chunk_table.size()));
  CHK(dalloc(e, &e->chunk_partial, e->chunk_table.size()));
  CHK(fused_init(e));
  return MOBROB_OK;
}

int check_device(const mobrob_ppo_config_t* cfg) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(MOBROB_ERR_NO_DEVICE, "no HIP device visible: libmobrob_ppo has no CPU fallback");
  if (cfg->device_id < 0 || cfg->device_id >= ndev) return fail(MOBROB_ERR_INVALID, "device_id %d out of range (%d devices)", cfg->device_id, ndev);
  HIPC(hipSetDevice(cfg->device_id));
  hipDeviceProp_t prop;
  HIPC(hipGetDeviceProperties(&prop, cfg->device_id));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(MOBROB_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 (MI355X) only", cfg->device_id, prop.gcnArchName);
  return MOBROB_OK;
}

int engine_create(const mobrob_ppo_config_t* cfg, void* arena, size_t arena_bytes, mobrob_ppo_engine_t** out) {
  CHK(check_cfg(cfg));
  CHK(check_device(cfg));
  auto* e = new mobrob_ppo_engine();
  *out = e;  // so that destroy() can clean up after a partial failure
  if (const char* v = getenv("MOBROB_ROLLOUT64_TILE_MAX")) e->rollout64_tile_max = atoi(v);  // 0: one-wave kernel only
  if (const char* v = getenv("MOBROB_PAIR64_MIN_TILES")) e->pair64_min_tiles = atoi(v);  // 0: block kernel for large minibatches
  if (const char* v = getenv("MOBROB_SPLIT64_MAX_TILES")) e->split64_max_tiles = atoi(v);  // 0: block kernel only (A/B, tests)
  if (const char* v = getenv("MOBROB_EPOCH_KERNEL")) e->epoch_kernel_on = atoi(v) != 0;     // 0: three launches per optimizer step, always
  if (const char* v = getenv("MOBROB_GEMM_TILES")) e->gemm_tiles = atoi(v);                 // generic chain: tiles per wave (launch_gemm)
  CHK(engine_dims(e, cfg));
  HIPC(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
  e->own_stream = true;
  if (arena) {
    if (reinterpret_cast<uintptr_t>(arena) % kArenaAlign) return fail(MOBROB_ERR_INVALID, "arena must be %zu-byte aligned", kArenaAlign);
    e->arena = static_cast<char*>(arena);
    e->arena_bytes = arena_bytes;
  } else {  // one allocation per engine, sized by the same planning pass callers of create_in_arena use
    size_t need = 0;
    CHK(mobrob_ppo_device_bytes(cfg, &need));
    void* q = nullptr;
    HIPC(hipMalloc(&q, need));
    e->allocs.push_back(q);
    e->arena = static_cast<char*>(q);
    e->arena_bytes = need;
  }
  CHK(engine_alloc(e));
  HIPC(hipMemcpyAsync(e->chunks_dev, e->chunk_table.data(), e->chunk_table.size() * sizeof(NormChunk), hipMemcpyHostToDevice, e->stream));
  // `_last_episode_starts` is all-True at _setup_learn (Appendix A.5)
  std::vector<float> ones(e->N, 1.0f);
  HIPC(hipMemcpyAsync(e->prev_dones, ones.data(), (size_t)e->N * 4, hipMemcpyHostToDevice, e->stream));
  HIPC(hipStreamSynchronize(e->stream));
  repack(e);
  if (e->sde) sde_resample(e, 0, e->N, e->draw_counter++, nullptr, true);   // proba_distribution_net samples the first weights at _build
  HIPC(hipStreamSynchronize(e->stream));
  return MOBROB_OK;
}

}  // namespace

extern "C" {

int mobrob_ppo_device_bytes(const mobrob_ppo_config_t* cfg, size_t* bytes) {
  if (!cfg || !bytes) return fail(MOBROB_ERR_INVALID, "null argument");
  CHK(check_cfg(cfg));
  mobrob_ppo_engine plan;  // host-only sizing pass: no device is touched
  plan.plan_only = true;
  CHK(engine_dims(&plan, cfg));
  CHK(engine_alloc(&plan));
  *bytes = plan.arena_used;
  return MOBROB_OK;
}

void* mobrob_ppo_device_alloc(int32_t device_id, size_t bytes) {
  void* p = nullptr;
  if (hipSetDevice(device_id) != hipSuccess || hipMalloc(&p, bytes) != hipSuccess) {
    fail(MOBROB_ERR_HIP, "device_alloc(%zu bytes) failed", bytes);
    return nullptr;
  }
  return p;
}
void mobrob_ppo_device_free(void* p) { if (p) (void)hipFree(p); }

int mobrob_ppo_create(const mobrob_ppo_config_t* cfg, mobrob_ppo_engine_t** out) {
  if (!cfg || !out) return fail(MOBROB_ERR_INVALID, "null argument");
  return engine_create(cfg, nullptr, 0, out);
}

int mobrob_ppo_create_in_arena(const mobrob_ppo_config_t* cfg, void* arena, size_t arena_bytes, mobrob_ppo_engine_t** out) {
  if (!cfg || !out || !arena) return fail(MOBROB_ERR_INVALID, "null argument");
  return engine_create(cfg, arena, arena_bytes, out);
}

void mobrob_ppo_destroy(mobrob_ppo_engine_t* e) {
  if (!e) return;
  (void)hipStreamSynchronize(e->stream);
  (void)mobrob_ppo_comm_destroy(e);
  (void)mobrob_ppo_oneshot_close(e);
  prof_resolve(e);
  if (e->cstream) {
    (void)hipStreamSynchronize(e->cstream);
    for (auto& st : e->stage) {
      (void)hipHostFree(st.obs); (void)hipHostFree(st.eps); (void)hipHostFree(st.rew); (void)hipHostFree(st.term);
      (void)hipHostFree(st.dones); (void)hipHostFree(st.trunc);
    }
    (void)hipHostFree(e->o_raw); (void)hipHostFree(e->o_clip); (void)hipHostFree(e->o_val); (void)hipHostFree(e->o_lp);
    (void)hipEventDestroy(e->ev_in); (void)hipEventDestroy(e->ev_k); (void)hipEventDestroy(e->ev_store);
    (void)hipStreamDestroy(e->cstream);
  }
  for (hipEvent_t ev : e->ev_part)
    if (ev) (void)hipEventDestroy(ev);
  if (e->srv_flags) (void)hipHostFree(e->srv_flags);
  if (e->srv_abort) (void)hipFree(e->srv_abort);
  if (e->epoch_stage) (void)hipHostFree(e->epoch_stage);
  if (e->epoch_err_host) (void)hipHostFree(e->epoch_err_host);
  if (e->epoch_ev) (void)hipEventDestroy(e->epoch_ev);
  if (e->ro_exec) (void)hipGraphExecDestroy(e->ro_exec);
  if (e->ro_graph) (void)hipGraphDestroy(e->ro_graph);
  if (e->vstream) {
    (void)hipStreamSynchronize(e->vstream);
    for (auto ev : e->ev_chunks) (void)hipEventDestroy(ev);
    (void)hipEventDestroy(e->ev_vdone);
    (void)hipStreamDestroy(e->vstream);
  }
  for (auto ev : e->ev_pool) (void)hipEventDestroy(ev);
  for (void* p : e->allocs) (void)hipFree(p);
  if (e->own_stream && e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
}

int mobrob_ppo_set_stream(mobrob_ppo_engine_t* e, void* s) {
  if (!e) return fail(MOBROB_ERR_INVALID, "null engine");
  HIPC(hipStreamSynchronize(e->stream));
  if (e->ro_exec) { (void)hipGraphExecDestroy(e->ro_exec); e->ro_exec = nullptr; }
  if (e->ro_graph) { (void)hipGraphDestroy(e->ro_graph); e->ro_graph = nullptr; }
  if (e->own_stream && e->stream) HIPC(hipStreamDestroy(e->stream));
  if (s == nullptr) {
    HIPC(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    e->own_stream = true;
  } else {
    e->stream = static_cast<hipStream_t>(s);
    e->own_stream = false;
  }
  return MOBROB_OK;
}

int mobrob_ppo_synchronize(mobrob_ppo_engine_t* e) {
  if (!e) return fail(MOBROB_ERR_INVALID, "null engine");
  HIPC(hipStreamSynchronize(e->stream));
  return check_async_error(e);
}

int64_t mobrob_ppo_param_count(const mobrob_ppo_engine_t* e) { return e ? e->P : -1; }

int mobrob_ppo_get_params(mobrob_ppo_engine_t* e, float* out, int64_t n) {
  if (!e || !out || n != e->P) return fail(MOBROB_ERR_INVALID, "get_params: n=%lld, expected %d", (long long)n, e ? e->P : -1);
  HIPC(hipMemcpyAsync(out, e->params, (size_t)n * 4, hipMemcpyDeviceToHost, e->stream));
  HIPC(hipStreamSynchronize(e->stream));
  return MOBROB_OK;
}
int mobrob_ppo_set_params(mobrob_ppo_engine_t* e, const float* in, int64_t n) {
  if (!e || !in || n != e->P) return fail(MOBROB_ERR_INVALID, "set_params: n=%lld, expected %d", (long long)n, e ? e->P : -1);
  HIPC(hipMemcpyAsync(e->params, in, (size_t)n * 4, hipMemcpyHostToDevice, e->stream));
  repack(e);
  HIPC(hipStreamSynchronize(e->stream));
  return MOBROB_OK;
}
int mobrob_ppo_get_optimizer_state(mobrob_ppo_engine_t* e, float* m, float* v, int64_t n, int64_t* step) {
  if (!e || n != e->P) return fail(MOBROB_ERR_INVALID, "get_optimizer_state: bad size");
  if (m) HIPC(hipMemcpyAsync(m, e->m, (size_t)n * 4, hipMemcpyDeviceToHost, e->stream));
  if (v) HIPC(hipMemcpyAsync(v, e->v, (size_t)n * 4, hipMemcpyDeviceToHost, e->stream));
  HIPC(hipStreamSynchronize(e->stream));
  if (step) *step = e->adam_step;
  return MOBROB_OK;
}
int mobrob_ppo_set_optimizer_state(mobrob_ppo_engine_t* e, const float* m, const float* v, int64_t n, int64_t step) {
  if (!e || !m || !v || n != e->P || step < 0) return fail(MOBROB_ERR_INVALID, "set_optimizer_state: bad argument");
  HIPC(hipMemcpyAsync(e->m, m, (size_t)n * 4, hipMemcpyHostToDevice, e->stream));
  HIPC(hipMemcpyAsync(e->v, v, (size_t)n * 4, hipMemcpyHostToDevice, e->stream));
  HIPC(hipStreamSynchronize(e->stream));
  e->adam_step = step;
  return MOBROB_OK;
}

// ---- rollout ---------------------------------------------------------------------------------------
int mobrob_ppo_rollout_begin(mobrob_ppo_engine_t* e) {
  if (!e) return fail(MOBROB_ERR_INVALID, "null engine");
  e->t = 0;
  e->rollout_ready = false; e->train_rec_valid = false;
  e->nparts = 0;
  for (auto& q : e->pinned_seen) q = nullptr;
  e->pinned_seen_n = 0;
  e->draw_ro0 = e->draw_counter;
  for (int p = 0; p < MOBROB_MAX_PARTS; ++p) {
    e->part_act_t[p] = e->part_store_t[p] = 0;
    e->part_obs_t[p] = -1;
  }
  return MOBROB_OK;
}

int mobrob_ppo_act(mobrob_ppo_engine_t* e, const float* obs, const float* eps, float* a_raw, float* a_clip,
                   float* values, float* logp) {
  if (!e || !obs) return fail(MOBROB_ERR_INVALID, "act: null argument");
  if (e->t >= e->T) return fail(MOBROB_ERR_STATE, "act: rollout buffer full (t=%d, n_steps=%d)", e->t, e->T);
  CHK(streamer_init(e));
  const size_t N = e->N, A = e->A, D = e->D;
  // Pinned fast path: every buffer of the call is hipHostMalloc memory -> the GPU reads / writes it in place
  // (k_pull_rows, k_copy_f32) on the compute stream; one stream, no events, one synchronisation.
  if (!eps && is_pinned(obs) && (!a_raw || is_pinned(a_raw)) && (!a_clip || is_pinned(a_clip)) &&
      (!values || is_pinned(values)) && (!logp || is_pinned(logp))) {
    hipLaunchKernelGGL(k_pull_rows, dim3(cdiv((int)(N * e->Dp), 256)), dim3(256), 0, e->stream, obs,
                       e->obs + (size_t)e->t * N * e->Dp, (int)N, (int)D, e->Dp);
    act_slot(e, e->t, nullptr);
    auto push = [&](const float* dev, float* host, size_t n) {
      if (host) hipLaunchKernelGGL(k_copy_f32, dim3(cdiv((int)n, 256)), dim3(256), 0, e->stream, dev, host, (int)n);
    };
    push(e->clip_act, a_clip, N * A);
    push(e->actions + (size_t)e->t * N * A, a_raw, N * A);
    push(e->values + (size_t)e->t * N, values, N);
    push(e->logp + (size_t)e->t * N, logp, N);
    HIPC(hipGetLastError());
    HIPC(hipStreamSynchronize(e->stream));
    return MOBROB_OK;
  }
  auto& st = e->stage[e->stage_i];
  // H2D on the side stream (pinned source -> truly asynchronous), compute stream waits on the event
  const float* src = stage_in(obs, st.obs, N * D);
  CHK(upload_obs_on(e, e->cstream, src, e->obs + (size_t)e->t * N * e->Dp, e->N));
  if (eps) {
    const float* es = stage_in(eps, st.eps, N * A);
    HIPC(hipMemcpyAsync(e->eps_dev, es, N * A * 4, hipMemcpyHostToDevice, e->cstream));
  }
  HIPC(hipEventRecord(e->ev_in, e->cstream));
  HIPC(hipStreamWaitEvent(e->stream, e->ev_in, 0));
  act_slot(e, e->t, eps ? e->eps_dev : nullptr);
  HIPC(hipEventRecord(e->ev_k, e->stream));
  HIPC(hipStreamWaitEvent(e->cstream, e->ev_k, 0));
  // D2H of the requested outputs: straight into pinned user buffers, else through pinned staging
  struct Out { float* user; float* stagebuf; const float* dev; size_t n; };
  const Out outs[4] = {{a_clip, e->o_clip, e->clip_act, N * A},
                       {a_raw, e->o_raw, e->actions + (size_t)e->t * N * A, N * A},
                       {values, e->o_val, e->values + (size_t)e->t * N, N},
                       {logp, e->o_lp, e->logp + (size_t)e->t * N, N}};
  bool staged[4] = {false, false, false, false};
  for (int i = 0; i < 4; ++i) {
    if (!outs[i].user) continue;
    staged[i] = !is_pinned(outs[i].user);
    HIPC(hipMemcpyAsync(staged[i] ? outs[i].stagebuf : outs[i].user, outs[i].dev, outs[i].n * 4, hipMemcpyDeviceToHost,
                        e->cstream));
  }
  HIPC(hipStreamSynchronize(e->cstream));
  for (int i = 0; i < 4; ++i)
    if (outs[i].user && staged[i]) memcpy(outs[i].user, outs[i].stagebuf, outs[i].n * 4);
  return MOBROB_OK;
}

int mobrob_ppo_store(mobrob_ppo_engine_t* e, const float* rewards, const uint8_t* dones, const uint8_t* truncated,
                     const float* terminal_obs) {
  if (!e || !rewards || !dones) return fail(MOBROB_ERR_INVALID, "store: null argument");
  if (e->t >= e->T) return fail(MOBROB_ERR_STATE, "store: rollout buffer full");
  CHK(streamer_init(e));
  const size_t N = e->N, D = e->D;
  if (is_pinned(rewards) && is_pinned(dones) && (!truncated || is_pinned(truncated)) &&
      (!terminal_obs || is_pinned(terminal_obs))) {
    // Pinned fast path: the kernels read the caller's buffers in place.  The caller may overwrite them only after
    // the next act() has returned (act synchronises the compute stream) -- the order every rollout loop has.
    bool any = false;
    if (truncated && terminal_obs)
      for (size_t i = 0; i < N; ++i) any |= truncated[i] != 0;
    if (any) {
      hipLaunchKernelGGL(k_pull_rows, dim3(cdiv((int)(N * e->Dp), 256)), dim3(256), 0, e->stream, terminal_obs, e->term_obs,
                         (int)N, (int)D, e->Dp);
      HIPC(hipMemcpyAsync(e->trunc_dev, truncated, N, hipMemcpyHostToDevice, e->stream));
      value_flagged(e, e->term_obs, e->trunc_dev, e->term_val);
    }
    hipLaunchKernelGGL(k_store_step, dim3(cdiv(e->N, 256)), dim3(256), 0, e->stream, rewards, e->prev_dones,
                       any ? e->trunc_dev : nullptr, e->term_val, (float)e->cfg.gamma, e->N,
                       e->rewards + (size_t)e->t * N, e->es + (size_t)e->t * N);
    hipLaunchKernelGGL(k_u8_to_f32, dim3(cdiv(e->N, 256)), dim3(256), 0, e->stream, dones, e->prev_dones, e->N);
    HIPC(hipGetLastError());
    e->t++;
    return MOBROB_OK;
  }
  auto& st = e->stage[e->stage_i];
  // Fully asynchronous: the scalars are staged in pinned memory (the caller may reuse its buffers immediately),
  // copied on the side stream and consumed by the compute stream behind an event; nothing here waits for the GPU.
  // The staging slot is reused two steps later, after act() of the next step has synchronised the side stream.
  memcpy(st.rew, rewards, N * 4);
  memcpy(st.dones, dones, N);
  HIPC(hipMemcpyAsync(e->rew_tmp, st.rew, N * 4, hipMemcpyHostToDevice, e->cstream));
  HIPC(hipMemcpyAsync(e->dones_u8, st.dones, N, hipMemcpyHostToDevice, e->cstream));
  bool any_trunc = false;
  if (truncated && terminal_obs)
    for (size_t i = 0; i < N; ++i) any_trunc |= truncated[i] != 0;
  if (any_trunc) {
    memcpy(st.trunc, truncated, N);
    memcpy(st.term, terminal_obs, N * D * 4);
    HIPC(hipMemcpyAsync(e->trunc_dev, st.trunc, N, hipMemcpyHostToDevice, e->cstream));
    CHK(upload_obs_on(e, e->cstream, st.term, e->term_obs, e->N));
  }
  HIPC(hipEventRecord(e->ev_store, e->cstream));
  HIPC(hipStreamWaitEvent(e->stream, e->ev_store, 0));
  if (any_trunc) value_flagged(e, e->term_obs, e->trunc_dev, e->term_val);
  hipLaunchKernelGGL(k_store_step, dim3(cdiv(e->N, 256)), dim3(256), 0, e->stream, e->rew_tmp, e->prev_dones,
                     any_trunc ? e->trunc_dev : nullptr, e->term_val, (float)e->cfg.gamma, e->N,
                     e->rewards + (size_t)e->t * N, e->es + (size_t)e->t * N);
  hipLaunchKernelGGL(k_u8_to_f32, dim3(cdiv(e->N, 256)), dim3(256), 0, e->stream, e->dones_u8, e->prev_dones, e->N);
  // the next step's H2D of rew_tmp/dones_u8 must not overtake these kernels
  HIPC(hipEventRecord(e->ev_k, e->stream));
  HIPC(hipStreamWaitEvent(e->cstream, e->ev_k, 0));
  HIPC(hipGetLastError());
  e->stage_i ^= 1;
  e->t++;
  return MOBROB_OK;
}

// ---- pipelined host-env rollout: the same arithmetic as act()/store(), one contiguous row range at a time ----
namespace {
int part_range(mobrob_ppo_engine* e, int part, int nparts, int* r0, int* n) {
  if (nparts < 1 || nparts > MOBROB_MAX_PARTS || nparts > e->N || part < 0 || part >= nparts)
    return fail(MOBROB_ERR_INVALID, "part %d of %d (1..%d parts, at most one per env)", part, nparts, MOBROB_MAX_PARTS);
  if (e->nparts == 0) e->nparts = nparts;
  if (e->nparts != nparts) return fail(MOBROB_ERR_STATE, "nparts changed from %d to %d inside a rollout", e->nparts, nparts);
  *r0 = (int)((int64_t)e->N * part / nparts);
  *n = (int)((int64_t)e->N * (part + 1) / nparts) - *r0;
  return MOBROB_OK;
}
}  // namespace

int mobrob_ppo_act_part(mobrob_ppo_engine_t* e, int32_t part, int32_t nparts, const float* obs, float* a_clip) {
  if (!e || !obs || !a_clip) return fail(MOBROB_ERR_INVALID, "act_part: null argument");
  int r0, n;
  CHK(part_range(e, part, nparts, &r0, &n));
  const int t = e->part_act_t[part];
  if (t >= e->T) return fail(MOBROB_ERR_STATE, "act_part: rollout buffer full (part %d, t=%d)", part, t);
  if (t != e->part_store_t[part]) return fail(MOBROB_ERR_STATE, "act_part: part %d acted on step %d but has not stored it", part, t - 1);
  if (!is_pinned_cached(e, obs) || !is_pinned_cached(e, a_clip))
    return fail(MOBROB_ERR_INVALID, "act_part needs device-visible pinned buffers (mobrob_ppo_host_alloc)");
  if (!e->ev_part[part]) HIPC(hipEventCreateWithFlags(&e->ev_part[part], hipEventDisableTiming));
  const size_t D = e->D, A = e->A;
  if (e->part_obs_t[part] != t)  // not already pulled by the preceding store_part(next_obs)
    hipLaunchKernelGGL(k_pull_rows, dim3(cdiv(n * e->Dp, 256)), dim3(256), 0, e->stream, obs + (size_t)r0 * D,
                       e->obs + ((size_t)t * e->N + r0) * e->Dp, n, (int)D, e->Dp);
  act_rows(e, t, r0, n, nullptr, e->draw_ro0 + (uint32_t)t, nullptr, a_clip + (size_t)r0 * A);
  HIPC(hipGetLastError());
  HIPC(hipEventRecord(e->ev_part[part], e->stream));
  e->part_act_t[part] = t + 1;
  if (e->draw_counter < e->draw_ro0 + (uint32_t)t + 1) e->draw_counter = e->draw_ro0 + (uint32_t)t + 1;
  return MOBROB_OK;
}

int mobrob_ppo_wait_part(mobrob_ppo_engine_t* e, int32_t part) {
  if (!e || part < 0 || part >= MOBROB_MAX_PARTS || !e->ev_part[part])
    return fail(MOBROB_ERR_STATE, "wait_part: part %d has no act_part in flight", part);
  HIPC(hipEventSynchronize(e->ev_part[part]));
  return MOBROB_OK;
}

int mobrob_ppo_store_part(mobrob_ppo_engine_t* e, int32_t part, int32_t nparts, const float* rewards,
                          const uint8_t* dones, const uint8_t* truncated, const float* terminal_obs,
                          const float* next_obs) {
  if (!e || !rewards || !dones) return fail(MOBROB_ERR_INVALID, "store_part: null argument");
  int r0, n;
  CHK(part_range(e, part, nparts, &r0, &n));
  const int t = e->part_store_t[part];
  if (t + 1 != e->part_act_t[part]) return fail(MOBROB_ERR_STATE, "store_part: part %d has no acted step to store (t=%d)", part, t);
  if (!is_pinned_cached(e, rewards) || !is_pinned_cached(e, dones) || (truncated && !is_pinned_cached(e, truncated)) ||
      (terminal_obs && !is_pinned_cached(e, terminal_obs)) || (next_obs && !is_pinned_cached(e, next_obs)))
    return fail(MOBROB_ERR_INVALID, "store_part needs device-visible pinned buffers (mobrob_ppo_host_alloc)");
  bool any = false;
  if (truncated && terminal_obs)
    for (int i = r0; i < r0 + n; ++i) any |= truncated[i] != 0;
  const size_t o = (size_t)t * e->N + r0;
  StorePullArgs a{};
  a.rew_in = rewards + r0; a.dones = dones + r0;
  a.trunc = any ? truncated + r0 : nullptr; a.term_obs = any ? terminal_obs + (size_t)r0 * e->D : nullptr;
  a.vn = value_net_args(e);
  a.D = e->D; a.Dp = e->Dp; a.n = n; a.gamma = (float)e->cfg.gamma;
  a.prev_dones = e->prev_dones + r0; a.rew_out = e->rewards + o; a.es_out = e->es + o; a.term_val = e->term_val + r0;
  if (next_obs) {  // slot t+1 exists for every t < T (slot T holds the last observations)
    a.next_obs = next_obs + (size_t)r0 * e->D;
    a.obs_slot = e->obs + ((size_t)(t + 1) * e->N + r0) * e->Dp;
    e->part_obs_t[part] = t + 1;
  }
  const size_t sm = (size_t)(e->D + value_net_width_sum(e) + 32) * sizeof(float);
  hipLaunchKernelGGL(k_store_pull_part, dim3(cdiv(n, kPartRows)), dim3(256), sm, e->stream, a);
  HIPC(hipGetLastError());
  e->part_store_t[part] = t + 1;
  int tmin = e->T;
  for (int p = 0; p < nparts; ++p) tmin = std::min(tmin, e->part_store_t[p]);
  e->t = tmin;
  return MOBROB_OK;
}

namespace {
// defined behind the device-rollout code it shares (enqueue of the persistent kernel's chunks and the overlapped value passes)
int collect_host_served(mobrob_ppo_engine* e, mobrob_env_step_range_fn step_range, void* env, int nparts, float* obs,
                        float* actions_clipped, float* rewards, uint8_t* dones, uint8_t* truncated, float* terminal_obs, bool* served);
}  // namespace

// The whole pipelined rollout in one call: rollout_begin, n_steps x nparts x (wait_part, env step of the range,
// store_part + act_part), finish_rollout -- the collector loop of SB3's collect_rollouts as native code, driving a
// native vectorised environment through one function pointer.
int mobrob_ppo_collect_host(mobrob_ppo_engine_t* e, mobrob_env_step_range_fn step_range, void* env, int32_t nparts,
                            float* obs, float* actions_clipped, float* rewards, uint8_t* dones, uint8_t* truncated,
                            float* terminal_obs) {
  if (!e || !step_range || !obs || !actions_clipped || !rewards || !dones || !truncated || !terminal_obs)
    return fail(MOBROB_ERR_INVALID, "collect_host: null argument");
  CHK(mobrob_ppo_rollout_begin(e));
  {  // fused engines (256-wide x3, 64-wide): the persistent rollout kernel serves the host environment (no launch, no event per step)
    bool served = false;
    const int rc = collect_host_served(e, step_range, env, nparts, obs, actions_clipped, rewards, dones, truncated, terminal_obs, &served);
    if (served || rc != MOBROB_OK) return rc;
  }
  for (int p = 0; p < nparts; ++p) CHK(mobrob_ppo_act_part(e, p, nparts, obs, actions_clipped));
  const bool timing = getenv("MOBROB_COLLECT_TIMING") != nullptr;
  // Two host threads (round 5, OPT-IN: MOBROB_COLLECT_THREADS=1).  The single-thread loop below spends, per vector step of 4096 envs at
  // two parts (MOBROB_COLLECT_TIMING=1, profiles/r5/host_path_timing.txt): 55 us stepping the simulator, 12 us in HIP calls (two
  // launches and an event record per part), 16 - 21 us waiting for the policy's actions.  With a DRIVER thread that owns the stream
  // -- it turns "part p stepped" into store_part + act_part and "act of part p finished" (hipEventQuery, no blocking wait) into
  // "part p ready" -- the calling thread only steps the environment.  Same kernels, same per-part order on the one stream: the rollout
  // is the single-thread loop's bit for bit.  MEASURED (whole iteration, same box, alternating): 198.2 / 198.9 ms single thread,
  // 200.0 - 203.4 ms with the driver thread (16, 15 or 14 env threads): what bounds a part's cycle is sim -> launch latency -> two small
  // kernels -> sim, and the spinning driver takes a core from the simulator's team.  Kept for hosts with cores to spare; off by default.
  const bool threaded = nparts >= 2 && !timing && getenv("MOBROB_COLLECT_THREADS") && atoi(getenv("MOBROB_COLLECT_THREADS")) != 0;
  if (threaded) {
    struct Shared {
      std::atomic<int> ready[MOBROB_MAX_PARTS];     // acts of part p the GPU has finished (actions of step ready - 1 are in host memory)
      std::atomic<int> simdone[MOBROB_MAX_PARTS];   // steps of part p the simulator has finished
      std::atomic<int> ntrunc[MOBROB_MAX_PARTS];    // truncated rows of the part's newest step
      std::atomic<int> err{0};
    } sh;
    for (int p = 0; p < MOBROB_MAX_PARTS; ++p) { sh.ready[p] = 0; sh.simdone[p] = 0; sh.ntrunc[p] = 0; }
    std::string driver_msg;
    int driver_rc = MOBROB_OK;
    const int T = e->T;
    auto driver_body = [&] {
      if (hipSetDevice(e->cfg.device_id) != hipSuccess) { driver_rc = MOBROB_ERR_HIP; driver_msg = "collect_host driver: hipSetDevice failed"; sh.err = 1; return; }
      int acted[MOBROB_MAX_PARTS], synced[MOBROB_MAX_PARTS], stored[MOBROB_MAX_PARTS];
      for (int p = 0; p < nparts; ++p) { acted[p] = 1; synced[p] = 0; stored[p] = 0; }
      for (;;) {
        bool all_done = true, progress = false;
        for (int p = 0; p < nparts && !sh.err.load(std::memory_order_relaxed); ++p) {
          if (synced[p] < acted[p]) {                       // the part's newest act: finished?
            const hipError_t q = hipEventQuery(e->ev_part[p]);
            if (q == hipSuccess) {
              synced[p] = acted[p];
              sh.ready[p].store(synced[p], std::memory_order_release);
              progress = true;
            } else if (q != hipErrorNotReady) {
              driver_rc = MOBROB_ERR_HIP; driver_msg = std::string("collect_host driver: ") + hipGetErrorString(q); sh.err = 1;
            }
          }
          if (stored[p] < T && sh.simdone[p].load(std::memory_order_acquire) > stored[p]) {   // the part was stepped: store, act again
            const int nt = sh.ntrunc[p].load(std::memory_order_relaxed);
            int rc = mobrob_ppo_store_part(e, p, nparts, rewards, dones, nt ? truncated : nullptr, nt ? terminal_obs : nullptr, obs);
            ++stored[p];
            if (rc == MOBROB_OK && stored[p] < T) { rc = mobrob_ppo_act_part(e, p, nparts, obs, actions_clipped); ++acted[p]; }
            if (rc != MOBROB_OK) { driver_rc = rc; driver_msg = g_err; sh.err = 1; }
            progress = true;
          }
          if (stored[p] < T) all_done = false;
        }
        if (all_done || sh.err.load(std::memory_order_relaxed)) break;
        if (!progress) __builtin_ia32_pause();
      }
    };
    std::thread driver;
    try {
      driver = std::thread(driver_body);
    } catch (...) {   // no thread to be had: nothing has been stepped yet, so the single-thread loop below takes over from here
      driver = std::thread();
    }
    if (driver.joinable()) {
    int sim_rc = MOBROB_OK;
    for (int t = 0; t < T && !sh.err.load(std::memory_order_relaxed); ++t) {
      for (int p = 0; p < nparts; ++p) {
        int r0, n;
        if (part_range(e, p, nparts, &r0, &n) != MOBROB_OK) { sim_rc = MOBROB_ERR_INVALID; sh.err = 1; break; }
        while (sh.ready[p].load(std::memory_order_acquire) < t + 1 && !sh.err.load(std::memory_order_relaxed)) __builtin_ia32_pause();
        if (sh.err.load(std::memory_order_relaxed)) break;
        const int32_t ntrunc = step_range(env, r0, r0 + n, actions_clipped, obs, rewards, dones, truncated, terminal_obs);
        if (ntrunc < 0) { sim_rc = MOBROB_ERR_STATE; sh.err = 1; break; }
        sh.ntrunc[p].store(ntrunc, std::memory_order_relaxed);
        sh.simdone[p].store(t + 1, std::memory_order_release);
      }
    }
    driver.join();
    if (driver_rc != MOBROB_OK) return fail(driver_rc, "%s", driver_msg.c_str());
    if (sim_rc != MOBROB_OK) return fail(sim_rc, "collect_host: the environment's step_range failed");
    return mobrob_ppo_finish_rollout(e, obs, dones);
    }
  }
  double tw = 0, te = 0, tq = 0;
  auto now = [] { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3; };
  for (int t = 0; t < e->T; ++t) {
    for (int p = 0; p < nparts; ++p) {
      int r0, n;
      CHK(part_range(e, p, nparts, &r0, &n));
      const double t0 = timing ? now() : 0;
      CHK(mobrob_ppo_wait_part(e, p));
      const double t1 = timing ? now() : 0;
      const int32_t ntrunc = step_range(env, r0, r0 + n, actions_clipped, obs, rewards, dones, truncated, terminal_obs);
      const double t2 = timing ? now() : 0;
      if (ntrunc < 0) return fail(MOBROB_ERR_STATE, "collect_host: the environment's step_range returned %d", ntrunc);
      CHK(mobrob_ppo_store_part(e, p, nparts, rewards, dones, ntrunc ? truncated : nullptr,
                                ntrunc ? terminal_obs : nullptr, obs));
      if (t + 1 < e->T) CHK(mobrob_ppo_act_part(e, p, nparts, obs, actions_clipped));
      if (timing) { tw += t1 - t0; te += t2 - t1; tq += now() - t2; }
    }
  }
  if (timing)
    fprintf(stderr, "[collect_host] per step: wait %.1f us, env %.1f us, enqueue %.1f us (%d parts)\n", tw / e->T, te / e->T,
            tq / e->T, nparts);
  return mobrob_ppo_finish_rollout(e, obs, dones);
}

int mobrob_ppo_finish_rollout(mobrob_ppo_engine_t* e, const float* last_obs, const uint8_t* dones) {
  if (!e || !last_obs || !dones) return fail(MOBROB_ERR_INVALID, "finish_rollout: null argument");
  if (e->t != e->T) return fail(MOBROB_ERR_STATE, "finish_rollout: %d of %d steps stored", e->t, e->T);
  CHK(streamer_init(e));
  const size_t N = e->N;
  auto& st = e->stage[e->stage_i];
  float* slot = e->obs + (size_t)e->T * N * e->Dp;
  const float* src = stage_in(last_obs, st.obs, N * e->D);
  CHK(upload_obs_on(e, e->cstream, src, slot, e->N));
  memcpy(st.dones, dones, N);
  HIPC(hipMemcpyAsync(e->dones_u8, st.dones, N, hipMemcpyHostToDevice, e->cstream));
  HIPC(hipEventRecord(e->ev_in, e->cstream));
  HIPC(hipStreamWaitEvent(e->stream, e->ev_in, 0));
  hipLaunchKernelGGL(k_u8_to_f32, dim3(cdiv(e->N, 256)), dim3(256), 0, e->stream, e->dones_u8, e->last_dones, e->N);
  forward(e, slot, e->N, false, nullptr, true, e->last_values);
  run_gae(e);
  HIPC(hipStreamSynchronize(e->stream));
  e->rollout_ready = true; e->train_rec_valid = false;
  return MOBROB_OK;
}

int mobrob_ppo_compute_gae(mobrob_ppo_engine_t* e) {
  if (!e) return fail(MOBROB_ERR_INVALID, "null engine");
  run_gae(e);
  HIPC(hipStreamSynchronize(e->stream));
  e->rollout_ready = true; e->train_rec_valid = false;
  return MOBROB_OK;
}

int mobrob_ppo_x3_mode(const mobrob_ppo_engine_t* e) {
  if (!e || !e->fused.enabled || e->fused.net[0].W2x == nullptr) return 0;
  return 1 | (e->fused.train_x3 ? 2 : 0) | (e->fused.train_x3 && e->fused.train_chain ? 4 : 0);
}

int mobrob_ppo_update_mode(const mobrob_ppo_engine_t* e) { return e ? e->last_update_mode : 0; }

int mobrob_ppo_explained_variance(mobrob_ppo_engine_t* e, double* out) {
  if (!e || !out) return fail(MOBROB_ERR_INVALID, "explained_variance: null argument");
  if (!e->rollout_ready) return fail(MOBROB_ERR_STATE, "explained_variance: rollout not finished");
  const int n = e->N * e->T;
  hipLaunchKernelGGL(k_explained_variance_partials, dim3(kEvBlocks), dim3(256), 0, e->stream, e->values, e->ret, n, e->expvar_part);
  double part[kEvBlocks * 4];
  HIPC(hipMemcpyAsync(part, e->expvar_part, sizeof part, hipMemcpyDeviceToHost, e->stream));
  HIPC(hipStreamSynchronize(e->stream));
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  for (int b = 0; b < kEvBlocks; ++b)
    for (int k = 0; k < 4; ++k) s[k] += part[b * 4 + k];
  const double var_y = s[1] / n - (s[0] / n) * (s[0] / n), var_d = s[3] / n - (s[2] / n) * (s[2] / n);
  *out = var_y > 0.0 ? 1.0 - var_d / var_y : NAN;  // SB3: nan when the returns do not vary
  return MOBROB_OK;
}

int mobrob_ppo_mark_rollout_ready(mobrob_ppo_engine_t* e) {
  if (!e) return fail(MOBROB_ERR_INVALID, "null engine");
  e->rollout_ready = true; e->train_rec_valid = false;
  e->t = e->T;
  return MOBROB_OK;
}

namespace {
// enqueue one whole device-resident rollout (T steps + last values + GAE) on the engine stream
uint64_t env_seed_of(const mobrob_ppo_engine* e) {
  return e->cfg.seed ^ (0x9E3779B97F4A7C15ull * (uint64_t)(e->cfg.rank + 1));
}

// One persistent launch for all T steps (policy forward + sample + env + store), then the value network over all
// stored observations in one batched pass, then GAE (kernels_rollout.h).
bool rollout_persistent_ok(const mobrob_ppo_engine* e) {
  return e->cfg.rollout_persistent && e->fused.enabled;  // both fused widths (256: kernels_rollout.h top, 64: bottom)
}
// side stream and events of the overlapped value pass: created OUTSIDE any stream capture (resource creation is
// not a capturable operation)
int rollout_side_stream_init(mobrob_ppo_engine* e) {
  if (!e->vstream) {
    HIPC(hipStreamCreateWithFlags(&e->vstream, hipStreamNonBlocking));
    HIPC(hipEventCreateWithFlags(&e->ev_vdone, hipEventDisableTiming));
  }
  while ((int)e->ev_chunks.size() < e->T / 16 + 2) {  // chunks are >= 16 steps
    hipEvent_t ev;
    HIPC(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    e->ev_chunks.push_back(ev);
  }
  return MOBROB_OK;
}

int enqueue_rollout_persistent(mobrob_ppo_engine* e, const mobrob_ppo_engine::RolloutSpec& sp) {
  const int N = e->N, Dp = e->Dp, T = e->T;
  const size_t slot = (size_t)N * Dp;
  HIPC(hipMemcpyAsync(e->obs, e->obs + (size_t)T * slot, slot * 4, hipMemcpyDeviceToDevice, e->stream));
  RolloutArgs a{};
  a.pi = e->fused.net[0];
  a.log_std = Pp(e, T_LOGSTD); a.seed = eps_seed(e); a.draw_base = e->ctr_dev;
  a.lo = (float)e->cfg.action_low; a.hi = (float)e->cfg.action_high;
  a.kind = sp.kind; a.env_seed = env_seed_of(e); a.step_base = e->ctr_dev + 1;
  a.p_term = sp.p_term; a.time_limit = sp.time_limit; a.goal = sp.goal;
  a.bt = BootArgs{Pp(e, T_VW1), Pp(e, T_VB1), Pp(e, T_VW2), Pp(e, T_VB2), Pp(e, T_VW), Pp(e, T_VB), e->G1, e->G2,
                  (float)e->cfg.gamma, e->term_val};  // (fused path: two tanh layers)
  a.N = N; a.D = e->D; a.A = e->A;
  a.obs = e->obs; a.actions = e->actions; a.logp = e->logp; a.rewards = e->rewards; a.es = e->es;
  a.term_obs = e->term_obs; a.trunc = e->trunc_dev; a.clip_act = e->clip_act;
  a.ep_len = e->ep_len; a.prev_dones = e->prev_dones; a.gstate = e->gstate[0]; a.ep_stats = e->ep_stats;
  if (e->fused.H == GH) {  // 64-wide nets (kernels_rollout.h, bottom half)
    const int nwv = rollout64_waves(Dp);
    const int tiles = cdiv(N, 32);
    const bool tile_kernel = tiles <= e->rollout64_tile_max;  // one workgroup per tile while every tile gets a CU of its own
    // A rollout of <= 192 tiles leaves CUs idle for thousands of dependent steps: like the 256-wide path, cut it into chunks
    // and value the observations of a finished chunk on the side stream while the next chunk rolls out.  Worth the extra
    // launches only when the value pass is more than a few launches' worth of work.
    const bool overlap = tile_kernel && tiles <= 192 && (size_t)T * N >= ((size_t)1 << 18);
    const int chunk = overlap ? std::max(16, cdiv(T, 20)) : T;
    {
      ProfScope ps(e, MOBROB_K_ENV);
      for (int t0 = 0; t0 < T; t0 += chunk) {
        a.t0 = t0; a.t1 = std::min(T, t0 + chunk);
        if (tile_kernel) {
          FUSED_DISPATCH_DP(Dp, hipLaunchKernelGGL((k_rollout64_tile<DPc>), dim3(tiles), dim3(256), rollout64_tile_lds_bytes(Dp),
                                                   e->stream, a));
        } else {
          FUSED_DISPATCH_DP(Dp, hipLaunchKernelGGL((k_rollout64_persistent<DPc>), dim3(cdiv(tiles, nwv)), dim3(nwv * 64),
                                                   rollout64_lds_bytes(Dp), e->stream, a));
        }
        if (overlap && a.t1 < T) {  // observations [t0, t1) are final
          hipEvent_t ev = e->ev_chunks[t0 / chunk];
          HIPC(hipEventRecord(ev, e->stream));
          HIPC(hipStreamWaitEvent(e->vstream, ev, 0));
          fused_forward(e->fused, e->obs + (size_t)t0 * N * Dp, (a.t1 - t0) * N, false, nullptr, e->Ap, true, e->values + (size_t)t0 * N,
                        e->vstream);
        }
      }
    }
    hipLaunchKernelGGL(k_add_counters, dim3(1), dim3(64), 0, e->stream, e->ctr_dev, (uint32_t)T, (uint32_t)T);
    HIPC(hipMemcpyAsync(e->last_dones, e->prev_dones, (size_t)N * 4, hipMemcpyDeviceToDevice, e->stream));
    {
      ProfScope ps(e, MOBROB_K_ACT);  // V of the last chunk and V(last_obs) (obs[T]; values[T*N..] = last_values)
      const int done_rows = overlap ? ((T - 1) / chunk) * chunk * N : 0;
      forward(e, e->obs + (size_t)done_rows * Dp, (T + 1) * N - done_rows, false, nullptr, true, e->values + done_rows);
      if (overlap && done_rows > 0) {  // join the side stream before GAE
        HIPC(hipEventRecord(e->ev_vdone, e->vstream));
        HIPC(hipStreamWaitEvent(e->stream, e->ev_vdone, 0));
      }
    }
    run_gae(e);
    return MOBROB_OK;
  }
  // The rollout blocks (32 envs each, ~100 KB of LDS) leave CUs idle when N < 32 * 256; the value pass of the steps
  // already finished runs there at the same time: the rollout is cut into chunks, chunk c's value pass is enqueued
  // on a second stream behind an event and overlaps the rollout of chunk c+1.
  if (!e->fused.train_x3) pack_x3_all(e);  // otherwise every optimizer step (apply_adam) and set_params keep the x3 packs current
  const int rblocks = cdiv(N, 32);
  const bool overlap = rblocks <= 192;                      // otherwise the rollout itself fills the device
  const int chunk = overlap ? std::max(16, cdiv(T, 20)) : T;
  const int vgrid_max = overlap ? std::max(32, 256 - rblocks) : 256;
  auto value_pass = [&](hipStream_t st, int r0, int r1, int grid_max) {  // rows [r0, r1) of obs -> values
    FUSED_DISPATCH_DP(Dp, hipLaunchKernelGGL((k_value_batch<DPc>), dim3(std::min(grid_max, cdiv(r1 - r0, FR))),
                                             dim3(FTHREADS), e->fused.lds_bytes, st, e->fused.net[1],
                                             e->obs + (size_t)r0 * Dp, r1 - r0, e->values + r0));
  };
  {
    ProfScope ps(e, MOBROB_K_ENV);
    for (int t0 = 0; t0 < T; t0 += chunk) {
      a.t0 = t0; a.t1 = std::min(T, t0 + chunk);
      // x3 engines: the eight-wave form with W2's leading pieces stationary in registers (kernels_rollout.h, S8);
      // MOBROB_ROLLOUT_S8=0 keeps the four-wave form (A/B and the bit-equality test of the two)
      static const bool s8_on = !kRolloutStationary && !(getenv("MOBROB_ROLLOUT_S8") && atoi(getenv("MOBROB_ROLLOUT_S8")) == 0);
      const bool s8 = s8_on && a.pi.W2x != nullptr;
      if (s8 && a.kind == 1) {
        FUSED_DISPATCH_DP(Dp, hipLaunchKernelGGL((k_rollout_persistent<DPc, 1, true>), dim3(rblocks), dim3(kRolloutThreads),
                                                 rollout_lds_bytes(Dp, true), e->stream, a));
      } else if (s8) {
        FUSED_DISPATCH_DP(Dp, hipLaunchKernelGGL((k_rollout_persistent<DPc, 2, true>), dim3(rblocks), dim3(kRolloutThreads),
                                                 rollout_lds_bytes(Dp, true), e->stream, a));
      } else if (a.kind == 1) {
        FUSED_DISPATCH_DP(Dp, hipLaunchKernelGGL((k_rollout_persistent<DPc, 1>), dim3(rblocks), dim3(kRolloutThreads),
                                                 rollout_lds_bytes(Dp), e->stream, a));
      } else {
        FUSED_DISPATCH_DP(Dp, hipLaunchKernelGGL((k_rollout_persistent<DPc, 2>), dim3(rblocks), dim3(kRolloutThreads),
                                                 rollout_lds_bytes(Dp), e->stream, a));
      }
      if (overlap && a.t1 < T) {  // observations [t0, t1) are final: value them on the side stream
        hipEvent_t ev = e->ev_chunks[t0 / chunk];
        HIPC(hipEventRecord(ev, e->stream));
        HIPC(hipStreamWaitEvent(e->vstream, ev, 0));
        value_pass(e->vstream, t0 * N, a.t1 * N, vgrid_max);
      }
    }
  }
  hipLaunchKernelGGL(k_add_counters, dim3(1), dim3(64), 0, e->stream, e->ctr_dev, (uint32_t)T, (uint32_t)T);
  HIPC(hipMemcpyAsync(e->last_dones, e->prev_dones, (size_t)N * 4, hipMemcpyDeviceToDevice, e->stream));
  {
    ProfScope ps(e, MOBROB_K_ACT);  // V of the last chunk and V(last_obs) (obs[T]; values[T*N..] = last_values)
    const int done_rows = overlap ? ((T - 1) / chunk) * chunk * N : 0;
    value_pass(e->stream, done_rows, (T + 1) * N, 256);  // the rollout is over: the whole device
    if (overlap && done_rows > 0) {  // join the side stream before GAE
      HIPC(hipEventRecord(e->ev_vdone, e->vstream));
      HIPC(hipStreamWaitEvent(e->stream, e->ev_vdone, 0));
    }
  }
  run_gae(e);
  return MOBROB_OK;
}

int enqueue_rollout(mobrob_ppo_engine* e, const mobrob_ppo_engine::RolloutSpec& sp) {
  if (rollout_persistent_ok(e)) return enqueue_rollout_persistent(e, sp);
  const int N = e->N, Dp = e->Dp, per = Dp / 4;
  const size_t slot = (size_t)N * Dp;
  const uint64_t env_seed = env_seed_of(e);
  // the previous rollout's last observation is this rollout's first
  HIPC(hipMemcpyAsync(e->obs, e->obs + (size_t)e->T * slot, slot * 4, hipMemcpyDeviceToDevice, e->stream));
  const BootNetArgs bt = boot_args(e);
  const size_t sm = env_step_lds_bytes(Dp, value_net_width_sum(e));
  for (int t = 0; t < e->T; ++t) {
    act_slot(e, t, nullptr, true);
    {
      ProfScope ps(e, MOBROB_K_ENV);
      // env step + rollout_buffer.add scalars + time-limit bootstrap of the (rare) truncated rows in one launch
      if (sp.kind == 1) {
        hipLaunchKernelGGL(k_env_step_store, dim3(cdiv(N * per, 256)), dim3(256), sm, e->stream, env_seed, (uint32_t)t,
                           e->ctr_dev + 1, N, e->D, Dp, sp.p_term, sp.time_limit, e->ep_len, e->ep_len2,
                           e->obs + (size_t)(t + 1) * slot, e->term_obs, e->prev_dones, e->dones_tmp, e->trunc_dev,
                           e->rewards + (size_t)t * N, e->es + (size_t)t * N, bt);
        std::swap(e->ep_len, e->ep_len2);
      } else {
        GoalEnvArgs g{};
        g.seed = env_seed; g.step_rel = (uint32_t)t; g.step_base = e->ctr_dev + 1;
        g.N = N; g.D = e->D; g.Dp = Dp; g.A = e->A; g.p = sp.goal;
        g.act = e->clip_act; g.st_in = e->gstate[0]; g.st_out = e->gstate[1];
        g.obs_next = e->obs + (size_t)(t + 1) * slot; g.term_obs = e->term_obs;
        g.prev_dones = e->prev_dones; g.next_dones = e->dones_tmp; g.trunc = e->trunc_dev;
        g.rew_out = e->rewards + (size_t)t * N; g.es_out = e->es + (size_t)t * N; g.ep_stats = e->ep_stats;
        hipLaunchKernelGGL(k_goal_env_step_store, dim3(cdiv(N * per, 256)), dim3(256), sm, e->stream, g, bt);
        std::swap(e->gstate[0], e->gstate[1]);
      }
    }
    std::swap(e->prev_dones, e->dones_tmp);
  }
  hipLaunchKernelGGL(k_add_counters, dim3(1), dim3(64), 0, e->stream, e->ctr_dev, (uint32_t)e->T, (uint32_t)e->T);
  HIPC(hipMemcpyAsync(e->last_dones, e->prev_dones, (size_t)N * 4, hipMemcpyDeviceToDevice, e->stream));
  forward(e, e->obs + (size_t)e->T * slot, N, false, nullptr, true, e->last_values);
  run_gae(e);
  return MOBROB_OK;
}

// device-resident rollout of either env kind: (re)start the env if needed, then replay / enqueue the T-step loop
int collect_device(mobrob_ppo_engine* e, const mobrob_ppo_engine::RolloutSpec& sp) {
  const int N = e->N, Dp = e->Dp, per = Dp / 4;
  const size_t slot = (size_t)N * Dp;
  if (e->env_started != sp.kind) {
    float* last = e->obs + (size_t)e->T * slot;  // reset writes the "previous last observation"
    if (sp.kind == 1) {
      hipLaunchKernelGGL(k_env_reset, dim3(cdiv(N * per, 256)), dim3(256), 0, e->stream, env_seed_of(e), N, e->D, Dp, last,
                         e->ep_len);
    } else {
      hipLaunchKernelGGL(k_goal_env_reset, dim3(cdiv(N * per, 256)), dim3(256), 0, e->stream, env_seed_of(e), N, e->D, Dp,
                         sp.goal, e->gstate[0], last);
      HIPC(hipMemsetAsync(e->ep_stats, 0, kEpStatsDoubles * sizeof(double), e->stream));
      e->ep_ring_read = 0;
    }
    std::vector<float> ones(N, 1.0f);  // a fresh env starts every episode: `_last_episode_starts` all True
    HIPC(hipMemcpyAsync(e->prev_dones, ones.data(), (size_t)N * 4, hipMemcpyHostToDevice, e->stream));
    HIPC(hipStreamSynchronize(e->stream));
    e->env_started = sp.kind;
  }
  e->rollout_ready = false; e->train_rec_valid = false;
  if (rollout_persistent_ok(e)) CHK(rollout_side_stream_init(e));
  // Graph replay: every kernel argument of the T-step loop is fixed (slot pointers, ping-pong buffers with even T,
  // counters relative to device-resident bases), so the loop is captured once and replayed per rollout.
  // The persistent path is ~45 launches on two streams: enqueued eagerly (the launches hide behind the 18 ms of
  // GPU work; a captured multi-stream graph was both slower to launch and, after many capture / destroy cycles in
  // one process, crashed inside the runtime).  Only the per-step path (thousands of launches) is replayed as a graph.
  const bool use_graph = e->cfg.rollout_graph && e->T % 2 == 0 && !rollout_persistent_ok(e);
  if (!use_graph) {
    CHK(enqueue_rollout(e, sp));
  } else {
    if (e->ro_exec == nullptr || memcmp(&e->ro_spec, &sp, sizeof sp) != 0) {
      HIPC(hipStreamSynchronize(e->stream));  // never destroy an executable graph that may still be running
      if (e->ro_exec) { (void)hipGraphExecDestroy(e->ro_exec); e->ro_exec = nullptr; }
      if (e->ro_graph) { (void)hipGraphDestroy(e->ro_graph); e->ro_graph = nullptr; }
      const bool prof = e->prof_on;
      e->prof_on = false;  // event records cannot be part of the captured graph
      hipError_t be = hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal);
      if (be != hipSuccess) {  // e.g. a stream that does not support capture: run eagerly from now on
        (void)hipGetLastError();
        e->prof_on = prof;
        e->cfg.rollout_graph = 0;
        CHK(enqueue_rollout(e, sp));
        HIPC(hipGetLastError());
        e->t = e->T;
        e->rollout_ready = true; e->train_rec_valid = false;
        return MOBROB_OK;
      }
      const int rc = enqueue_rollout(e, sp);
      hipError_t ce = hipStreamEndCapture(e->stream, &e->ro_graph);
      e->prof_on = prof;
      if (rc != MOBROB_OK) return rc;
      if (ce != hipSuccess) return fail(MOBROB_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(ce));
      HIPC(hipGraphInstantiate(&e->ro_exec, e->ro_graph, nullptr, nullptr, 0));
      memcpy(&e->ro_spec, &sp, sizeof sp);
    }
    ProfScope ps(e, MOBROB_K_ACT);  // with graph replay the ACT scope covers the whole rollout (forward+env+GAE)
    HIPC(hipGraphLaunch(e->ro_exec, e->stream));
  }
  HIPC(hipGetLastError());
  e->t = e->T;
  e->rollout_ready = true; e->train_rec_valid = false;
  return MOBROB_OK;
}

// Host environments SERVED by the persistent rollout kernel (round 5; kernels_rollout.h, KIND 3).  The launch-per-step collector above
// pays, per row range and step, two kernel launches, an event, the event's completion latency and the weight stream of a fresh
// k_fused_act -- 74 - 76 us of GPU-side time per vector step of 4096 envs against 10 us for the same arithmetic inside the device
// rollout.  Here the device rollout's own kernel runs the policy (weights stationary, S8) and its env phase is the host's: a
// workgroup writes its rows' clipped actions into the caller's pinned buffer, raises its flag word in pinned memory and polls the
// host's word for its row range; the host waits for the flags of a range, steps it, raises its word.  No HIP call inside the step
// loop.  Same Philox counters, same forward / sampling / storage / bootstrap code as the device rollout; against the launch-per-step
// collector the buffers agree to float32 rounding (its k_fused_act runs the policy on the f32 pipe, the rollout kernel on the bf16
// pipe with split operands: the same relation the device rollout has to its per-step form), what the kernel only moves -- clipped
// actions to the host, rewards / observations from it -- is exact (tests/test_engine_gpu.py::test_served_host_rollout_...).
// Conditions (else *served stays false and the caller runs the launch-per-step loop): 256-wide x3 engine with the eight-wave rollout
// kernel, or (round 6) a 64-wide engine within the tile kernel's range (k_rollout64_tile<.., 3>: the shape of every reference config);
// whole 32-row tiles per row range, every workgroup resident at once (tiles <= CUs: a waiting workgroup never yields its CU),
// coherent pinned buffers (served_buffer_problem), no announced co-tenant of the device.  MOBROB_COLLECT_SERVER=0 switches it off; MOBROB_SERVER_TIMEOUT_S (default 60) bounds every wait on either side.
int collect_host_served(mobrob_ppo_engine* e, mobrob_env_step_range_fn step_range, void* env, int nparts, float* obs,
                        float* actions_clipped, float* rewards, uint8_t* dones, uint8_t* truncated, float* terminal_obs, bool* served) {
  *served = false;
  const int mode = getenv("MOBROB_COLLECT_SERVER") ? atoi(getenv("MOBROB_COLLECT_SERVER")) : 1;   // 0 off, 1 when possible, 2 required (tests); read per rollout
  static const bool s8_on = !kRolloutStationary && !(getenv("MOBROB_ROLLOUT_S8") && atoi(getenv("MOBROB_ROLLOUT_S8")) == 0);
  const int N = e->N, T = e->T, Dp = e->Dp;
  const int rblocks = cdiv(N, 32);
  if (mode == 0) return MOBROB_OK;
  const char* why = nullptr;
  int cus = 0;
  const bool wide = e->fused.enabled && e->fused.H == FH;   // 256-wide: k_rollout_persistent<.., 3, S8>; 64-wide: k_rollout64_tile<.., 3>
  if (!rollout_persistent_ok(e)) why = "the fused persistent rollout kernels are off for this engine";
  else if (wide && (!s8_on || e->fused.net[0].W2x == nullptr)) why = "not a 256-wide x3 engine with the eight-wave rollout kernel";
  else if (!wide && (e->fused.H != GH || rblocks > e->rollout64_tile_max)) why = "not a 64-wide engine within the tile kernel's range";
  else if (getenv("MOBROB_COLLECT_TIMING") || (getenv("MOBROB_COLLECT_THREADS") && atoi(getenv("MOBROB_COLLECT_THREADS")) != 0)) why = "an instrumented / threaded collector was asked for";
  // a 32-row tile must not straddle two row ranges; ONE range takes any number of environments (the reference YAMLs: 2 - 16, half a tile)
  else if (nparts < 1 || nparts > MOBROB_MAX_PARTS || (nparts > 1 && (N % nparts != 0 || (N / nparts) % 32 != 0))) why = "row ranges are not whole 32-row tiles";
  else if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->cfg.device_id) != hipSuccess || rblocks > cus) why = "more tiles than compute units";
  // Other tenants of the device that this process cannot count: ranks rehearsing data parallelism on ONE device, a CU mask.  Every
  // workgroup of a served rollout must be resident at once (a waiting workgroup never yields its CU), so with them the launch-per-step
  // collector -- which always works -- is the one that runs.  (A tenant nobody announced is caught by the residency check below.)
  else if (getenv("MOBROB_DP_SAME_DEVICE") && atoi(getenv("MOBROB_DP_SAME_DEVICE")) != 0 && e->cfg.world_size > 1) why = "several data-parallel ranks share this device (MOBROB_DP_SAME_DEVICE)";
  else if (getenv("HSA_CU_MASK") || getenv("ROC_GLOBAL_CU_MASK")) why = "a compute-unit mask is set (HSA_CU_MASK / ROC_GLOBAL_CU_MASK)";
  else {
    const size_t N_ = (size_t)N;
    const struct { const void* p; size_t bytes; } bufs[6] = {{obs, N_ * e->D * 4}, {actions_clipped, N_ * e->A * 4}, {rewards, N_ * 4},
                                                              {dones, N_}, {truncated, N_}, {terminal_obs, N_ * e->D * 4}};
    for (const auto& b : bufs)
      if (!why) why = served_buffer_problem(b.p, b.bytes);
  }
  // Every workgroup of a served rollout must be resident at once, and it keeps its compute unit until the host has stepped all n_steps:
  // engines of ONE process that collect at the same time (a fleet's threads) share a DEVICE's compute units through that device's
  // counter -- the one that does not fit takes the launch-per-step path instead of queueing behind a kernel that waits for a host.
  constexpr int kLeaseDevices = 64;
  static std::atomic<int> cus_serving[kLeaseDevices];   // zero-initialised (static storage)
  std::atomic<int>& dev_lease = cus_serving[(unsigned)e->cfg.device_id % kLeaseDevices];
  struct Lease {
    std::atomic<int>& c; int n; bool held;
    ~Lease() { if (held) c.fetch_sub(n); }
  } lease{dev_lease, rblocks, false};
  if (!why) {
    if (dev_lease.fetch_add(rblocks) + rblocks > cus) { dev_lease.fetch_sub(rblocks); why = "the device's compute units are serving another engine's rollout"; }
    else lease.held = true;
  }
  if (why) return mode == 2 ? fail(MOBROB_ERR_STATE, "collect_host: MOBROB_COLLECT_SERVER=2 but %s", why) : MOBROB_OK;
  CHK(rollout_side_stream_init(e));
  CHK(streamer_init(e));
  constexpr int kHostWords = 16 * MOBROB_MAX_PARTS;
  constexpr int kStampWords = 64 * 8 * 2;   // -DMOBROB_SERVE_STAMPS builds: 64 steps x 8 stamps (long long) of workgroup 0
  const int fb = (rblocks + 15) / 16 * 16;
  if (!e->srv_flags || e->srv_blocks < fb) {
    if (e->srv_flags) (void)hipHostFree(e->srv_flags);
    e->srv_flags = nullptr;
    HIPC(hipHostMalloc((void**)&e->srv_flags, (size_t)(fb + kHostWords + 16 + kStampWords) * sizeof(unsigned), hipHostMallocCoherent | hipHostMallocMapped));
    e->srv_blocks = fb;
  }
  if (!e->srv_abort) HIPC(hipMalloc((void**)&e->srv_abort, 256));   // abort word | one 8-byte relay word per row range
  unsigned* gpu_flag = e->srv_flags;
  unsigned* host_flag = e->srv_flags + e->srv_blocks;
  int* err_word = reinterpret_cast<int*>(host_flag + kHostWords);
  memset(e->srv_flags, 0, (size_t)(e->srv_blocks + kHostWords + 16 + kStampWords) * sizeof(unsigned));
  HIPC(hipMemsetAsync(e->srv_abort, 0, 256, e->stream));
  const double timeout_s = getenv("MOBROB_SERVER_TIMEOUT_S") ? atof(getenv("MOBROB_SERVER_TIMEOUT_S")) : 60.0;

  // slot 0 <- the environments' current observations (the kernel reads its first tile from the slot, like the device rollout)
  hipLaunchKernelGGL(k_pull_rows, dim3(cdiv(N * Dp, 256)), dim3(256), 0, e->stream, obs, e->obs, N, e->D, Dp);
  if (wide && !e->fused.train_x3) pack_x3_all(e);
  RolloutArgs a{};
  a.pi = e->fused.net[0];
  a.log_std = Pp(e, T_LOGSTD); a.seed = eps_seed(e); a.draw_base = nullptr; a.draw0 = e->draw_ro0;
  a.lo = (float)e->cfg.action_low; a.hi = (float)e->cfg.action_high;
  a.kind = 3; a.env_seed = 0; a.step_base = nullptr;
  a.bt = BootArgs{Pp(e, T_VW1), Pp(e, T_VB1), Pp(e, T_VW2), Pp(e, T_VB2), Pp(e, T_VW), Pp(e, T_VB), e->G1, e->G2,
                  (float)e->cfg.gamma, e->term_val};
  a.N = N; a.D = e->D; a.A = e->A;
  a.obs = e->obs; a.actions = e->actions; a.logp = e->logp; a.rewards = e->rewards; a.es = e->es;
  a.term_obs = e->term_obs; a.trunc = e->trunc_dev; a.clip_act = actions_clipped;   // (the caller's pinned buffer)
  a.ep_len = e->ep_len; a.prev_dones = e->prev_dones; a.gstate = e->gstate[0]; a.ep_stats = e->ep_stats;
  a.h_obs = obs; a.h_rew = rewards; a.h_done = dones; a.h_trunc = truncated; a.h_term = terminal_obs;
  a.h_gpu_flag = gpu_flag; a.h_host_flag = host_flag; a.h_error = err_word; a.abort_dev = e->srv_abort;
  a.rows_per_part = N / nparts; a.timeout_ticks = (long long)(timeout_s * 1e8);

  // From the first KIND-3 launch on, EVERY way out of this function that is not the normal one tells the (possibly resident, possibly
  // waiting) workgroups to stop first: they would otherwise spin for the whole timeout and the next synchronising call with them.
  auto give_up = [&](void) {
    for (int p = 0; p < nparts; ++p) __atomic_store_n(reinterpret_cast<unsigned long long*>(&host_flag[16 * p]), 0xFFFFFFFFull, __ATOMIC_RELEASE);
    (void)hipStreamSynchronize(e->stream);
    (void)hipStreamSynchronize(e->vstream);
  };
#define SRV_HIPC(expr)                                                                                          \
  do {                                                                                                          \
    hipError_t _e = (expr);                                                                                     \
    if (_e != hipSuccess) {                                                                                     \
      give_up();                                                                                                \
      return fail(MOBROB_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);   \
    }                                                                                                           \
  } while (0)
  // chunks of the step loop on the compute stream, V(obs) of a finished chunk on the side stream (as enqueue_rollout_persistent)
  const bool overlap = wide ? rblocks <= 192 : (rblocks <= 192 && (size_t)T * N >= ((size_t)1 << 18));   // (the device rollouts' rules)
  const int chunk = overlap ? std::max(16, cdiv(T, 20)) : T;
  const int vgrid_max = overlap ? std::max(32, 256 - rblocks) : 256;
  auto value_pass = [&](hipStream_t st, int r0, int r1, int grid_max) {
    if (wide) {
      FUSED_DISPATCH_DP(Dp, hipLaunchKernelGGL((k_value_batch<DPc>), dim3(std::min(grid_max, cdiv(r1 - r0, FR))), dim3(FTHREADS),
                                               e->fused.lds_bytes, st, e->fused.net[1], e->obs + (size_t)r0 * Dp, r1 - r0, e->values + r0));
    } else {
      fused_forward(e->fused, e->obs + (size_t)r0 * Dp, r1 - r0, false, nullptr, e->Ap, true, e->values + r0, st);
    }
  };
  if (!wide)   // the served tile kernel's dynamic LDS exceeds 64 KB at 64 observation columns (per device and cheap: set per rollout)
    FUSED_DISPATCH_DP(Dp, HIPC(hipFuncSetAttribute(reinterpret_cast<const void*>(k_rollout64_tile<DPc, 3>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   (int)rollout64_tile_lds_bytes(Dp, true))));
  {
    ProfScope ps(e, MOBROB_K_ENV);
    for (int t0 = 0; t0 < T; t0 += chunk) {
      a.t0 = t0; a.t1 = std::min(T, t0 + chunk);
      if (wide) {
        FUSED_DISPATCH_DP(Dp, hipLaunchKernelGGL((k_rollout_persistent<DPc, 3, true>), dim3(rblocks), dim3(kRolloutThreads),
                                                 rollout_lds_bytes(Dp, true), e->stream, a));
      } else {
        FUSED_DISPATCH_DP(Dp, hipLaunchKernelGGL((k_rollout64_tile<DPc, 3>), dim3(rblocks), dim3(256), rollout64_tile_lds_bytes(Dp, true), e->stream, a));
      }
      if (overlap && a.t1 < T) {
        hipEvent_t ev = e->ev_chunks[t0 / chunk];
        SRV_HIPC(hipEventRecord(ev, e->stream));
        SRV_HIPC(hipStreamWaitEvent(e->vstream, ev, 0));
        value_pass(e->vstream, t0 * N, a.t1 * N, vgrid_max);
      }
    }
  }
  {
    ProfScope ps(e, MOBROB_K_ACT);   // V of the last chunk (V(last_obs) is finish_rollout's)
    const int done_rows = overlap ? ((T - 1) / chunk) * chunk * N : 0;
    value_pass(e->stream, done_rows, T * N, 256);
    if (overlap && done_rows > 0) {
      SRV_HIPC(hipEventRecord(e->ev_vdone, e->vstream));
      SRV_HIPC(hipStreamWaitEvent(e->stream, e->ev_vdone, 0));
    }
  }
  SRV_HIPC(hipGetLastError());
#undef SRV_HIPC

  // ---- the host's side of the step loop: wait for the flags of a row range, step it, raise the range's word ----
  auto now_s = [] { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; };
  const int bpp = nparts == 1 ? rblocks : (N / nparts) / 32;   // workgroups per row range
  {
    // Residency check, before the environment is touched: the clipped actions of step 0 of EVERY workgroup within a short bound
    // (MOBROB_SERVER_RESIDENCY_S, default 2 s; a resident workgroup needs ~25 us).  A workgroup that is not resident -- another
    // process on the device, a CU mask nobody announced -- would leave the resident ones waiting for the host while the host waits
    // for it: instead of stalling for the whole timeout and failing the rollout, the launches are told to stop and the caller runs
    // the launch-per-step collector from the untouched rollout state (nothing has been stepped, no counter has moved).
    const double bound = getenv("MOBROB_SERVER_RESIDENCY_S") ? atof(getenv("MOBROB_SERVER_RESIDENCY_S")) : 2.0;
    const double r0 = now_s();
    bool all_resident = false;
    for (unsigned spins = 0; bound > 0.0; ++spins) {   // (a bound <= 0 always misses: how the tests reach the fallback below)
      int b = 0;
      while (b < rblocks && __atomic_load_n(&gpu_flag[b], __ATOMIC_ACQUIRE) >= 1u) ++b;
      if (b == rblocks) { all_resident = true; break; }
      __builtin_ia32_pause();
      if ((spins & 0x3FFu) == 0x3FFu && (__atomic_load_n(err_word, __ATOMIC_ACQUIRE) != 0 || now_s() - r0 > bound)) break;
    }
    if (!all_resident) {
      give_up();
      if (mode == 2) return fail(MOBROB_ERR_STATE, "collect_host: MOBROB_COLLECT_SERVER=2 but not every workgroup of the rollout kernel became resident within %.1f s (another tenant on the device?)", bound);
      return MOBROB_OK;   // *served is false: the caller's launch-per-step loop takes over
    }
  }
  const bool timing = getenv("MOBROB_SERVER_TIMING") != nullptr;   // per-step split of the host thread's time (stderr)
  double tw = 0, te = 0;
  for (int t = 0; t < T; ++t) {
    for (int p = 0; p < nparts; ++p) {
      const unsigned want = (unsigned)(t + 1);
      double t_wait = -1.0;
      const double c0 = timing ? now_s() : 0;
      for (int b = p * bpp; b < (p + 1) * bpp; ++b) {
        unsigned spins = 0;
        while (__atomic_load_n(&gpu_flag[b], __ATOMIC_ACQUIRE) < want) {
          __builtin_ia32_pause();
          if ((++spins & 0xFFFFu) == 0) {   // every ~65 k polls: the device gave up, or nothing moved for the whole timeout
            if (__atomic_load_n(err_word, __ATOMIC_ACQUIRE) != 0) {
              give_up();
              return fail(MOBROB_ERR_STATE, "collect_host: the rollout kernel gave up waiting for the host at step %d", *err_word - 1);
            }
            const double tn = now_s();
            if (t_wait < 0) t_wait = tn;
            if (tn - t_wait > timeout_s) {
              give_up();
              return fail(MOBROB_ERR_HIP, "collect_host: no actions from the device for %.0f s (step %d, row range %d)", timeout_s, t, p);
            }
          }
        }
      }
      const int r0 = p * (N / nparts);
      const double c1 = timing ? now_s() : 0;
      const int32_t ntrunc = step_range(env, r0, r0 + N / nparts, actions_clipped, obs, rewards, dones, truncated, terminal_obs);
      if (ntrunc < 0) {
        give_up();
        return fail(MOBROB_ERR_STATE, "collect_host: the environment's step_range returned %d", ntrunc);
      }
      __atomic_store_n(reinterpret_cast<unsigned long long*>(&host_flag[16 * p]), ((unsigned long long)(unsigned)ntrunc << 32) | want, __ATOMIC_RELEASE);
      if (timing) { tw += c1 - c0; te += now_s() - c1; }
    }
  }
#ifdef MOBROB_SERVE_STAMPS
  if (timing) {
    (void)hipStreamSynchronize(e->stream);
    const long long* st = reinterpret_cast<const long long*>(err_word + 16);
    double d[8] = {0};
    int n = 0;
    for (int t = 8; t + 1 < 64 && t + 1 < T; ++t, ++n) {
      d[0] += st[8 * t + 1] - st[8 * t + 0];        // publish -> host word seen
      d[1] += st[8 * t + 2] - st[8 * t + 1];        // barrier (4b)
      d[2] += st[8 * t + 3] - st[8 * t + 2];        // pull (this wave)
      d[3] += st[8 * t + 4] - st[8 * t + 3];        // barrier (4c)
      d[4] += st[8 * t + 5] - st[8 * t + 4];        // env phase + state update
      d[5] += st[8 * (t + 1) + 6] - st[8 * t + 5];  // layers, head, sampling (next step)
      d[6] += st[8 * (t + 1) + 7] - st[8 * (t + 1) + 6];  // drain of the action stores
      d[7] += st[8 * (t + 1) + 0] - st[8 * (t + 1) + 7];  // barrier (4)
    }
    fprintf(stderr, "[served stamps, workgroup 0, us] wait for host %.2f | (4b) %.2f | pull %.2f | (4c) %.2f | env+state %.2f | policy %.2f | drain %.2f | (4) %.2f\n",
            d[0] / n / 100, d[1] / n / 100, d[2] / n / 100, d[3] / n / 100, d[4] / n / 100, d[5] / n / 100, d[6] / n / 100, d[7] / n / 100);
  }
#endif
  if (timing)
    fprintf(stderr, "[collect_host served] per step: waiting for the device %.1f us, env %.1f us (%d row ranges)\n", 1e6 * tw / T, 1e6 * te / T, nparts);
  e->t = T;
  e->nparts = nparts;
  for (int p = 0; p < nparts; ++p) { e->part_act_t[p] = e->part_store_t[p] = T; e->part_obs_t[p] = T; }
  if (e->draw_counter < e->draw_ro0 + (uint32_t)T) e->draw_counter = e->draw_ro0 + (uint32_t)T;
  *served = true;
  const int rc = mobrob_ppo_finish_rollout(e, obs, dones);   // last observations / dones, V(last_obs), GAE; synchronises the stream
  if (rc == MOBROB_OK && __atomic_load_n(err_word, __ATOMIC_ACQUIRE) != 0)
    return fail(MOBROB_ERR_STATE, "collect_host: the rollout kernel gave up waiting for the host at step %d", *err_word - 1);
  return rc;
}
}  // namespace

int mobrob_ppo_collect_synthetic(mobrob_ppo_engine_t* e, float p_term, int32_t time_limit) {
  if (!e) return fail(MOBROB_ERR_INVALID, "null engine");
  mobrob_ppo_engine::RolloutSpec sp;
  memset(&sp, 0, sizeof sp);  // padding bytes too: specs are compared with memcmp
  sp.kind = 1; sp.p_term = p_term; sp.time_limit = time_limit;
  return collect_device(e, sp);
}

int mobrob_ppo_collect_goal_env(mobrob_ppo_engine_t* e, const mobrob_goal_env_t* env) {
  if (!e || !env) return fail(MOBROB_ERR_INVALID, "null argument");
  if (env->pos_dim < 1 || env->pos_dim > 3 || 3 * env->pos_dim > e->D)
    return fail(MOBROB_ERR_INVALID, "goal env: pos_dim must be 1..3 and 3*pos_dim <= obs_dim");
  if (e->A > 32) return fail(MOBROB_ERR_INVALID, "goal env: act_dim must be <= 32");
  if (env->time_limit < 1) return fail(MOBROB_ERR_INVALID, "goal env: time_limit must be >= 1");
  mobrob_ppo_engine::RolloutSpec sp;
  memset(&sp, 0, sizeof sp);
  sp.kind = 2; sp.time_limit = env->time_limit;
  GoalEnvParams& g = sp.goal;
  g.P = env->pos_dim; g.terminate_on_goal = env->terminate_on_goal != 0; g.time_limit = env->time_limit;
  g.dt = env->dt; g.extent = env->extent; g.reach = env->reach_radius; g.bonus = env->goal_bonus;
  g.extra_bonus = env->extra_bonus; g.noise = env->obs_noise;
  for (int j = 0; j < 3; ++j)
    for (int k = 0; k < 32; ++k) g.mix[j][k] = (j < env->pos_dim && k < e->A) ? env->mix[j][k] : 0.f;
  return collect_device(e, sp);
}

int mobrob_ppo_episode_stats(mobrob_ppo_engine_t* e, mobrob_episode_stats_t* out, int32_t reset) {
  if (!e || !out) return fail(MOBROB_ERR_INVALID, "null argument");
  double h[4];
  HIPC(hipMemcpyAsync(h, e->ep_stats, sizeof h, hipMemcpyDeviceToHost, e->stream));
  if (reset) HIPC(hipMemsetAsync(e->ep_stats, 0, sizeof h, e->stream));
  HIPC(hipStreamSynchronize(e->stream));
  out->episodes = (int64_t)h[0]; out->return_sum = h[1]; out->length_sum = h[2]; out->goals = (int64_t)h[3];
  return MOBROB_OK;
}

int mobrob_ppo_episode_records(mobrob_ppo_engine_t* e, float* out, int32_t max_records) {
  if (!e || !out || max_records < 0) return fail(MOBROB_ERR_INVALID, "episode_records: bad argument");
  std::vector<double> h(kEpStatsDoubles);
  HIPC(hipMemcpyAsync(h.data(), e->ep_stats, h.size() * sizeof(double), hipMemcpyDeviceToHost, e->stream));
  HIPC(hipStreamSynchronize(e->stream));
  const uint64_t written = (uint64_t)h[4];
  uint64_t first = e->ep_ring_read;
  if (written - first > (uint64_t)kEpRing) first = written - kEpRing;           // older ones were overwritten
  if (written - first > (uint64_t)max_records) first = written - max_records;   // the caller wants the newest
  int n = 0;
  for (uint64_t k = first; k < written; ++k, ++n) {
    memcpy(&out[2 * n], &h[5 + k % kEpRing], 2 * sizeof(float));  // one 64-bit record = (return, length) as two floats
  }
  e->ep_ring_read = written;
  return n;
}

// ---- env-side controllers (robot_ctrl.h) -----------------------------------------------------------------
}  // extern "C"
namespace {
// host arrays are staged through temporary device buffers; device arrays (dev_ptrs != 0) are used in place
struct CtrlBuf {
  mobrob_ppo_engine* e; bool dev; std::vector<void*> owned;
  std::vector<std::pair<void*, std::pair<void*, size_t>>> outs;  // device -> host copies to make at the end
  ~CtrlBuf() { for (void* p : owned) (void)hipFree(p); }
  int in(const float* host, size_t count, const float** out) {
    using T = float;
    if (dev) { *out = host; return MOBROB_OK; }
    void* d = nullptr;
    HIPC(hipMalloc(&d, count * sizeof(T)));
    owned.push_back(d);
    HIPC(hipMemcpyAsync(d, host, count * sizeof(T), hipMemcpyHostToDevice, e->stream));
    *out = static_cast<const T*>(d);
    return MOBROB_OK;
  }
  int inout(float* host, size_t count, float** out, bool copy_in) {
    using T = float;
    if (dev) { *out = host; return MOBROB_OK; }
    void* d = nullptr;
    HIPC(hipMalloc(&d, count * sizeof(T)));
    owned.push_back(d);
    if (copy_in) HIPC(hipMemcpyAsync(d, host, count * sizeof(T), hipMemcpyHostToDevice, e->stream));
    outs.push_back({d, {host, count * sizeof(T)}});
    *out = static_cast<T*>(d);
    return MOBROB_OK;
  }
  int finish() {
    for (auto& o : outs) HIPC(hipMemcpyAsync(o.second.first, o.first, o.second.second, hipMemcpyDeviceToHost, e->stream));
    if (!dev) HIPC(hipStreamSynchronize(e->stream));
    return MOBROB_OK;
  }
};
}  // namespace
extern "C" {

int mobrob_ctrl_turtlebot3(mobrob_ppo_engine_t* e, int32_t n, int32_t dev_ptrs, const float* pos, const float* theta,
                           const float* goal, const float* gain_changes, float* twist) {
  if (!e || n < 1 || !pos || !theta || !goal || !gain_changes || !twist)
    return fail(MOBROB_ERR_INVALID, "ctrl_turtlebot3: bad argument");
  CtrlBuf b{e, dev_ptrs != 0, {}, {}};
  const float *dp, *dt, *dg, *da;
  float* dtw;
  CHK(b.in(pos, (size_t)2 * n, &dp)); CHK(b.in(theta, (size_t)n, &dt)); CHK(b.in(goal, (size_t)2 * n, &dg));
  CHK(b.in(gain_changes, (size_t)2 * n, &da)); CHK(b.inout(twist, (size_t)2 * n, &dtw, false));
  hipLaunchKernelGGL(k_ctrl_turtlebot3, dim3(cdiv(n, 256)), dim3(256), 0, e->stream, n, dp, dt, dg, da, dtw);
  HIPC(hipGetLastError());
  return b.finish();
}

int mobrob_ctrl_drone_pid(mobrob_ppo_engine_t* e, int32_t n, int32_t dev_ptrs, const mobrob_drone_params_t* prm,
                          const float* pos, const float* rpy, const float* goal, const float* action, float* ctrl_state,
                          float* out) {
  if (!e || n < 1 || !prm || !pos || !rpy || !goal || !action || !ctrl_state || !out)
    return fail(MOBROB_ERR_INVALID, "ctrl_drone_pid: bad argument");
  if (!(prm->dt > 0.f) || !(prm->mass > 0.f)) return fail(MOBROB_ERR_INVALID, "ctrl_drone_pid: dt and mass must be positive");
  CtrlBuf b{e, dev_ptrs != 0, {}, {}};
  const float *dp, *dr, *dg, *da;
  float *ds, *dout;
  CHK(b.in(pos, (size_t)3 * n, &dp)); CHK(b.in(rpy, (size_t)3 * n, &dr)); CHK(b.in(goal, (size_t)3 * n, &dg));
  CHK(b.in(action, (size_t)18 * n, &da)); CHK(b.inout(ctrl_state, (size_t)12 * n, &ds, true));
  CHK(b.inout(out, (size_t)4 * n, &dout, false));
  DroneCtrlParams p{prm->mass, prm->g, prm->dt, prm->max_thrust, prm->max_xy_torque, prm->max_z_torque, prm->max_roll_pitch,
                    prm->tune_fac};
  hipLaunchKernelGGL(k_ctrl_drone_pid, dim3(cdiv(n, 256)), dim3(256), 0, e->stream, n, p, dp, dr, dg, da, ds, dout);
  HIPC(hipGetLastError());
  return b.finish();
}

// ---- update ----------------------------------------------------------------------------------------
int mobrob_ppo_num_minibatches(const mobrob_ppo_engine_t* e) { return e ? e->nmb : -1; }

// The training records (kernels_fused.h) are packed once per rollout: every call that can change actions / values /
// log-probs / advantages / returns clears train_rec_valid (rollouts, GAE, write_buffer); a caller that obtained a device
// pointer to one of them (buffer_info) can write without the engine seeing it, so from then on they are re-packed before
// every gradient launch.
static void ensure_train_records(mobrob_ppo_engine* e) {
  if (!e->fused.train_rec || (e->train_rec_valid && !e->train_rec_external)) return;
  TrainRecArgs r{};
  r.actions = e->actions; r.old_logp = e->logp; r.adv = e->adv; r.ret = e->ret; r.values = e->values;
  r.A = e->A; r.RW = train_rec_width(e->A); r.rows = e->N * e->T; r.rec = e->fused.train_rec;
  const long long items = (long long)r.rows * (r.RW / 4);
  hipLaunchKernelGGL(k_build_train_records, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, e->stream, r);
  e->train_rec_valid = true;
}

int mobrob_ppo_epoch_begin(mobrob_ppo_engine_t* e, const int64_t* perm) {
  if (!e) return fail(MOBROB_ERR_INVALID, "null engine");
  if (!e->rollout_ready) return fail(MOBROB_ERR_STATE, "epoch_begin: rollout not finished (finish_rollout / collect first)");
  const int total = e->N * e->T;
  if (perm) {
    // a caller's permutation is checked before a kernel scatters through it (k_perm_from_host writes rows[] and the
    // minibatch-of-row table at the indices it holds): every flat index once, none out of range.  FIRST, before any engine
    // state moves (the parity of the max-|adv| words below, the training records): a rejected call leaves nothing behind.
    std::vector<uint64_t> seen(((size_t)total + 63) / 64, 0);
    for (int i = 0; i < total; ++i) {
      const int64_t v = perm[i];
      if (v < 0 || v >= total) return fail(MOBROB_ERR_INVALID, "epoch_begin: perm[%d] = %lld is outside [0, %d)", i, (long long)v, total);
      uint64_t& w = seen[(size_t)v >> 6];
      const uint64_t bit = 1ull << (v & 63);
      if (w & bit) return fail(MOBROB_ERR_INVALID, "epoch_begin: perm holds index %lld twice (not a permutation of [0, %d))", (long long)v, total);
      w |= bit;
    }
  }
  ensure_train_records(e);
  AdvStatArgs as{};
  as.adv = e->adv; as.total = total; as.T = e->T; as.N = e->N; as.bl = e->Bl; as.nmb = e->nmb;
  as.bins = e->advbins;
  unsigned* maxw = reinterpret_cast<unsigned*>(e->advbins + (size_t)2 * e->nmb);  // two words, used alternately
  as.absmax_bits = maxw + (e->adv_pass & 1); as.absmax_next = maxw + ((e->adv_pass + 1) & 1);
  e->adv_pass++;
  if (perm) {
    HIPC(hipMemcpyAsync(e->perm_dev, perm, (size_t)total * 8, hipMemcpyHostToDevice, e->stream));
    hipLaunchKernelGGL(k_perm_from_host, dim3(cdiv(total, 256)), dim3(256), 0, e->stream, e->perm_dev, total, e->T,
                       e->N, e->Bl, e->rows);
    as.mb_of_row = reinterpret_cast<const int*>(e->perm_dev);
  } else {
    const uint64_t key = (e->cfg.seed * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)(e->cfg.rank + 1) << 48) ^ (e->perm_counter + 1);
    e->perm_counter++;
    as.half_bits = feistel_half_bits((uint64_t)total); as.k0 = (uint32_t)key; as.k1 = (uint32_t)(key >> 32);
    hipLaunchKernelGGL(k_perm_feistel, dim3(cdiv(total, 256)), dim3(256), 0, e->stream, total, e->T, e->N,
                       as.half_bits, as.k0, as.k1, e->rows, (int64_t*)nullptr);
  }
  // advantage statistics: one coalesced pass in storage order, order-independent integer sums (kernels_generic.h)
  const int sgrid = std::max(1, std::min(256, cdiv(total, 4 * 1024)));  // at most one workgroup per CU (see k_adv_absmax)
  const size_t slds = e->nmb <= kAdvLdsMinibatches ? (size_t)2 * e->nmb * adv_bin_replicas(e->nmb) * sizeof(unsigned long long) : 0;
  hipLaunchKernelGGL(k_adv_absmax, dim3(sgrid), dim3(1024), 0, e->stream, as);
  hipLaunchKernelGGL(k_adv_stats_stream, dim3(sgrid), dim3(1024), slds, e->stream, as);
  hipLaunchKernelGGL(k_adv_fold, dim3(cdiv(e->nmb, 64)), dim3(64), 0, e->stream, as, e->advstat);
  HIPC(hipGetLastError());
  e->epoch_open = true;
  return MOBROB_OK;
}

int mobrob_ppo_minibatch_grad(mobrob_ppo_engine_t* e, int32_t mb) {
  if (!e) return fail(MOBROB_ERR_INVALID, "null engine");
  if (!e->epoch_open) return fail(MOBROB_ERR_STATE, "minibatch_grad before epoch_begin");
  if (mb < 0 || mb >= e->nmb) return fail(MOBROB_ERR_INVALID, "minibatch %d out of range [0,%d)", mb, e->nmb);
  const int total = e->N * e->T;
  const int start = mb * e->Bl;
  const int B = std::min(e->Bl, total - start);
  const float inv_bg = 1.0f / (float)((int64_t)B * e->cfg.world_size);
  e->cur_count = B;
  if (e->fused.enabled) {
    if (e->fused.H == 64) fused64_minibatch_grad(e, mb, start, B, inv_bg);
    else {
      ensure_train_records(e);
      fused_minibatch_grad(e, mb, start, B, inv_bg);
    }
    HIPC(hipGetLastError());
    e->grad_pending = true;
    return MOBROB_OK;
  }
  ProfScope ps(e, MOBROB_K_TRAIN_GRAD);
  float* sums = e->grads + e->P;
  const int per = e->Dp / 4;
  // (the gather also zeroes the gradient vector + loss sums and the gSDE GEMM's output: everything this step adds into with atomics)
  hipLaunchKernelGGL(k_gather, dim3(cdiv(B * per, 256)), dim3(256), 0, e->stream, e->rows + start, B, e->obs, e->Dp,
                     e->actions, e->A, e->logp, e->adv, e->ret, e->Xg, e->actg, e->lpg, e->advg, e->retg, e->values,
                     e->clip_vf >= 0.0 ? e->oldvg : (float*)nullptr, e->grads, e->P + 8, e->sde ? e->sde_graw : (float*)nullptr,
                     e->sde ? e->HL * e->A : 0);
  forward_generic(e, e->Xg, B, true, e->mu, true, e->vout, true);
  LossArgs L{};
  L.mu = e->mu; L.ldmu = e->Ap; L.v = e->vout; L.actions = e->actg; L.old_logp = e->lpg; L.adv = e->advg;
  L.ret = e->retg; L.log_std = Pp(e, T_LOGSTD); L.advstat = e->advstat + 4 * (size_t)mb; L.B = B; L.A = e->A;
  L.normalize = e->cfg.normalize_advantage; L.clip = (float)e->cfg.clip_range; L.vf_coef = (float)e->cfg.vf_coef;
  L.clip_vf = (float)e->clip_vf; L.old_v = e->oldvg;
  L.ent_coef = (float)e->cfg.ent_coef; L.inv_bg = inv_bg; L.dmu = e->dmu; L.lddmu = e->Ap; L.dv = e->dv; L.lddv = 8;
  L.sums = sums; L.g_log_std = Gp(e, T_LOGSTD); L.g_b_action = Gp(e, T_AB); L.g_b_value = Gp(e, T_VB);
  if (e->sde) {
    sde_variance(e, B, true, e->sde_lat2);   // (log_std moved in the previous optimizer step; latent^2 for the gradient GEMM on the way)
    L.lat = e->hp[e->Lp - 1]; L.HL = e->HL; L.var = e->sde_var; L.ldvar = e->Ap; L.gsig = e->sde_gsig; L.ldg = e->Ap;
  }
  // (the padding columns of dmu / dv -- K padding of the NN GEMMs -- are zeroed by k_loss itself)
  hipLaunchKernelGGL(k_loss, dim3(cdiv(B, 256)), dim3(256), loss_lds_bytes(e->A), e->stream, L);
  if (e->sde) {   // g_log_std = 2 std (d std / d log_std) * ((latent^2)^T . gsig) (summed over the actions without full_std): the entropy term is inside gsig
    linear_bwd_weight(e, e->sde_lat2, e->HL, e->sde_gsig, e->Ap, e->sde_graw, e->A, e->HL, e->A, B);
    hipLaunchKernelGGL(k_sde_scale_grad, dim3(cdiv(e->HL * e->A, 256)), dim3(256), 0, e->stream, Gp(e, T_LOGSTD), e->sde_graw, Pp(e, T_LOGSTD), e->HL,
                       e->A, e->sde_mode);
  }   // (state-independent log_std: its entropy term is added by k_loss)
  // backward of both networks, last hidden layer first: dW of the layer above, then dz of this layer (dtanh / dReLU + bias sums)
  GemmQueue qp, qv;
  auto backward = [&](GemmQueue* q, int L, const int* Hw, float* const* h, float* const* dz, const float* dhead, int ldd, const float* headWp,
                      int head_rows, int head_pad, int t_headW, const int* tW, const int* tB) {
    const int HLw = Hw[L - 1];
    linear_bwd_weight(e, dhead, ldd, h[L - 1], HLw, Gp(e, t_headW), HLw, head_rows, HLw, B, q);
    linear_bwd_input(e, dhead, ldd, headWp, HLw, h[L - 1], HLw, dz[L - 1], HLw, Gp(e, tB[L - 1]), B, HLw, head_pad, q);
    for (int l = L - 1; l >= 1; --l) {
      linear_bwd_weight(e, dz[l], Hw[l], h[l - 1], Hw[l - 1], Gp(e, tW[l]), Hw[l - 1], Hw[l], Hw[l - 1], B, q);
      linear_bwd_input(e, dz[l], Hw[l], Pp(e, tW[l]), Hw[l - 1], h[l - 1], Hw[l - 1], dz[l - 1], Hw[l - 1], Gp(e, tB[l - 1]), B, Hw[l - 1], Hw[l], q);
    }
    linear_bwd_weight(e, dz[0], Hw[0], e->Xg, e->Dp, Gp(e, tW[0]), e->D, Hw[0], e->D, B, q);
  };
  // both networks' chains alternate weight-gradient and input-gradient GEMMs from the head down: queued, then launched pairwise
  backward(&qp, e->Lp, e->Hp, e->hp, e->dzp, e->dmu, e->Ap, e->aWp, e->A, e->Ap, T_AW, e->tPW, e->tPB);
  backward(&qv, e->Lv, e->Hv, e->hv, e->dzv, e->dv, 8, e->vWp, 1, 8, T_VW, e->tVW, e->tVB);
  run_queues(e, qp, qv, 2);
  HIPC(hipGetLastError());
  e->grad_pending = true;
  return MOBROB_OK;
}

namespace {
// everything of AdamPackArgs that does not change from step to step
void fill_adam_pack_args(mobrob_ppo_engine* e, AdamPackArgs& a) {
  a.p = e->params; a.g = e->grads; a.m = e->m; a.v = e->v; a.P = e->P;
  a.chunks = e->chunks_dev; a.partial = e->chunk_partial; a.nchunks = e->nchunks;
  a.max_norm = (float)e->cfg.max_grad_norm; a.beta1 = (float)e->cfg.adam_beta1; a.beta2 = (float)e->cfg.adam_beta2;
  a.eps = (float)e->cfg.adam_eps;
  for (int i = 0; i <= kMaxTensors; ++i) a.offs[i] = e->offs[i];
  a.ntens = e->ntens; a.two_by_two = e->Lp == 2 && e->Lv == 2; a.id_pw1 = T_PW1; a.id_vw1 = T_VW1; a.id_aw = T_AW; a.id_vw = T_VW; a.HL = e->HL; a.GL = e->GL;
  a.D = e->D; a.Dp = e->Dp; a.A = e->A; a.Ap = e->Ap; a.H1 = e->H1; a.H2 = e->H2; a.G1 = e->G1; a.G2 = e->G2;
  a.pW1p = e->pW1p; a.vW1p = e->vW1p; a.aWp = e->aWp; a.vWp = e->vWp;
  for (int n = 0; n < 2; ++n) {
    const bool on = e->fused.enabled;
    a.fW1f[n] = on ? (float*)e->fused.net[n].W1f : nullptr;
    a.fW2f[n] = on ? (float*)e->fused.net[n].W2f : nullptr;
    a.fW3f[n] = on ? (float*)e->fused.net[n].W3f : nullptr;
    // k_chain_train reads its own packs only: the backward packs of the other 256-wide gradient kernels (W2b, W3b, W2bx) are not
    // refreshed per step while it is the engine's gradient kernel (set_params / load rebuild every pack: fused_repack, pack_x3_all).
    // W3h stays: k_value_batch's 16-wide head reads it.
    const bool ch = on && e->fused.train_chain;
    a.fW2b[n] = on && !ch ? (float*)e->fused.net[n].W2b : nullptr;
    a.fW3b[n] = on && !ch ? (float*)e->fused.net[n].W3b : nullptr;
    a.fW3h[n] = (on && (n == 1 || e->fused.A <= 16)) ? (float*)e->fused.net[n].W3h : nullptr;
    a.fb1s[n] = on ? (float*)e->fused.net[n].b1s : nullptr;
    a.fb2s[n] = on ? (float*)e->fused.net[n].b2s : nullptr;
    const bool x3 = on && e->fused.train_x3;  // the gradient kernel reads the x3 packs every step: k_adam_pack keeps them current
    a.xW1[n] = x3 ? reinterpret_cast<unsigned short*>(const_cast<unsigned*>(e->fused.net[n].W1x)) : nullptr;
    a.xW2[n] = x3 ? reinterpret_cast<unsigned short*>(const_cast<unsigned*>(e->fused.net[n].W2x)) : nullptr;
    a.xW2b[n] = x3 && !ch ? reinterpret_cast<unsigned short*>(const_cast<unsigned*>(e->fused.net[n].W2bx)) : nullptr;
    // likewise the chain packs of k_chain_train
    a.cW1[n] = ch ? reinterpret_cast<unsigned short*>(const_cast<unsigned*>(e->fused.net[n].W1c)) : nullptr;
    a.cW2[n] = ch ? reinterpret_cast<unsigned short*>(const_cast<unsigned*>(e->fused.net[n].W2c)) : nullptr;
    a.cW2b[n] = ch ? reinterpret_cast<unsigned short*>(const_cast<unsigned*>(e->fused.net[n].W2bc)) : nullptr;
    a.cW3[n] = ch ? const_cast<float*>(e->fused.net[n].W3c) : nullptr;
    a.cW3b[n] = ch ? const_cast<float*>(e->fused.net[n].W3bc) : nullptr;
  }
}
}  // namespace

namespace {
struct ApplyCtx { float* stats_row; bool records; };
// first half of an optimizer step: loss statistics + per-tensor sums of squares of the (reduced) gradient
int apply_norms(mobrob_ppo_engine* e, ApplyCtx& c) {
  if (e->stats_n >= e->stats_cap) e->stats_n = 0;  // ring: oldest rows are dropped if nobody fetched them
  c.stats_row = e->stats + (size_t)e->stats_n * 8;
  e->stats_n++;
  StatsArgs st{};
  st.stats_row = c.stats_row; st.loss_sums = e->grads + e->P; st.log_std = Pp(e, T_LOGSTD);
  st.ent_coef = (float)e->cfg.ent_coef; st.vf_coef = (float)e->cfg.vf_coef;
  st.inv_bg = 1.0f / (float)((int64_t)e->cur_count * e->cfg.world_size); st.n_act = e->A; st.sde = e->sde;
  c.records = e->use_norm_records && e->fused.enabled;  // the reduction kernel of this step left the norm records
  if (!c.records)
    hipLaunchKernelGGL(k_sqnorm_chunks, dim3(e->nchunks), dim3(256), 0, e->stream, e->grads, e->chunks_dev,
                       e->chunk_partial, st);
  return MOBROB_OK;
}
// second half: clip coefficient, Adam, re-pack
int apply_adam(mobrob_ppo_engine* e, const ApplyCtx& c) {
  e->adam_step++;
  const double b1 = e->cfg.adam_beta1, b2 = e->cfg.adam_beta2;
  const double bc1 = 1.0 - std::pow(b1, (double)e->adam_step);
  const double bc2 = 1.0 - std::pow(b2, (double)e->adam_step);
  AdamPackArgs a{};
  fill_adam_pack_args(e, a);
  a.step_size = (float)(e->cfg.learning_rate / bc1);
  a.bc2_sqrt = (float)std::sqrt(bc2);
  if (c.records) {
    StatsArgs st{};
    st.stats_row = c.stats_row; st.loss_sums = e->grads + e->P; st.log_std = Pp(e, T_LOGSTD);
    st.ent_coef = (float)e->cfg.ent_coef; st.vf_coef = (float)e->cfg.vf_coef;
    st.inv_bg = 1.0f / (float)((int64_t)e->cur_count * e->cfg.world_size); st.n_act = e->A;
    a.partial = e->norm_rec_sum; a.fold_idx = e->fold_idx_dev; a.st = st;
    for (int i = 0; i <= kMaxTensors; ++i) a.fold_start[i] = e->fold_start[i];
  }
  a.stats_row = c.stats_row;
  a.loss_sums_zero = e->fused.enabled ? e->grads + e->P : nullptr;
  hipLaunchKernelGGL(k_adam_pack, dim3(cdiv(e->P, 256)), dim3(256), 0, e->stream, a);
  HIPC(hipGetLastError());
  e->grad_pending = false;
  return MOBROB_OK;
}
}  // namespace

namespace {
// clip + Adam of the pending gradient, with SB3's target_kl check in front of it when that option is on: the
// minibatch's approx_kl is known after the loss statistics; above 1.5 x target the optimizer step of THIS minibatch is
// dropped (*stopped = 1; the caller drops the rest of train()).  One 4-byte read-back per step.  Under data parallel
// the value is that of the union minibatch (its sum travelled with the gradient), bit-equal on every rank.
int apply_checked(mobrob_ppo_engine* e, int32_t* stopped) {
  *stopped = 0;
  ProfScope ps(e, MOBROB_K_APPLY);
  ApplyCtx c{};
  CHK(apply_norms(e, c));
  if (e->target_kl > 0.0) {
    float approx_kl = 0.f;
    HIPC(hipMemcpyAsync(&approx_kl, c.stats_row + 4, sizeof(float), hipMemcpyDeviceToHost, e->stream));
    HIPC(hipStreamSynchronize(e->stream));
    if ((double)approx_kl > 1.5 * e->target_kl) {
      *stopped = 1;
      e->grad_pending = false;
      // the row of the dropped step carries its losses (SB3 appends them before the check) but no gradient norm:
      // all-ones bits = NaN, skipped by the averages
      HIPC(hipMemsetAsync(c.stats_row + 6, 0xFF, sizeof(float), e->stream));
      if (e->fused.enabled) HIPC(hipMemsetAsync(e->grads + e->P, 0, 8 * sizeof(float), e->stream));  // what k_adam_pack would have re-zeroed
      return MOBROB_OK;
    }
  }
  return apply_adam(e, c);
}
}  // namespace

int mobrob_ppo_minibatch_apply(mobrob_ppo_engine_t* e) {
  if (!e) return fail(MOBROB_ERR_INVALID, "null engine");
  if (!e->grad_pending) return fail(MOBROB_ERR_STATE, "minibatch_apply without a pending gradient");
  if (e->target_kl > 0.0)
    return fail(MOBROB_ERR_STATE, "target_kl is set: a step-wise driver must call mobrob_ppo_minibatch_apply_checked and honour its stop flag");
  int32_t stopped = 0;
  return apply_checked(e, &stopped);
}

int mobrob_ppo_minibatch_apply_checked(mobrob_ppo_engine_t* e, int32_t* stopped) {
  if (!e || !stopped) return fail(MOBROB_ERR_INVALID, "minibatch_apply_checked: null argument");
  if (!e->grad_pending) return fail(MOBROB_ERR_STATE, "minibatch_apply without a pending gradient");
  return apply_checked(e, stopped);
}

int mobrob_ppo_sde_reset_noise(mobrob_ppo_engine_t* e) {
  if (!e) return fail(MOBROB_ERR_INVALID, "null engine");
  if (!e->sde) return fail(MOBROB_ERR_STATE, "sde_reset_noise: the engine was not created with use_sde");
  sde_resample(e, 0, e->N, e->draw_counter++, nullptr, true);
  HIPC(hipGetLastError());
  return MOBROB_OK;
}

int mobrob_ppo_sde_set_noise(mobrob_ppo_engine_t* e, const float* z) {
  if (!e) return fail(MOBROB_ERR_INVALID, "null engine");
  if (!e->sde) return fail(MOBROB_ERR_STATE, "sde_set_noise: the engine was not created with use_sde");
  if (e->sde_hold != (z != nullptr) && e->ro_exec) {
    // a captured device rollout has the resampling launches (or their absence) baked in: capture again under the new mode
    HIPC(hipStreamSynchronize(e->stream));
    (void)hipGraphExecDestroy(e->ro_exec); e->ro_exec = nullptr;
    if (e->ro_graph) { (void)hipGraphDestroy(e->ro_graph); e->ro_graph = nullptr; }
  }
  e->sde_hold = z != nullptr;
  if (!z) return MOBROB_OK;
  const size_t HLA = (size_t)e->HL * e->A, n = (size_t)e->N * HLA;
  if (n > (size_t)1 << 30) return fail(MOBROB_ERR_INVALID, "sde_set_noise: too many matrix elements");
  // staged through the matrices' own storage: z is uploaded into it and scaled in place
  HIPC(hipMemcpyAsync(e->sde_E, z, n * 4, hipMemcpyHostToDevice, e->stream));
  hipLaunchKernelGGL(k_sde_from_z, dim3(cdiv((int)n, 256)), dim3(256), 0, e->stream, e->sde_E, Pp(e, T_LOGSTD), (int)HLA, e->A, e->sde_mode, (int)n, e->sde_E);
  hipLaunchKernelGGL(k_copy_f32, dim3(cdiv((int)HLA, 256)), dim3(256), 0, e->stream, e->sde_E, e->sde_E1, (int)HLA);   // exploration_mat := env 0's
  HIPC(hipGetLastError());
  HIPC(hipStreamSynchronize(e->stream));   // z may be pageable host memory
  return MOBROB_OK;
}

int mobrob_ppo_set_hyper(mobrob_ppo_engine_t* e, int32_t which, double value) {
  if (!e) return fail(MOBROB_ERR_INVALID, "null engine");
  switch (which) {
    case MOBROB_HYPER_LEARNING_RATE:
      if (!(value >= 0.0)) return fail(MOBROB_ERR_INVALID, "learning_rate must be >= 0");
      e->cfg.learning_rate = value; break;
    case MOBROB_HYPER_CLIP_RANGE:
      if (!(value >= 0.0)) return fail(MOBROB_ERR_INVALID, "clip_range must be >= 0");
      e->cfg.clip_range = value; break;
    case MOBROB_HYPER_CLIP_RANGE_VF: e->clip_vf = value >= 0.0 ? value : -1.0; break;  // negative / NaN: None
    case MOBROB_HYPER_TARGET_KL: e->target_kl = value > 0.0 ? value : -1.0; break;
    case MOBROB_HYPER_ENT_COEF: e->cfg.ent_coef = value; break;
    case MOBROB_HYPER_VF_COEF: e->cfg.vf_coef = value; break;
    case MOBROB_HYPER_EPOCH_KERNEL: e->epoch_kernel_on = value != 0.0; break;
    default: return fail(MOBROB_ERR_INVALID, "unknown hyper-parameter id %d", which);
  }
  return MOBROB_OK;
}

int mobrob_ppo_last_train_info(const mobrob_ppo_engine_t* e, int32_t* epochs_started, int32_t* stopped_early,
                               int32_t* steps_applied) {
  if (!e) return fail(MOBROB_ERR_INVALID, "null engine");
  if (epochs_started) *epochs_started = e->last_epochs_started;
  if (stopped_early) *stopped_early = e->last_stopped_early;
  if (steps_applied) *steps_applied = e->last_steps_applied;
  return MOBROB_OK;
}

int mobrob_ppo_fetch_step_stats(mobrob_ppo_engine_t* e, float* out, int32_t max_rows) {
  if (!e || !out || max_rows < 0) return fail(MOBROB_ERR_INVALID, "fetch_step_stats: bad argument");
  const int n = std::min<int>(max_rows, e->stats_n);
  if (n > 0) {
    HIPC(hipMemcpyAsync(out, e->stats + (size_t)(e->stats_n - n) * 8, (size_t)n * 32, hipMemcpyDeviceToHost, e->stream));
  }
  HIPC(hipStreamSynchronize(e->stream));
  e->stats_n = 0;
  CHK(check_async_error(e));
  return n;
}

}  // extern "C"
// ---- data parallel: the whole update loop in C, one RCCL all-reduce per optimizer step on the engine's stream ----
namespace {
struct RcclApi {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;
  decltype(&ncclCommUserRank) CommUserRank = nullptr;
};
RcclApi g_rccl;
int rccl_load() {
  if (g_rccl.lib) return MOBROB_OK;
  // by SONAME: a process that already holds RCCL (torch.distributed's "nccl" backend) shares that copy
  void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) return fail(MOBROB_ERR_STATE, "RCCL not found: %s", dlerror());
#define RSYM(field, name)                                                     \
  g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name));   \
  if (!g_rccl.field) return fail(MOBROB_ERR_STATE, "RCCL symbol %s missing", name);
  RSYM(GetUniqueId, "ncclGetUniqueId")
  RSYM(CommInitRank, "ncclCommInitRank")
  RSYM(CommDestroy, "ncclCommDestroy")
  RSYM(AllReduce, "ncclAllReduce")
  RSYM(GetErrorString, "ncclGetErrorString")
  RSYM(CommCount, "ncclCommCount")
  RSYM(CommUserRank, "ncclCommUserRank")
#undef RSYM
  g_rccl.lib = h;
  return MOBROB_OK;
}
#define NCCLC(expr)                                                                                   \
  do {                                                                                                \
    ncclResult_t _r = (expr);                                                                         \
    if (_r != ncclSuccess) return fail(MOBROB_ERR_HIP, "%s failed: %s", #expr, g_rccl.GetErrorString(_r)); \
  } while (0)

size_t oneshot_payload_bytes(const mobrob_ppo_engine* e) {
  const size_t need = std::max((size_t)(e->P + 8) * sizeof(float), (size_t)e->nmb * 4 * sizeof(double));
  return (need + 255) / 256 * 256;
}
int oneshot_all_reduce(mobrob_ppo_engine* e, void* buf, size_t count, int dtype) {
  auto& o = e->oneshot;
  const size_t bytes = count * (dtype == 1 ? sizeof(double) : sizeof(float));
  if (bytes > o.payload) return fail(MOBROB_ERR_INVALID, "one-shot all-reduce: message of %zu bytes, exchange slot of %zu", bytes, o.payload);
  o.seq++;
  OneShotArgs a{};
  const size_t slot = (size_t)(o.seq & 1) * o.payload;
  a.local = buf; a.mine = o.xbuf + slot; a.my_flags = reinterpret_cast<unsigned long long*>(o.xbuf + 2 * o.payload);
  for (int r = 0; r < o.world; ++r) {
    a.peer[r] = o.peer[r] + slot;
    a.peer_flags[r] = reinterpret_cast<const unsigned long long*>(o.peer[r] + 2 * o.payload);
  }
  a.world = o.world; a.rank = o.rank; a.bytes = bytes; a.seq = o.seq; a.error = o.error; a.timeout_ticks = o.timeout_ticks;
  a.fence_form = getenv("MOBROB_ONESHOT_FENCE") != nullptr && atoi(getenv("MOBROB_ONESHOT_FENCE")) != 0;
  const int chunks = cdiv((int)bytes, kOneShotChunkBytes);
  if (dtype == 1) hipLaunchKernelGGL(k_oneshot_allreduce<double>, dim3(chunks), dim3(256), 0, e->stream, a);
  else hipLaunchKernelGGL(k_oneshot_allreduce<float>, dim3(chunks), dim3(256), 0, e->stream, a);
  HIPC(hipGetLastError());
  return MOBROB_OK;
}
// raised by a one-shot all-reduce whose peer never published (dead rank): reported at the next synchronising call
int check_async_error(mobrob_ppo_engine* e) {
  if (e->epoch_err_host && *(volatile int*)e->epoch_err_host != 0) {
    *(volatile int*)e->epoch_err_host = 0;
    return fail(MOBROB_ERR_STATE, "the co-operative epoch kernel gave up at a grid barrier (a workgroup never arrived: another tenant on the "
                                  "device?); the update of this train() is incomplete -- MOBROB_EPOCH_KERNEL=0 runs the three launches per step");
  }
  if (e->oneshot.error && *(volatile int*)e->oneshot.error != 0) {
    const int v = *(volatile int*)e->oneshot.error;
    *(volatile int*)e->oneshot.error = 0;
    return fail(MOBROB_ERR_STATE, "one-shot all-reduce: a peer rank never published message %d (dead or stalled rank)", v - 1);
  }
  return MOBROB_OK;
}

// sum `count` elements (dtype 0 = f32, 1 = f64) in place across the ranks, ordered on the engine's stream
int dp_all_reduce(mobrob_ppo_engine* e, void* buf, size_t count, int dtype, mobrob_allreduce_fn fn, void* ctx) {
  ProfScope ps(e, MOBROB_K_ALLREDUCE);
  e->allreduce_calls++;
  e->allreduce_bytes += count * (dtype == 1 ? 8 : 4);
  if (fn) {
    const int r = fn(ctx, buf, count, dtype, (void*)e->stream);
    return r == 0 ? MOBROB_OK : fail(MOBROB_ERR_STATE, "all-reduce callback returned %d", r);
  }
  if (e->oneshot.ready) return oneshot_all_reduce(e, buf, count, dtype);
  NCCLC(g_rccl.AllReduce(buf, buf, count, dtype == 1 ? ncclDouble : ncclFloat, ncclSum, e->comm, e->stream));
  return MOBROB_OK;
}
}  // namespace

namespace {
// ---- one co-operative launch per epoch (kernels_epoch64.h) ----
// Eligible: single rank, 64-wide fused engine, every minibatch of the epoch small enough for the split-tile gradient kernel
// (fused64_minibatch_grad's `split`), norm records in use (no target_kl: its stop needs a host read per step), only the dominant
// kernel bracketed when profiling, grid <= compute units.  *grid_out: workgroups of the launch.
bool epoch_kernel_eligible(mobrob_ppo_engine* e, bool dp, int* grid_out) {
  static_assert(kEpochRedBlocks * 256 >= 10 * 1024 + 200 && (kEpochRedBlocks - 1) * 256 < 10 * 1024 + 200, "kEpochRedBlocks = ceil(s64_size() / 256)");
  const FusedState& f = e->fused;
  if (dp || e->cfg.world_size != 1 || !e->epoch_kernel_on || !f.enabled || f.H != GH || e->epoch_bar == nullptr) return false;
  if (e->target_kl > 0.0 || !e->use_norm_records || kEpochRedBlocks != cdiv(s64_size(), 256)) return false;
  if (e->prof_on && (e->prof_mask & ((1u << MOBROB_K_GRAD_REDUCE) | (1u << MOBROB_K_APPLY))) != 0) return false;   // --phases: the per-step launches are what gets bracketed
  const int total = e->N * e->T;
  const int ntiles = cdiv(std::min(e->Bl, total), GR);
  const int grid = 2 * std::min(f.max_grid / 2, cdiv(ntiles, g_train_waves(e->Dp)));
  const bool split = ntiles <= e->split64_max_tiles && 2 * ntiles <= f.max_grid && ntiles <= (grid / 2) * g_train_waves(e->Dp);
  if (!split) return false;
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->cfg.device_id) != hipSuccess) return false;
  const int G = std::max(2 * ntiles, 2 * kEpochRedBlocks);
  if (G > cus) return false;
  *grid_out = G;
  return true;
}
constexpr size_t kEpochLdsBytes = 84 * 1024;   // more than half of a CU's 160 KB: one workgroup per CU (the residency the hand-offs were validated in)

int launch_epoch_kernel(mobrob_ppo_engine* e, int ep, int G) {
  FusedState& f = e->fused;
  const int total = e->N * e->T, nmb = e->nmb;
  // per-step constants of this epoch: Adam's bias corrections in float64 exactly as apply_adam forms them, and the statistics rows the
  // steps log into (apply_norms' ring)
  const size_t slot_bytes = (size_t)nmb * 12;
  if (!e->epoch_stage) {
    HIPC(hipHostMalloc((void**)&e->epoch_stage, slot_bytes * (size_t)std::max(1, e->cfg.n_epochs), hipHostMallocDefault));
    HIPC(hipHostMalloc((void**)&e->epoch_err_host, sizeof(int), hipHostMallocCoherent | hipHostMallocMapped));
    *e->epoch_err_host = 0;
    HIPC(hipEventCreateWithFlags(&e->epoch_ev, hipEventDisableTiming));
  }
  if (ep == 0) HIPC(hipEventSynchronize(e->epoch_ev));   // the previous train()'s staging copies are done (never recorded: returns at once)
  char* slot = e->epoch_stage + slot_bytes * (size_t)(ep % std::max(1, e->cfg.n_epochs));
  float* consts = reinterpret_cast<float*>(slot);
  int* idx = reinterpret_cast<int*>(slot + (size_t)nmb * 8);
  const double b1 = e->cfg.adam_beta1, b2 = e->cfg.adam_beta2;
  for (int mb = 0; mb < nmb; ++mb) {
    e->adam_step++;
    consts[2 * mb] = (float)(e->cfg.learning_rate / (1.0 - std::pow(b1, (double)e->adam_step)));
    consts[2 * mb + 1] = (float)std::sqrt(1.0 - std::pow(b2, (double)e->adam_step));
    if (e->stats_n >= e->stats_cap) e->stats_n = 0;
    idx[mb] = e->stats_n++;
  }
  HIPC(hipMemcpyAsync(e->epoch_consts, consts, (size_t)nmb * 8, hipMemcpyHostToDevice, e->stream));
  HIPC(hipMemcpyAsync(e->epoch_idx, idx, (size_t)nmb * 4, hipMemcpyHostToDevice, e->stream));
  HIPC(hipEventRecord(e->epoch_ev, e->stream));
  HIPC(hipMemsetAsync(e->epoch_bar, 0, 1024 * sizeof(unsigned), e->stream));

  Epoch64Args ea{};
  {  // gradient phase (fused64_minibatch_grad)
    Fused64TrainArgs& a = ea.tr;
    a.net[0] = f.net[0]; a.net[1] = f.net[1];
    a.obs = e->obs; a.actions = e->actions; a.A = e->A; a.old_logp = e->logp; a.adv = e->adv; a.ret = e->ret;
    a.log_std = e->params + e->offs[T_LOGSTD];
    a.normalize = e->cfg.normalize_advantage;
    a.clip = (float)e->cfg.clip_range; a.vf_coef = (float)e->cfg.vf_coef; a.ent_coef = (float)e->cfg.ent_coef;
    a.clip_vf = (float)e->clip_vf; a.old_values = e->values;
    a.slabs = f.slabs; a.sums = e->grads + e->P; a.stamps = f.stamps;
    a.wpack[0] = reinterpret_cast<const float*>(f.net[0].W1f);
    a.wpack[1] = reinterpret_cast<const float*>(f.net[1].W1f);
  }
  {  // reduction phase
    Slab64ReduceArgs& r = ea.rd;
    r.slabs = f.slabs; r.group = g_train_waves(e->Dp); r.grads = e->grads; r.P = e->P;
    for (int i = 0; i < 14; ++i) r.offs[i] = e->offs[i];
    r.D = e->D; r.A = e->A; r.ent_coef = (float)e->cfg.ent_coef;
    r.sums = e->grads + e->P;
    r.rec_sum = e->norm_rec_sum; r.rec_t = e->norm_rec_t;
  }
  {  // clip + Adam + packs (apply_adam with the norm records)
    AdamPackArgs& a = ea.ad;
    fill_adam_pack_args(e, a);
    StatsArgs st{};
    st.loss_sums = e->grads + e->P; st.log_std = Pp(e, T_LOGSTD);
    st.ent_coef = (float)e->cfg.ent_coef; st.vf_coef = (float)e->cfg.vf_coef; st.n_act = e->A;
    a.partial = e->norm_rec_sum; a.fold_idx = e->fold_idx_dev; a.st = st;
    for (int i = 0; i <= kMaxTensors; ++i) a.fold_start[i] = e->fold_start[i];
    a.loss_sums_zero = e->grads + e->P;
  }
  ea.rows = e->rows; ea.advstat = e->advstat;
  ea.total = total; ea.bl = e->Bl; ea.nmb = nmb; ea.world = e->cfg.world_size;
  ea.step_consts = e->epoch_consts; ea.stats_idx = e->epoch_idx; ea.stats = e->stats;
  ea.barrier = e->epoch_bar; ea.error_host = e->epoch_err_host;
  const double timeout_s = getenv("MOBROB_EPOCH_TIMEOUT_S") ? atof(getenv("MOBROB_EPOCH_TIMEOUT_S")) : 10.0;
  ea.timeout_ticks = (long long)(timeout_s * 1e8);
  void* kargs[1] = {&ea};
  {
    ProfScope ps(e, MOBROB_K_TRAIN_GRAD);   // the whole epoch: gradient, reduction and Adam phases are one launch
    hipError_t le = hipSuccess;
    FUSED_DISPATCH_DP(e->Dp, FUSED64_DISPATCH_NJ(e->A, {
      const void* fn = reinterpret_cast<const void*>(k_epoch64<DPc, NJc>);
      le = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kEpochLdsBytes);
      if (le == hipSuccess) le = hipLaunchCooperativeKernel(fn, dim3(G), dim3(256), kargs, (unsigned)kEpochLdsBytes, e->stream);
    }));
    if (le != hipSuccess) return fail(MOBROB_ERR_HIP, "k_epoch64 launch (%d workgroups): %s", G, hipGetErrorString(le));
  }
  e->cur_count = std::min(e->Bl, total - (nmb - 1) * e->Bl);
  e->grad_pending = false;
  e->last_steps_applied += nmb;
  return MOBROB_OK;
}

// PPO.train() [SB3 ppo/ppo.py], single rank (dp == false) or data parallel (dp == true: the advantage statistics are
// all-reduced once per epoch, the gradient TOGETHER WITH the eight loss sums behind it once per optimizer step, so the
// logged statistics and the target_kl decision are those of the union minibatch on every rank alike).
int train_loop(mobrob_ppo_engine* e, const int64_t* perms, bool dp, mobrob_allreduce_fn fn, void* ctx) {
  const size_t total = (size_t)e->N * e->T;
  e->stats_n = 0;
  const bool kl = e->target_kl > 0.0;
  e->last_epochs_started = 0; e->last_stopped_early = 0; e->last_steps_applied = 0;
  // The reduction kernel leaves per-block (tensor, sum of squares) records and k_adam_pack folds them: no
  // k_sqnorm_chunks launch.  (Round 2 had this for 64-wide nets only: at 2x256 the table has 712 records and the fold
  // was one serial chain per tensor in every k_adam_pack block, 12.9 instead of 11.4 ms per iteration; the fold is now
  // four wave reductions.)  Never under data parallel: the records are norms of the LOCAL gradient, the clip needs
  // those of the summed one.
  struct RecordsOn {  // nothing can touch the gradient between reduction and clip inside this loop
    mobrob_ppo_engine* e;
    RecordsOn(mobrob_ppo_engine* e_, bool dp_) : e(e_) { e->use_norm_records = !dp_ && e->fused.enabled && e->target_kl <= 0.0 && getenv("MOBROB_NO_NORM_RECORDS") == nullptr; }
    ~RecordsOn() { e->use_norm_records = false; }
  } records_on(e, dp);
  int epoch_grid = 0;
  const bool epoch_kernel = epoch_kernel_eligible(e, dp, &epoch_grid);
  e->last_update_mode = epoch_kernel ? 1 : 0;
  for (int ep = 0; ep < e->cfg.n_epochs; ++ep) {
    CHK(mobrob_ppo_epoch_begin(e, perms ? perms + (size_t)ep * total : nullptr));
    // per-minibatch (sum, sum of squares, count) of the advantages: global statistics for the normalisation
    if (dp) CHK(dp_all_reduce(e, e->advstat, (size_t)e->nmb * 4, 1, fn, ctx));
    e->last_epochs_started = ep + 1;
    if (ep == e->cfg.n_epochs - 1 || kl) e->stats_n = 0;  // the rows kept are those of the last epoch that ran
    if (epoch_kernel) {   // all optimizer steps of the epoch in ONE co-operative launch (kernels_epoch64.h): same arithmetic, same bits
      CHK(launch_epoch_kernel(e, ep, epoch_grid));
      continue;
    }
    for (int mb = 0; mb < e->nmb && !e->last_stopped_early; ++mb) {
      CHK(mobrob_ppo_minibatch_grad(e, mb));
      // THE exchange step, one per optimizer step: [P] gradient + [8] loss sums (policy, value, approx_kl, clip
      // fraction, row count ...) in one message
      if (dp) CHK(dp_all_reduce(e, e->grads, (size_t)e->P + 8, 0, fn, ctx));
      int32_t stopped = 0;  // target_kl [SB3 PPO.train]: this step and everything after it is dropped
      CHK(apply_checked(e, &stopped));
      if (stopped) {
        e->last_stopped_early = 1;
        break;
      }
      e->last_steps_applied++;
    }
    if (e->last_stopped_early) break;
  }
  e->epoch_open = false;
  return MOBROB_OK;
}
}  // namespace

extern "C" {

int mobrob_ppo_train_enqueue(mobrob_ppo_engine_t* e, const int64_t* perms) {
  if (!e) return fail(MOBROB_ERR_INVALID, "null engine");
  if (e->cfg.world_size != 1)
    return fail(MOBROB_ERR_STATE, "mobrob_ppo_train is the single-rank loop; data-parallel ranks call mobrob_ppo_train_dp "
                                  "(or drive epoch_begin/minibatch_grad/[all-reduce]/minibatch_apply)");
  return train_loop(e, perms, false, nullptr, nullptr);
}

int mobrob_ppo_comm_unique_id(uint8_t* out128) {
  if (!out128) return fail(MOBROB_ERR_INVALID, "comm_unique_id: null argument");
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  CHK(rccl_load());
  ncclUniqueId id;
  NCCLC(g_rccl.GetUniqueId(&id));
  memcpy(out128, &id, sizeof id);
  return MOBROB_OK;
}
int mobrob_ppo_comm_prepare(mobrob_ppo_engine_t* e) {
  if (!e) return fail(MOBROB_ERR_INVALID, "comm_prepare: null engine");
  if (e->comm) return fail(MOBROB_ERR_STATE, "comm_prepare: the engine already has a communicator");
  CHK(rccl_load());
  HIPC(hipSetDevice(e->cfg.device_id));
  return MOBROB_OK;
}
int mobrob_ppo_comm_init_rank(mobrob_ppo_engine_t* e, const uint8_t* id128, int32_t rank, int32_t nranks) {
  if (!e || !id128) return fail(MOBROB_ERR_INVALID, "comm_init: null argument");
  if (nranks != e->cfg.world_size)
    return fail(MOBROB_ERR_INVALID, "comm_init: %d ranks, but the engine splits its minibatch for world_size %d", nranks, e->cfg.world_size);
  if (rank < 0 || rank >= nranks) return fail(MOBROB_ERR_INVALID, "comm_init: rank %d of %d", rank, nranks);
  CHK(mobrob_ppo_comm_prepare(e));
  ncclUniqueId id;
  memcpy(&id, id128, sizeof id);
  NCCLC(g_rccl.CommInitRank(&e->comm, nranks, id, rank));
  return MOBROB_OK;
}
int mobrob_ppo_comm_init(mobrob_ppo_engine_t* e, const uint8_t* id128) {
  if (!e) return fail(MOBROB_ERR_INVALID, "comm_init: null argument");
  return mobrob_ppo_comm_init_rank(e, id128, e->cfg.rank, e->cfg.world_size);
}
int mobrob_ppo_comm_info(mobrob_ppo_engine_t* e, int32_t* nranks, int32_t* rank) {
  if (!e) return fail(MOBROB_ERR_INVALID, "null engine");
  int n = 0, r = -1;
  if (e->comm) {
    NCCLC(g_rccl.CommCount(e->comm, &n));
    NCCLC(g_rccl.CommUserRank(e->comm, &r));
  }
  if (nranks) *nranks = n;
  if (rank) *rank = r;
  return MOBROB_OK;
}
int mobrob_ppo_comm_destroy(mobrob_ppo_engine_t* e) {
  if (!e) return fail(MOBROB_ERR_INVALID, "null engine");
  if (e->comm) {
    (void)hipStreamSynchronize(e->stream);
    (void)g_rccl.CommDestroy(e->comm);
    e->comm = nullptr;
  }
  return MOBROB_OK;
}
int mobrob_ppo_oneshot_export(mobrob_ppo_engine_t* e, uint8_t* handle64) {
  if (!e || !handle64) return fail(MOBROB_ERR_INVALID, "oneshot_export: null argument");
  static_assert(sizeof(hipIpcMemHandle_t) == MOBROB_IPC_HANDLE_BYTES, "hipIpcMemHandle_t is 64 bytes");
  auto& o = e->oneshot;
  HIPC(hipSetDevice(e->cfg.device_id));
  if (!o.xbuf) {
    o.payload = oneshot_payload_bytes(e);
    if (cdiv((int)o.payload, kOneShotChunkBytes) > kOneShotMaxChunks)
      return fail(MOBROB_ERR_INVALID, "one-shot all-reduce: a %zu-byte message needs more than %d chunks", o.payload, kOneShotMaxChunks);
    const size_t total = 2 * o.payload + kOneShotMaxChunks * sizeof(unsigned long long);
    HIPC(hipMalloc((void**)&o.xbuf, total));
    // flags = 0 < every sequence number, complete before a peer can hold the handle: ordered on the engine's stream and waited for
    // (hipMemset on device memory is neither guaranteed host-synchronous nor ordered against a non-blocking stream)
    HIPC(hipMemsetAsync(o.xbuf, 0, total, e->stream));
    HIPC(hipStreamSynchronize(e->stream));
    HIPC(hipHostMalloc((void**)&o.error, sizeof(int), hipHostMallocDefault));
    *o.error = 0;
    const char* t = getenv("MOBROB_ONESHOT_TIMEOUT_MS");
    o.timeout_ticks = (long long)(t ? atoll(t) : 20000) * 100000;  // wall_clock64 counts at 100 MHz
  }
  hipIpcMemHandle_t h;
  HIPC(hipIpcGetMemHandle(&h, o.xbuf));
  memcpy(handle64, &h, sizeof h);
  return MOBROB_OK;
}
int mobrob_ppo_oneshot_open(mobrob_ppo_engine_t* e, const uint8_t* handles, int32_t rank, int32_t nranks) {
  if (!e || !handles) return fail(MOBROB_ERR_INVALID, "oneshot_open: null argument");
  auto& o = e->oneshot;
  if (!o.xbuf) return fail(MOBROB_ERR_STATE, "oneshot_open before oneshot_export");
  if (o.ready) return fail(MOBROB_ERR_STATE, "oneshot_open: the exchange is already open");
  if (nranks != e->cfg.world_size || nranks > kOneShotMaxRanks || rank < 0 || rank >= nranks)
    return fail(MOBROB_ERR_INVALID, "oneshot_open: rank %d of %d (engine world_size %d, at most %d ranks)", rank, nranks, e->cfg.world_size, kOneShotMaxRanks);
  HIPC(hipSetDevice(e->cfg.device_id));
  for (int r = 0; r < nranks; ++r) {
    if (r == rank) { o.peer[r] = o.xbuf; continue; }
    hipIpcMemHandle_t h;
    memcpy(&h, handles + (size_t)r * MOBROB_IPC_HANDLE_BYTES, sizeof h);
    void* p = nullptr;
    hipError_t er = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (er != hipSuccess) {
      (void)hipGetLastError();
      for (int q = 0; q < r; ++q)
        if (q != rank && o.peer[q]) { (void)hipIpcCloseMemHandle(o.peer[q]); o.peer[q] = nullptr; }
      return fail(MOBROB_ERR_HIP, "hipIpcOpenMemHandle(rank %d): %s", r, hipGetErrorString(er));
    }
    o.peer[r] = static_cast<char*>(p);
  }
  o.world = nranks; o.rank = rank; o.seq = 0; o.ready = true;
  return MOBROB_OK;
}
int mobrob_ppo_oneshot_close(mobrob_ppo_engine_t* e) {
  if (!e) return fail(MOBROB_ERR_INVALID, "null engine");
  auto& o = e->oneshot;
  if (!o.xbuf) return MOBROB_OK;
  (void)hipStreamSynchronize(e->stream);
  for (int r = 0; r < o.world; ++r)
    if (r != o.rank && o.peer[r]) (void)hipIpcCloseMemHandle(o.peer[r]);
  (void)hipFree(o.xbuf);
  (void)hipHostFree(o.error);
  o = mobrob_ppo_engine::OneShot{};
  return MOBROB_OK;
}
int mobrob_ppo_train_dp(mobrob_ppo_engine_t* e, const int64_t* perms, mobrob_allreduce_fn fn, void* ctx) {
  if (!e) return fail(MOBROB_ERR_INVALID, "null engine");
  if (!fn && !e->comm && !e->oneshot.ready)
    return fail(MOBROB_ERR_STATE, "train_dp: no communicator (mobrob_ppo_comm_init), no one-shot exchange (mobrob_ppo_oneshot_open) and no all-reduce callback");
  CHK(train_loop(e, perms, true, fn, ctx));
  if (!fn && e->oneshot.ready) {
    // A one-shot message whose peer never published leaves this rank's gradient LOCAL: such a step must not pass as a data-parallel
    // one.  The update is awaited here and the error word turned into a failure of this call (the caller ends the job).
    HIPC(hipStreamSynchronize(e->stream));
    CHK(check_async_error(e));
  }
  return MOBROB_OK;
}

extern "C++" {   // templates: C++ linkage inside the C-ABI block
namespace {
// exchange self-check (mobrob_ppo_exchange_selfcheck): small integers -- every partial sum is exact in float32 (and float64) whatever
// the order; `salt` makes every message of a check a different vector
template <typename T>
__device__ __forceinline__ T selfcheck_value(unsigned i, int rank, unsigned salt) {
  const unsigned hsh = ((i + 0x51ED27u * salt) * 2654435761u + (unsigned)(rank + 1) * 0x9E3779B9u) >> 20;
  return (T)((int)(hsh & 0xFFFu) - 2048);
}
template <typename T>
__global__ void k_selfcheck_fill(T* buf, int n, int rank, unsigned salt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) buf[i] = selfcheck_value<T>((unsigned)i, rank, salt);
}
template <typename T>
__global__ void k_selfcheck_compare(const T* buf, int n, int world, unsigned salt, int* mismatches) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  T want = 0;
  for (int r = 0; r < world; ++r) want += selfcheck_value<T>((unsigned)i, r, salt);   // rank order; exact anyway
  if (!(buf[i] == want)) atomicAdd(mismatches, 1);
}
}  // namespace
}  // extern "C++"

// Known vectors through the exchange train_dp would use, compared with the rank-ordered sums -- at communicator set-up, before any
// gradient depends on it.  which: 0 = the engine's RCCL communicator, 1 = the one-shot exchange.  Collective over the ranks.
// The update loop sends two kinds of message -- the [P + 8] float gradient message per optimizer step and the [n_minibatches][4]
// float64 advantage sums per epoch -- and the one-shot exchange alternates between two payload slots: the check sends TWO messages
// of EACH kind (both slots, slot reuse, both element types).  It runs on a scratch buffer of its own, so the gradient vector and
// the loss accumulators behind it are never touched; it is refused while an epoch is open or a gradient awaits its apply.
// *mismatches = elements (over the four messages) that are not bit-equal to the expected sum (0 = the exchange is sound).
int mobrob_ppo_exchange_selfcheck(mobrob_ppo_engine_t* e, int32_t which, int32_t* mismatches) {
  if (!e || !mismatches) return fail(MOBROB_ERR_INVALID, "exchange_selfcheck: null argument");
  *mismatches = -1;
  if (e->epoch_open || e->grad_pending)
    return fail(MOBROB_ERR_STATE, "exchange_selfcheck: an epoch is open or a gradient is pending (call it between updates)");
  int world = 0, rank = -1;
  if (which == 0) {
    if (!e->comm) return fail(MOBROB_ERR_STATE, "exchange_selfcheck: no RCCL communicator");
    NCCLC(g_rccl.CommCount(e->comm, &world));
    NCCLC(g_rccl.CommUserRank(e->comm, &rank));
  } else if (which == 1) {
    if (!e->oneshot.ready) return fail(MOBROB_ERR_STATE, "exchange_selfcheck: the one-shot exchange is not open");
    world = e->oneshot.world; rank = e->oneshot.rank;
  } else {
    return fail(MOBROB_ERR_INVALID, "exchange_selfcheck: which = %d", which);
  }
  const int nf = e->P + 8, nd = e->nmb * 4;
  char* scratch = nullptr;
  int* bad = nullptr;
  const size_t sbytes = std::max((size_t)nf * sizeof(float), (size_t)nd * sizeof(double));
  HIPC(hipMalloc((void**)&scratch, sbytes + 256));
  if (hipMalloc((void**)&bad, sizeof(int)) != hipSuccess) { (void)hipFree(scratch); return fail(MOBROB_ERR_HIP, "exchange_selfcheck: hipMalloc"); }
  (void)hipMemsetAsync(bad, 0, sizeof(int), e->stream);
  int rc = MOBROB_OK;
  for (unsigned msg = 0; msg < 4 && rc == MOBROB_OK; ++msg) {   // float, float, double, double
    const int dtype = msg >> 1, n = dtype ? nd : nf;
    if (dtype) hipLaunchKernelGGL(k_selfcheck_fill<double>, dim3(cdiv(n, 256)), dim3(256), 0, e->stream, reinterpret_cast<double*>(scratch), n, rank, msg);
    else hipLaunchKernelGGL(k_selfcheck_fill<float>, dim3(cdiv(n, 256)), dim3(256), 0, e->stream, reinterpret_cast<float*>(scratch), n, rank, msg);
    if (which == 1) rc = oneshot_all_reduce(e, scratch, (size_t)n, dtype);
    else if (g_rccl.AllReduce(scratch, scratch, (size_t)n, dtype ? ncclDouble : ncclFloat, ncclSum, e->comm, e->stream) != ncclSuccess)
      rc = fail(MOBROB_ERR_HIP, "exchange_selfcheck: ncclAllReduce failed");
    if (rc != MOBROB_OK) break;
    if (dtype) hipLaunchKernelGGL(k_selfcheck_compare<double>, dim3(cdiv(n, 256)), dim3(256), 0, e->stream, reinterpret_cast<const double*>(scratch), n, world, msg, bad);
    else hipLaunchKernelGGL(k_selfcheck_compare<float>, dim3(cdiv(n, 256)), dim3(256), 0, e->stream, reinterpret_cast<const float*>(scratch), n, world, msg, bad);
  }
  if (rc == MOBROB_OK) {
    int h = -1;
    hipError_t er = hipMemcpyAsync(&h, bad, sizeof(int), hipMemcpyDeviceToHost, e->stream);
    if (er == hipSuccess) er = hipStreamSynchronize(e->stream);
    if (er != hipSuccess) rc = fail(MOBROB_ERR_HIP, "exchange_selfcheck: %s", hipGetErrorString(er));
    else *mismatches = h;
  }
  (void)hipStreamSynchronize(e->stream);
  (void)hipFree(bad);
  (void)hipFree(scratch);
  if (rc != MOBROB_OK) return rc;
  // a peer that never published raised the error word: reported here, and LEFT SET for the closing path only if the caller ignores
  // this return value -- check_async_error clears it once it has been turned into a failure
  if (which == 1) CHK(check_async_error(e));
  return MOBROB_OK;
}
int mobrob_ppo_allreduce_counters(mobrob_ppo_engine_t* e, int64_t* calls, int64_t* bytes, int32_t reset) {
  if (!e) return fail(MOBROB_ERR_INVALID, "null engine");
  if (calls) *calls = e->allreduce_calls;
  if (bytes) *bytes = e->allreduce_bytes;
  if (reset) { e->allreduce_calls = 0; e->allreduce_bytes = 0; }
  return MOBROB_OK;
}

int mobrob_ppo_train(mobrob_ppo_engine_t* e, const int64_t* perms, mobrob_ppo_train_stats_t* st) {
  if (!e) return fail(MOBROB_ERR_INVALID, "null engine");
  // An update that will run its epochs as co-operative launches (k_epoch64) is taken from a SNAPSHOT of the optimizer's state: should a
  // launch give up at a grid barrier (a workgroup that never became resident: more spinning tenants on the device than it has room for),
  // the state is restored and the whole update runs again as three launches per step -- the same bits, never a failed or half-applied
  // train().  The engine then keeps the three launches (mobrob_ppo_update_mode reports it).  (mobrob_ppo_train_enqueue has no such
  // retry: it returns before the outcome is known; there the abort fails the next synchronising call.)
  int grid_unused = 0;
  struct RecordsProbe {   // eligibility as train_loop will see it (it switches use_norm_records on for the loop's duration)
    mobrob_ppo_engine* e; bool old;
    explicit RecordsProbe(mobrob_ppo_engine* e_) : e(e_), old(e_->use_norm_records) { e->use_norm_records = e->fused.enabled && e->target_kl <= 0.0 && getenv("MOBROB_NO_NORM_RECORDS") == nullptr; }
    ~RecordsProbe() { e->use_norm_records = old; }
  };
  bool snap = false;
  {
    RecordsProbe rp(e);
    snap = e->cfg.world_size == 1 && e->epoch_snap != nullptr && epoch_kernel_eligible(e, false, &grid_unused);
  }
  const int64_t adam_step0 = e->adam_step;
  const uint64_t perm_counter0 = e->perm_counter;
  if (snap) {
    const size_t pb = (size_t)e->P * sizeof(float);
    HIPC(hipMemcpyAsync(e->epoch_snap, e->params, pb, hipMemcpyDeviceToDevice, e->stream));
    HIPC(hipMemcpyAsync(e->epoch_snap + e->P, e->m, pb, hipMemcpyDeviceToDevice, e->stream));
    HIPC(hipMemcpyAsync(e->epoch_snap + 2 * (size_t)e->P, e->v, pb, hipMemcpyDeviceToDevice, e->stream));
  }
  CHK(mobrob_ppo_train_enqueue(e, perms));
  if (snap && (e->last_update_mode & 1)) {
    HIPC(hipStreamSynchronize(e->stream));
    if (e->epoch_err_host && *(volatile int*)e->epoch_err_host != 0) {
      *(volatile int*)e->epoch_err_host = 0;
      fprintf(stderr, "[mobrob_ppo] the co-operative epoch kernel gave up at a grid barrier (workgroups not resident together: other spinning tenants "
                      "on the device?); the update is re-run as three launches per optimizer step and this engine keeps that form\n");
      const size_t pb = (size_t)e->P * sizeof(float);
      HIPC(hipMemcpyAsync(e->params, e->epoch_snap, pb, hipMemcpyDeviceToDevice, e->stream));
      HIPC(hipMemcpyAsync(e->m, e->epoch_snap + e->P, pb, hipMemcpyDeviceToDevice, e->stream));
      HIPC(hipMemcpyAsync(e->v, e->epoch_snap + 2 * (size_t)e->P, pb, hipMemcpyDeviceToDevice, e->stream));
      HIPC(hipMemsetAsync(e->grads + e->P, 0, 8 * sizeof(float), e->stream));
      repack(e);   // every weight pack from the restored parameters
      e->adam_step = adam_step0; e->perm_counter = perm_counter0;   // (the same permutations again; the parity of the max-|adv| words simply goes on)
      e->epoch_kernel_on = false;
      CHK(mobrob_ppo_train_enqueue(e, perms));
    }
  }
  if (st) {
    std::vector<float> rows((size_t)e->nmb * 8);
    const int n = mobrob_ppo_fetch_step_stats(e, rows.data(), e->nmb);
    if (n < 0) return n;
    double acc[7] = {0};
    int n_norm = 0;  // a step dropped by target_kl logs its losses but has no gradient norm (NaN)
    for (int i = 0; i < n; ++i) {
      for (int k = 0; k < 6; ++k) acc[k] += rows[(size_t)i * 8 + k];
      if (!std::isnan(rows[(size_t)i * 8 + 6])) { acc[6] += rows[(size_t)i * 8 + 6]; n_norm++; }
    }
    const double d = n > 0 ? n : 1;
    st->policy_loss = (float)(acc[0] / d); st->value_loss = (float)(acc[1] / d); st->entropy_loss = (float)(acc[2] / d);
    st->loss = (float)(acc[3] / d); st->approx_kl = (float)(acc[4] / d); st->clip_fraction = (float)(acc[5] / d);
    st->grad_norm = (float)(acc[6] / (n_norm > 0 ? n_norm : 1)); st->n_minibatches = e->last_steps_applied;
  } else {
    HIPC(hipStreamSynchronize(e->stream));
  }
  return MOBROB_OK;
}

// ---- inference --------------------------------------------------------------------------------------
int mobrob_ppo_predict(mobrob_ppo_engine_t* e, const float* obs, int32_t n, int32_t deterministic, const float* eps,
                       float* actions, float* values) {
  if (!e || !obs || n < 1) return fail(MOBROB_ERR_INVALID, "predict: bad argument");
  for (int s = 0; s < n; s += e->rows_max) {
    const int c = std::min(e->rows_max, n - s);
    CHK(upload_obs(e, obs + (size_t)s * e->D, e->pred_obs, c));
    forward(e, e->pred_obs, c, actions != nullptr, e->mu, values != nullptr, e->vout);
    if (actions) {
      float* scratch = e->pred_act;
      if (deterministic) {
        hipLaunchKernelGGL(k_clip_mean, dim3(cdiv(c * e->A, 256)), dim3(256), 0, e->stream, e->mu, e->Ap, c, e->A,
                           (float)e->cfg.action_low, (float)e->cfg.action_high, scratch);
      } else {
        if (e->sde) {   // get_noise: the environments' own matrices for a batch of n_envs rows, else the single exploration_mat
          const bool own = n == e->N;
          sde_variance(e, c, true);
          hipLaunchKernelGGL(k_sample_sde, dim3(cdiv(c, 4)), dim3(256), 0, e->stream, e->mu, e->Ap, e->hp[e->Lp - 1], e->HL, e->sde_var, e->Ap,
                             own ? e->sde_E : e->sde_E1, s, own ? 0 : 1, c, e->HL, e->A, (float)e->cfg.action_low, (float)e->cfg.action_high,
                             (float*)nullptr, scratch, (float*)nullptr);
          HIPC(hipMemcpyAsync(actions + (size_t)s * e->A, scratch, (size_t)c * e->A * 4, hipMemcpyDeviceToHost, e->stream));
          if (values) HIPC(hipMemcpyAsync(values + s, e->vout, (size_t)c * 4, hipMemcpyDeviceToHost, e->stream));
          HIPC(hipStreamSynchronize(e->stream));
          continue;
        }
        const float* epsd = nullptr;
        if (eps) {
          HIPC(hipMemcpyAsync(e->eps_dev, eps + (size_t)s * e->A, (size_t)c * e->A * 4, hipMemcpyHostToDevice, e->stream));
          epsd = e->eps_dev;
        }
        hipLaunchKernelGGL(k_sample, dim3(cdiv(c, 256)), dim3(256), 0, e->stream, e->mu, e->Ap, Pp(e, T_LOGSTD), epsd,
                           c, e->A, (float)e->cfg.action_low, (float)e->cfg.action_high, eps_seed(e), e->draw_counter,
                           (const uint32_t*)nullptr, (float*)nullptr, scratch, (float*)nullptr);
        e->draw_counter++;
      }
      HIPC(hipMemcpyAsync(actions + (size_t)s * e->A, scratch, (size_t)c * e->A * 4, hipMemcpyDeviceToHost, e->stream));
    }
    if (values) HIPC(hipMemcpyAsync(values + s, e->vout, (size_t)c * 4, hipMemcpyDeviceToHost, e->stream));
    HIPC(hipStreamSynchronize(e->stream));
  }
  return MOBROB_OK;
}

// ---- buffers ----------------------------------------------------------------------------------------
static bool feeds_train_records(int32_t which) {
  return which == MOBROB_BUF_ACTIONS || which == MOBROB_BUF_VALUES || which == MOBROB_BUF_LOG_PROBS ||
         which == MOBROB_BUF_ADVANTAGES || which == MOBROB_BUF_RETURNS;
}

static int buffer_lookup(mobrob_ppo_engine* e, int32_t which, void** ptr, size_t* bytes) {
  if (!e) return fail(MOBROB_ERR_INVALID, "null engine");
  const size_t N = e->N, T = e->T;
  void* p = nullptr;
  size_t b = 0;
  switch (which) {
    case MOBROB_BUF_OBS: p = e->obs; b = (T + 1) * N * e->Dp * 4; break;
    case MOBROB_BUF_ACTIONS: p = e->actions; b = T * N * e->A * 4; break;
    case MOBROB_BUF_REWARDS: p = e->rewards; b = T * N * 4; break;
    case MOBROB_BUF_EPISODE_STARTS: p = e->es; b = T * N * 4; break;
    case MOBROB_BUF_VALUES: p = e->values; b = T * N * 4; break;
    case MOBROB_BUF_LOG_PROBS: p = e->logp; b = T * N * 4; break;
    case MOBROB_BUF_ADVANTAGES: p = e->adv; b = T * N * 4; break;
    case MOBROB_BUF_RETURNS: p = e->ret; b = T * N * 4; break;
    case MOBROB_BUF_PARAMS: p = e->params; b = (size_t)e->P * 4; break;
    case MOBROB_BUF_GRADS: p = e->grads; b = (size_t)e->P * 4; break;
    case MOBROB_BUF_ADVSTAT: p = e->advstat; b = (size_t)e->nmb * 4 * 8; break;
    case MOBROB_BUF_LAST_VALUES: p = e->last_values; b = N * 4; break;
    case MOBROB_BUF_LAST_DONES: p = e->last_dones; b = N * 4; break;
    case MOBROB_BUF_CLIPPED_ACTIONS: p = e->clip_act; b = N * e->A * 4; break;
    case MOBROB_BUF_EPISODE_START_STATE: p = e->prev_dones; b = N * 4; break;
    case MOBROB_BUF_TERMINAL_OBS: p = e->term_obs; b = N * e->Dp * 4; break;
    case MOBROB_BUF_TERMINAL_VALUES: p = e->term_val; b = N * 4; break;
    case MOBROB_BUF_TRUNCATED: p = e->trunc_dev; b = N; break;
    case MOBROB_BUF_ENV_STATE: p = e->gstate[0]; b = N * kGoalStateFloats * 4; break;
    case MOBROB_BUF_GRAD_EXCHANGE: p = e->grads; b = (size_t)(e->P + 8) * 4; break;
    case MOBROB_BUF_SDE_NOISE:
      if (!e->sde) return fail(MOBROB_ERR_STATE, "the engine was not created with use_sde");
      p = e->sde_E; b = N * (size_t)e->HL * e->A * 4; break;
    default: return fail(MOBROB_ERR_INVALID, "unknown buffer id %d", which);
  }
  if (ptr) *ptr = p;
  if (bytes) *bytes = b;
  return MOBROB_OK;
}

int mobrob_ppo_buffer_info(mobrob_ppo_engine_t* e, int32_t which, void** ptr, size_t* bytes) {
  CHK(buffer_lookup(e, which, ptr, bytes));
  if (ptr && feeds_train_records(which)) e->train_rec_external = true;  // see ensure_train_records
  return MOBROB_OK;
}

int mobrob_ppo_read_buffer(mobrob_ppo_engine_t* e, int32_t which, void* host, size_t bytes) {
  if (!e || !host) return fail(MOBROB_ERR_INVALID, "read_buffer: null argument");
  void* p; size_t b;
  CHK(buffer_lookup(e, which, &p, &b));
  if (which == MOBROB_BUF_OBS) {  // host layout [T+1][N][D]
    const size_t rows = (size_t)(e->T + 1) * e->N;
    if (bytes != rows * e->D * 4) return fail(MOBROB_ERR_INVALID, "read_buffer(OBS): expected %zu bytes", rows * e->D * 4);
    HIPC(hipMemcpy2DAsync(host, (size_t)e->D * 4, p, (size_t)e->Dp * 4, (size_t)e->D * 4, rows, hipMemcpyDeviceToHost, e->stream));
  } else {
    if (bytes != b) return fail(MOBROB_ERR_INVALID, "read_buffer(%d): expected %zu bytes, got %zu", which, b, bytes);
    HIPC(hipMemcpyAsync(host, p, b, hipMemcpyDeviceToHost, e->stream));
  }
  HIPC(hipStreamSynchronize(e->stream));
  return MOBROB_OK;
}

int mobrob_ppo_write_buffer(mobrob_ppo_engine_t* e, int32_t which, const void* host, size_t bytes) {
  if (!e || !host) return fail(MOBROB_ERR_INVALID, "write_buffer: null argument");
  void* p; size_t b;
  CHK(buffer_lookup(e, which, &p, &b));
  if (feeds_train_records(which)) e->train_rec_valid = false;
  if (which == MOBROB_BUF_OBS) {
    const size_t rows = (size_t)(e->T + 1) * e->N;
    if (bytes != rows * e->D * 4) return fail(MOBROB_ERR_INVALID, "write_buffer(OBS): expected %zu bytes", rows * e->D * 4);
    HIPC(hipMemcpy2DAsync(p, (size_t)e->Dp * 4, host, (size_t)e->D * 4, (size_t)e->D * 4, rows, hipMemcpyHostToDevice, e->stream));
  } else {
    if (bytes != b) return fail(MOBROB_ERR_INVALID, "write_buffer(%d): expected %zu bytes, got %zu", which, b, bytes);
    HIPC(hipMemcpyAsync(p, host, b, hipMemcpyHostToDevice, e->stream));
    if (which == MOBROB_BUF_PARAMS) repack(e);
  }
  HIPC(hipStreamSynchronize(e->stream));
  return MOBROB_OK;
}

int mobrob_ppo_feistel_permutation(mobrob_ppo_engine_t* e, int64_t n, uint64_t key, int64_t* out) {
  if (!e || !out || n < 1 || n > ((int64_t)1 << 30)) return fail(MOBROB_ERR_INVALID, "feistel_permutation: bad argument");
  int64_t* d = nullptr;
  HIPC(hipMalloc(&d, (size_t)n * 8));
  hipLaunchKernelGGL(k_perm_feistel, dim3(cdiv((int)n, 256)), dim3(256), 0, e->stream, (int)n, 1, (int)n,
                     feistel_half_bits((uint64_t)n), (uint32_t)key, (uint32_t)(key >> 32), (int*)nullptr, d);
  hipError_t r = hipMemcpyAsync(out, d, (size_t)n * 8, hipMemcpyDeviceToHost, e->stream);
  if (r == hipSuccess) r = hipStreamSynchronize(e->stream);
  (void)hipFree(d);
  if (r != hipSuccess) return fail(MOBROB_ERR_HIP, "feistel_permutation: %s", hipGetErrorString(r));
  return MOBROB_OK;
}

// ---- profiling --------------------------------------------------------------------------------------
int mobrob_ppo_profile_enable(mobrob_ppo_engine_t* e, int32_t on) {
  if (!e) return fail(MOBROB_ERR_INVALID, "null engine");
  HIPC(hipStreamSynchronize(e->stream));
  prof_resolve(e);
  e->prof_on = on != 0;
  e->prof_mask = on == 1 ? ~0u : (uint32_t)on >> 1;  // 1: every phase; otherwise bit (id + 1) selects phase id
  for (int i = 0; i < MOBROB_K_COUNT; ++i) { e->prof_ms[i] = 0; e->prof_calls[i] = 0; }
  return MOBROB_OK;
}
int mobrob_ppo_profile_read(mobrob_ppo_engine_t* e, double* ms, int64_t* calls) {
  if (!e) return fail(MOBROB_ERR_INVALID, "null engine");
  HIPC(hipStreamSynchronize(e->stream));
  prof_resolve(e);
  for (int i = 0; i < MOBROB_K_COUNT; ++i) {
    if (ms) ms[i] = e->prof_ms[i];
    if (calls) calls[i] = e->prof_calls[i];
  }
  return MOBROB_OK;
}

#if defined(MOBROB_STAMPS) || defined(MOBROB_PAIR_STAMPS) || defined(MOBROB_EPOCH_STAMPS)
// diagnostic build only (not part of include/mobrob_ppo.h)
int mobrob_dbg_read_stamps(mobrob_ppo_engine_t* e, unsigned long long* out32, int reset) {
  HIPC(hipStreamSynchronize(e->stream));
  HIPC(hipMemcpy(out32, e->fused.stamps, 32 * 8, hipMemcpyDeviceToHost));
  if (reset) HIPC(hipMemset(e->fused.stamps, 0, 32 * 8));
  return 0;
}
#endif

}  // extern "C"
