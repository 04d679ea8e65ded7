// Persistent device rollout (hidden width 256): ONE launch runs all T steps of collect_rollouts for the block's 32
// environments -- policy forward (MFMA, as k_fused_act) -> Gaussian sample + log-prob -> env step (synthetic source
// or goal environment) with VecEnv auto-reset -> rollout_buffer.add scalars -> time-limit bootstrap -> next step.
//
// Why it is legal: environments are independent and the policy weights are constant during a rollout, so a block
// never needs anything another block produced -- no grid synchronisation, no kernel boundary per step (a boundary
// costs ~4-5 us of drain/fill even inside a hipGraph; at 1000 steps x 2 kernels that was 40 % of the rollout).
// The observation tile of step t+1 is produced in LDS by the env phase of step t (and streamed to the rollout
// buffer for training); episode length / goal state / episode-start flags live in LDS for the whole launch.
// The value network is NOT evaluated here: V(obs[t]) only feeds GAE, so it is computed afterwards for all T*N
// stored observations in one batched pass (k_value_batch below: 64-row tiles through the training kernel's forward
// code) at throughput instead of per-step latency.
// The rare truncated rows need V(terminal_obs) at once (r += gamma V): the block evaluates the value MLP for
// such a row with all 256 threads (value_row_lds, same arithmetic as the per-step path).
//
// Arithmetic per row and step is that of k_fused_act + k_env_step_store / k_goal_env_step_store; the Philox
// counters are (row, chunk, base + t), so the persistent and the per-step rollouts are bit-identical
// (tests/test_engine_gpu.py::test_persistent_rollout_equals_per_step_rollout).
#pragma once
#include <type_traits>
#include "kernels_env.h"
#include "kernels_fused.h"
#include "kernels_fused64.h"

namespace mobrob {

// timing-only ablation (never in the product build): -DROLL_SKIP=<mask> drops phases of k_rollout_persistent
#ifndef ROLL_SKIP
#define ROLL_SKIP 0
#endif
#define ROLL_ON(bit) (!((ROLL_SKIP) & (bit)))
#define ROLL_TANH(x) (ROLL_ON(32) ? fast_tanh_scaled(x) : (x))

struct RolloutArgs {
  FusedNet pi;                       // policy network packs
  const float* log_std; uint64_t seed; const uint32_t* draw_base; float lo, hi;
  int kind;                          // 1 synthetic source, 2 goal environment
  uint64_t env_seed; const uint32_t* step_base;
  float p_term; int time_limit;
  GoalEnvParams goal;
  BootArgs bt;
  int N, D, A, t0, t1;
  // rollout storage
  float* obs;                        // [T+1][N][DP]
  float* actions; float* logp; float* rewards; float* es;  // [T][N][.]
  float* term_obs; uint8_t* trunc; float* clip_act;        // latest-step buffers
  // carried state (in/out)
  int* ep_len; float* prev_dones; float* gstate; double* ep_stats;
  uint32_t draw0;                    // Philox draw index of step 0 when draw_base is null (host-side counter)
  // kind 3 -- HOST environments served by this launch (k_rollout_persistent<.., 3, true>, round 5): the env phase of a step is the
  // host's: the workgroup hands its rows' clipped actions over in pinned memory, waits for the host to step them, pulls the result.
  const float* h_obs; const float* h_rew; const uint8_t* h_done; const uint8_t* h_trunc; const float* h_term;  // [N][D] | [N] ... (pinned)
  unsigned* h_gpu_flag;              // [workgroups] pinned: t + 1 once the clipped actions of step t of the workgroup's rows are in host memory
  const unsigned* h_host_flag;       // [parts][16] pinned: the first 8 bytes = (truncated rows in the part's newest step) << 32 | steps of the
                                     //                      part the host has finished (low word 0xFFFFFFFF: give up); one 8-byte store
  // abort_dev (below) is followed by one 8-byte relay word per part: the part's first workgroup re-publishes the host's word there
  int* h_error;                      // pinned: set when a workgroup waited longer than timeout_ticks
  int* abort_dev;                    // device word: a launch that gave up tells the launches queued behind it
  int rows_per_part; long long timeout_ticks;   // of wall_clock64() (100 MHz)
};

// LDS carve-up of the rollout tile with the h1 activations kept as three bf16 PLANES (the x3 pieces, split once by the layer-1
// epilogue) instead of one float32 tile: k_rollout_persistent<.., S8 = true>.  Plane p, row r, column k at H1 bytes
// (p 32 + r) * 528 + 2 k: a row stride of 264 bf16 = 132 words shifts consecutive rows by four banks, the conflict-free pattern of
// the float32 tiles' 260-float rows for 16-byte fragment reads.
constexpr int kPlaneLd = FH + 8;                              // bf16 elements per plane row
constexpr int kPlaneFloats = 3 * 32 * kPlaneLd / 2;           // the three planes of a 32-row tile, in floats
template <int DP>
struct Lay32S {
  static constexpr int R = 32;
  static constexpr int LDX = DP + 4;
  static constexpr int X = 0;
  static constexpr int H1 = X + R * LDX;                      // planes [3][32][264] bf16 (scratch of the bootstrap afterwards)
  static constexpr int H2 = H1 + kPlaneFloats;
  static constexpr int DO = H2 + R * FLDH;
  static constexpr int XP = DO + 4 * R * FLDO;                 // the observation tile as three bf16 planes [3][32][DP + 8] (MOBROB_S8_XPLANES)
  static constexpr int XLD = DP + 8;                           // bf16 elements per plane row: 36 words at DP = 64, conflict-free b128 fragment reads
  static constexpr int END = XP + 3 * R * XLD / 2;
};
template <int DP, bool S8 = false>
struct LayRo {
  using B = std::conditional_t<S8, Lay32S<DP>, Lay32<DP>>;
  static constexpr int CA = B::END;         // [32][33] clipped actions of the current step
  static constexpr int ST = CA + 32 * 33;   // [32][16] row state: goal state [0..11] | prev_done [12] | ep_len [13]
  static constexpr int BL = ST + 32 * 16;   // bootstrap list: cnt[4] | row[32] | reward[32]
  static constexpr int AC = BL + 4 + 64;    // per-action constants: sd[32] | 2 sd^2 [32] | log sd [32] | b3[32]
  static constexpr int ZN = AC + 128;       // [2][32][32] standard normals of the sampling stage, double-buffered by step parity
  static constexpr int TM = ZN + 2 * 32 * 32;   // [32][33] log-prob terms of the current step
  static constexpr int EN = TM + 32 * 33;   // [2][32][DP] standard normals of the env phase (observation noise), by step parity
  static constexpr int END = EN + 2 * 32 * DP;
};
inline size_t rollout_lds_bytes(int Dp, bool s8 = false) {
  return fused_lds_act_bytes(Dp) + (s8 ? (size_t)(kPlaneFloats - 32 * FLDH + 3 * 32 * (Dp + 8) / 2) * sizeof(float) : 0) +
         (size_t)(32 * 33 + 32 * 16 + 68 + 128 + 2 * 32 * 32 + 32 * 33 + 2 * 32 * Dp) * sizeof(float);
}
// The workgroup is EIGHT waves: four run the policy forward / sampling / env rules of the tile (one per SIMD, as before),
// four "noise waves" (the second wave of every SIMD) draw the random numbers -- Philox4x32-10 + Box-Muller for
// the sampling normals and for the observation noise of the env phase, 600 draws per step and tile -- into LDS.  Same
// counters, same values as before (and as the per-step kernels); the draws used to sit in front of the GEMMs and inside the
// env phase: 2.9 of 15.2 us per step.  Round 3 drew step t's numbers under the GEMMs of step t; with the x3 GEMMs that costs
// 1.8 us per step un-hidden (the operand splits of the policy waves need the same VALU issue slots).  Round 4: the numbers of
// step t + 1 are drawn during the sampling / env / state phases of step t, when the policy waves' VALU work is light, into
// the other half of a double buffer (the counters depend on (row, chunk, step) only, not on the state).
// (The second draw of a row that ends an episode -- its reset observation -- stays with the policy waves: rare.)
// MOBROB_ROLLOUT_STATIONARY (default 0: measured SLOWER, kept as a switch): the other plan for the same kernel -- FOUR waves, one per SIMD with all 512 registers
// of a lane, and the policy's weights STATIONARY in them for the whole launch: wave w holds the fragment packs of its 64
// columns of W1 (2 x DP/8 x 4 registers), W2 (2 x 32 x 4 = 256) and its K-slice of the head (32): no weight byte crosses the
// L2 <-> CU path in the step loop (round 1-2: 340 KB of fragments per tile and step, 2.9 TB/s chip-wide, and a latency the
// 4-deep prefetch ring only just covered).  Same MFMA sequence per accumulator, same bits.  The noise waves need the second
// wave slot of every SIMD and therefore exclude this plan.  A/B on one box: 15.49 ms per 1000-step rollout of
// 4096 envs against 14.69 ms with streamed fragments + noise waves (15.07 ms in round 2) -- the weight stream was never
// what paced the step; 352 resident registers cost v_accvgpr_read copies in front of the MFMAs instead.
#ifndef MOBROB_S8_STAGGER
#define MOBROB_S8_STAGGER 2
#endif
#ifndef MOBROB_ROLLOUT_STATIONARY
#define MOBROB_ROLLOUT_STATIONARY 0
#endif
constexpr bool kRolloutStationary = MOBROB_ROLLOUT_STATIONARY != 0;
constexpr int kRolloutThreads = kRolloutStationary ? FTHREADS : 2 * FTHREADS;
template <int N>
struct RFrags { f32x4 f[N]; };
template <int N>
__device__ __forceinline__ RFrags<N> rfrags_load(const f32x4* __restrict__ Bp, int lane) {
  RFrags<N> w;
  const unsigned bo = (unsigned)lane * 16u;
#pragma unroll
  for (int kg = 0; kg < N; ++kg) w.f[kg] = ldg16(Bp, bo + (unsigned)kg * 1024u);
  return w;
}
// c0 / c1 += A[32 x 8 NKG] (LDS) . (two 32-column blocks whose fragments sit in registers); k order of gemm_lds_packed_r32
template <int LDA, int NKG>
__device__ __forceinline__ void gemm_two_resident(int a_off, const RFrags<NKG>& wa, const RFrags<NKG>& wb, f32x16& c0, f32x16& c1,
                                                  int lane) {
  const int r = lane & 31, h = lane >> 5;
  const int ab = 4 * opaque((a_off + r * LDA + 4 * h) >> 2);
#pragma unroll
  for (int kg = 0; kg < NKG; ++kg) {
    const f32x4 u = *reinterpret_cast<const f32x4*>(&lds[ab + 8 * kg]);
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) {
      c0 = MFMA32(u[s_], wa.f[kg][s_], c0);
      c1 = MFMA32(u[s_], wb.f[kg][s_], c1);
    }
  }
}

// the launch's RolloutArgs as they sit in the kernarg segment (the kernels below take ONE by-value parameter: offset 0)
typedef const __attribute__((address_space(4))) RolloutArgs* RolloutArgsK;
__device__ __forceinline__ RolloutArgsK rollout_kernargs() {
  RolloutArgsK p = (RolloutArgsK)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));   // opaque per use: the loads behind it are not hoisted out of the step loop
  return p;
}

#ifndef MOBROB_ROLLOUT_BARRIER5
#define MOBROB_ROLLOUT_BARRIER5 0
#endif
#ifndef MOBROB_S8_W1_FIRST
#define MOBROB_S8_W1_FIRST 1
#endif
#ifndef MOBROB_S8_XPLANES   // S8: the observation tile is kept as bf16 planes too (split once by whoever writes it) and layer 1 reads fragments, no VALU
#define MOBROB_S8_XPLANES 1
#endif
// four consecutive columns 4 c .. 4 c + 3 of row rr -> the three planes (each piece: four bf16 = one 8-byte store)
template <class LS>
__device__ __forceinline__ void xplanes_store4(int rr, int c, const f32x4& o) {
  unsigned a1, a2, a3, b1, b2, b3;
  x3_split2(o[0], o[1], a1, a2, a3);
  x3_split2(o[2], o[3], b1, b2, b3);
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  const int f0 = LS::XP + (rr * LS::XLD + 4 * c) / 2;            // float offset of the first plane's quad (XLD and 4 c are even)
  *reinterpret_cast<u32x2*>(&lds[f0]) = u32x2{a1, b1};
  *reinterpret_cast<u32x2*>(&lds[f0 + 32 * LS::XLD / 2]) = u32x2{a2, b2};
  *reinterpret_cast<u32x2*>(&lds[f0 + 64 * LS::XLD / 2]) = u32x2{a3, b3};
}

// KIND: the env source compiled in (1 synthetic, 2 goal environment; = a.kind): one variant carries one env's scalars -- with both in
// one kernel ~200 scalar registers were parked in VGPR lanes and read back (v_readlane) inside the step loop.
// S8 (round 5; x3 engines only): the two hidden-layer GEMMs run on ALL EIGHT waves -- wave w owns the 32 output columns 32 w .. 32 w + 31
// of both layers -- with the two leading bf16 pieces of its W2 block STATIONARY in 128 registers for the whole launch:
//   * What paced the four-wave form: every step streamed the three pieces of W1 and W2 (98 + 393 KB per tile) through the CU's
//     vector memory port, 64 B/clk: 7.7 k cycles per step against 7.7 k cycles of MFMA issue on the four policy waves -- two limits
//     of the same size that do not overlap perfectly (6.98 us measured for 3.5 us of matrix time, profiles/r4/rollout_phase_ablation.txt).
//     Now only piece 2 of W2 (one sixth of the products' operand bytes) and W1 cross the port: 245 KB per step.
//   * The noise waves idle through the GEMM phases anyway (they draw during sampling / env / state phases): their SIMD slots and their
//     256 registers each carry half of the matrix work instead.
//   * h1 is split into its three bf16 pieces ONCE, by the layer-1 epilogue that produces it, and kept as three bf16 planes (Lay32S);
//     layer 2's A fragments are three ds_read_b128 per k step and no VALU (the four-wave form split the same 32 rows in every wave:
//     704 VALU instructions per wave and layer).
// Same products, same order per accumulator as gemm_x3_r32 (X3_MFMA6 over the k steps in natural order): the rollout's numbers are
// those of the four-wave x3 form bit for bit (tests/test_engine_gpu.py::test_s8_rollout_equals_the_four_wave_x3_rollout).
// Sampling, env rules, storage, bootstrap: unchanged, on the four policy waves.
#ifdef MOBROB_SERVE_STAMPS   // diagnostic build: workgroup 0 records wall_clock64() (100 MHz) at six points of its first 64 served steps
#define SERVE_STAMP(k) if (KIND == 3 && blockIdx.x == 0 && t < 64) { if ((threadIdx.x & 63) == 0 && (threadIdx.x >> 6) == 1) reinterpret_cast<long long*>(ar.h_error + 16)[8 * t + (k)] = wall_clock64(); }
#else
#define SERVE_STAMP(k)
#endif
#define ar (*ap_)
// Philox keys inside the step loop: derived from the kernel arguments where they are used.  As loop invariants
// their ten-round key schedules (k + r W, twenty scalars per key pair and stream) were hoisted out of the step loop and parked.
#define EK0 ((uint32_t)ar.env_seed)
#define EK1 ((uint32_t)(ar.env_seed >> 32))
template <int DP, int KIND, bool S8 = false>
__global__ __launch_bounds__(kRolloutThreads, 1) void k_rollout_persistent(RolloutArgs a) {
  static_assert(!S8 || !kRolloutStationary, "S8 needs the eight-wave workgroup");
  static_assert(KIND != 3 || S8, "the host-env form exists for the eight-wave kernel only");
  if constexpr (KIND == 3) {   // a launch in front of this one gave up on the host: do nothing (uniform)
    if (__hip_atomic_load(a.abort_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
  }
  using L = LayRo<DP, S8>;
  using LB = typename L::B;
  constexpr int ldx = LB::LDX, per = DP / 4, R = 32;
  const int tid0 = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const bool noise_wave = !kRolloutStationary && wave >= 4;  // wave-uniform
  const FusedNet W = a.pi;
  // hidden layers on the bf16 matrix pipe with three-way split float32 operands (kernels_fused.h, gemm_x3_r32) when the engine
  // maintains the x3 packs; block-uniform
  const bool x3 = W.W2x != nullptr;
  const u32x4* W1x = reinterpret_cast<const u32x4*>(W.W1x);
  const u32x4* W2x = reinterpret_cast<const u32x4*>(W.W2x);
  const int row0 = blockIdx.x * R;
  const int N = a.N, A = a.A, D = a.D;
  int* cnt = reinterpret_cast<int*>(&lds[L::BL]);
  int* lrow = cnt + 4;
  float* lrew = &lds[L::BL + 4 + 32];
  // ---- carried state and the observation tile of step t0 -> LDS ----
  if (tid0 < R) {
    const int n = row0 + tid0;
    float* S = &lds[L::ST + tid0 * 16];
    if (n < N) {
      if (KIND == 2) {
#pragma unroll
        for (int j = 0; j < kGoalStateFloats; ++j) S[j] = a.gstate[(size_t)n * kGoalStateFloats + j];
      } else if (KIND == 1) {
        reinterpret_cast<int*>(S)[13] = a.ep_len[n];
      }
      S[12] = a.prev_dones[n];
    }
  }
  if (tid0 < 32) {  // per-action constants of the Gaussian head (the same expressions k_fused_act evaluates per row)
    float sd = 1.f, bb = 0.f;
    if (tid0 < A) { sd = expf(a.log_std[tid0]); bb = W.b3[tid0]; }
    lds[L::AC + tid0] = sd;
    lds[L::AC + 32 + tid0] = 2.0f * (sd * sd);
    lds[L::AC + 64 + tid0] = logf(sd);
    lds[L::AC + 96 + tid0] = bb;
  }
  for (int i = tid0; i < R * per; i += kRolloutThreads) {
    const int rr = i / per, c = i - rr * per;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row0 + rr < N)
      v = ldg16(a.obs, (unsigned)((size_t)a.t0 * N + row0 + rr) * (unsigned)(DP * 4) + (unsigned)(c * 16));
    *reinterpret_cast<f32x4*>(&lds[LB::X + rr * ldx + 4 * c]) = v;
    if constexpr (S8 && MOBROB_S8_XPLANES) xplanes_store4<LB>(rr, c, v);
  }
  __syncthreads();
  const uint32_t dbase = a.draw_base ? *a.draw_base : a.draw0, sbase = a.step_base ? *a.step_base : 0u;
  const uint32_t ek0 = (uint32_t)a.env_seed, ek1 = (uint32_t)(a.env_seed >> 32);

  double es_n = 0.0, es_ret = 0.0, es_len = 0.0, es_goal = 0.0;  // finished episodes of this thread's row
  constexpr int NKG1 = DP / 8;
  RFrags<NKG1> w1a, w1b;
  RFrags<32> w2a, w2b;
  RFrags<8> w3s;
  if (kRolloutStationary) {  // this wave's columns 64 wave .. 64 wave + 63 of both hidden layers, K-slice [64 wave, +64) of the head
    const int l0 = tid0 & 63;
    w1a = rfrags_load<NKG1>(W.W1f + (size_t)(2 * wave) * NKG1 * 64, l0);
    w1b = rfrags_load<NKG1>(W.W1f + (size_t)(2 * wave + 1) * NKG1 * 64, l0);
    w2a = rfrags_load<32>(W.W2f + (size_t)(2 * wave) * 32 * 64, l0);
    w2b = rfrags_load<32>(W.W2f + (size_t)(2 * wave + 1) * 32 * 64, l0);
    w3s = rfrags_load<8>(W.W3f + (size_t)(wave * 8) * 64, l0);
  }
  // S8: pieces 0 and 1 of this wave's 32 columns of W2, all sixteen k steps: resident for the launch
  u32x4 W2s[S8 ? 16 : 1][2];
  if constexpr (S8) {
    const u32x4* bx = W2x + (size_t)wave * (FH / 16) * 192 + (tid0 & 63);
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      W2s[ks][0] = bx[(ks * 3 + 0) * 64];
      W2s[ks][1] = bx[(ks * 3 + 1) * 64];
    }
  }
  // S8: the first kP2 streamed pieces of W2 are requested one step AHEAD, behind the layer-2 GEMM of the previous step: their L2
  // latency runs under the head / sampling / env phases instead of in front of layer 2's second MFMA.  (The same for this wave's
  // block of the W1 pack -- 12 DP / 16 registers held through the step -- spilled at 32 and 48 observation columns.)
  constexpr int NKS1s = S8 ? DP / 16 : 1;
  constexpr int kP2 = 3;
  u32x4 P2[kP2 + 1];
  // ... and (synthetic source only: the goal env's phases have no registers to spare) the three pieces of W1's FIRST k step, so that
  // layer 1's first six MFMAs do not wait for the weight fragments requested at its start
  constexpr bool kW1First = S8 && KIND == 1 && MOBROB_S8_W1_FIRST;
  X3Frag P10;
  auto s8_prefetch = [&]() {
    const u32x4* b2x = W2x + (size_t)wave * (FH / 16) * 192 + (tid0 & 63);
#pragma unroll
    for (int k = 0; k < kP2; ++k) P2[k] = b2x[(k * 3 + 2) * 64];
    if constexpr (kW1First) {
      const u32x4* b1x = W1x + (size_t)wave * NKS1s * 192 + (tid0 & 63);
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) P10.p[pc] = b1x[pc * 64];
    }
  };
  // (not for the goal environment at 64 observation columns: twelve more registers through the env phase spilled there)
  constexpr bool kP2Ahead = S8 && !(KIND == 2 && DP == 64);
  if constexpr (kP2Ahead) s8_prefetch();
  // Kernel arguments inside the step loop are read THROUGH the kernarg segment where they are used (`ar`: the same struct behind a
  // constant-address-space reference, re-materialised per step so that nothing is hoisted): by value they are ~60 pointers and
  // ~120 scalars that the compiler loads once and then has to keep through the loop -- 147 .. 220 of them parked in lanes of vector
  // registers, read back with v_readlane on the step's critical path (__graft_entry__.build() fails above SGPR_PARK_LIMIT = 84).
  const int t_begin = a.t0, t_end = a.t1;
  for (int t = t_begin; t < t_end; ++t) {
    RolloutArgsK ap_ = rollout_kernargs();   // `ar` below: (*ap_), re-materialised behind every barrier (a phase keeps only what it uses)
    const int tid = opaque(tid0), lane = tid & 63;
    const int r = lane & 31, h = lane >> 5;
    if constexpr (S8) {   // ---- hidden layers on all eight waves ----
      {  // layer 1: A = the float32 observation tile, split in the k loop (K <= 64: four k steps); B = this wave's block of the W1 pack
        constexpr int NKS1 = DP / 16;
        f32x16 c0 = splat16(ar.pi.b1s[32 * wave + r]);
        const u32x4* b1x = W1x + (size_t)wave * NKS1 * 192 + lane;
        X3Frag P[NKS1s];
#pragma unroll
        for (int ks = kW1First ? 1 : 0; ks < NKS1; ++ks)
#pragma unroll
          for (int pc = 0; pc < 3; ++pc) P[ks].p[pc] = b1x[(ks * 3 + pc) * 64];
        if constexpr (kW1First) P[0] = P10;
        const int ab = 4 * opaque((LB::X + r * ldx + 8 * h) >> 2);
        if (ROLL_ON(2)) {
#pragma unroll
          for (int ks = 0; ks < NKS1; ++ks) {
            X3Frag U;
            if constexpr (MOBROB_S8_XPLANES) {   // row r, columns 16 ks + 8 h .. + 7 of plane pc: float offset XP + (32 pc + r) XLD / 2 + 8 ks + 4 h
              const int axp = 4 * opaque((LB::XP + r * (LB::XLD / 2) + 4 * h) >> 2);
#pragma unroll
              for (int pc = 0; pc < 3; ++pc) U.p[pc] = *reinterpret_cast<const u32x4*>(&lds[axp + pc * 32 * (LB::XLD / 2) + 8 * ks]);
            } else {
              U = x3_split8(*reinterpret_cast<const f32x4*>(&lds[ab + 16 * ks]), *reinterpret_cast<const f32x4*>(&lds[ab + 16 * ks + 4]));
            }
            X3_MFMA6(U, P[ks], c0)
          }
        }
        // epilogue: tanh, then the three bf16 pieces of every value into the planes (element (row crc(i) + 4 h, column 32 wave + r))
        unsigned short* pl = reinterpret_cast<unsigned short*>(&lds[LB::H1]) + opaque((4 * h) * kPlaneLd + 32 * wave + r);
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
          unsigned p1, p2, p3;
          x3_split2(ROLL_TANH(c0[i]), ROLL_TANH(c0[i + 1]), p1, p2, p3);
          const int o0 = crc(i) * kPlaneLd, o1 = crc(i + 1) * kPlaneLd;
          pl[o0] = (unsigned short)(p1 & 0xffffu);                     pl[o1] = (unsigned short)(p1 >> 16);
          pl[32 * kPlaneLd + o0] = (unsigned short)(p2 & 0xffffu);     pl[32 * kPlaneLd + o1] = (unsigned short)(p2 >> 16);
          pl[64 * kPlaneLd + o0] = (unsigned short)(p3 & 0xffffu);     pl[64 * kPlaneLd + o1] = (unsigned short)(p3 >> 16);
        }
      }
      LDS_BARRIER(); ap_ = rollout_kernargs();  // (1) after layer 1
      {  // layer 2: A fragments from the planes (no VALU), B pieces 0 / 1 from registers, piece 2 streamed kP2 k steps ahead
        f32x16 c0 = splat16(ar.pi.b2s[32 * wave + r]);
        const u32x4* b2x = W2x + (size_t)wave * (FH / 16) * 192 + lane;
        if constexpr (!kP2Ahead) s8_prefetch();
        // plane p, row r, columns 16 ks + 8 h .. + 7: float offset H1 + (32 p + r) * 132 + 8 ks + 4 h
        const int ap = 4 * opaque((LB::H1 + r * (kPlaneLd / 2) + 4 * h) >> 2);
        auto frag = [&](int ks) {
          X3Frag f;
#pragma unroll
          for (int pc = 0; pc < 3; ++pc) f.p[pc] = *reinterpret_cast<const u32x4*>(&lds[ap + pc * 32 * (kPlaneLd / 2) + 8 * ks]);
          return f;
        };
        X3Frag U = frag(0);
        // The two waves of a SIMD run the same k loop: left alone they reach their LDS read bursts and their MFMAs together.  The
        // second-dispatched half starts the loop 128 cycles late (MI355X_MICROARCH.md, two waves per SIMD, item 9): rollout 10.37 ->
        // 10.15 ms on one box, two alternations (s_sleep 4: the same; static priority for that half instead: 10.68).  Timing only.
        if (MOBROB_S8_STAGGER > 0 && wave >= 4) __builtin_amdgcn_s_sleep(MOBROB_S8_STAGGER);
        if (ROLL_ON(4)) {
#pragma unroll
          for (int ks = 0; ks < FH / 16; ++ks) {
            X3Frag Un = U;
            if (ks + 1 < FH / 16) Un = frag(ks + 1);
            if (ks + kP2 < FH / 16) P2[(ks + kP2) % (kP2 + 1)] = b2x[((ks + kP2) * 3 + 2) * 64];
            X3Frag Bf;
            Bf.p[0] = W2s[ks][0]; Bf.p[1] = W2s[ks][1]; Bf.p[2] = P2[ks % (kP2 + 1)];
            X3_MFMA6(U, Bf, c0)
            U = Un;
          }
        }
        if (kP2Ahead && t + 1 < t_end) s8_prefetch();   // the next step's weight fragments (the ring slots P2[0 .. kP2) are free again: 16 % 4 == 0)
        const int o = opaque(LB::H2 + 4 * h * FLDH + 32 * wave + r);
#pragma unroll
        for (int i = 0; i < 16; ++i) lds[o + crc(i) * FLDH] = ROLL_TANH(c0[i]);
      }
      LDS_BARRIER(); ap_ = rollout_kernargs();  // (2) after layer 2
    }
    // the random numbers of step ts into buffer ts & 1
    auto draw_sampling = [&](int ts, int hid) {  // standard normals of the sampling stage (consumed after the head)
      if (!ROLL_ON(1)) return;
      const int ngrp = (A + 3) >> 2;
      const int zb = L::ZN + (ts & 1) * (32 * 32);
      for (int i = hid; i < R * ngrp; i += FTHREADS) {
        const int rr_ = i / ngrp, gq = i - rr_ * ngrp;
        float z[4];
        box_muller4(philox4x32_10((uint32_t)(row0 + rr_), (uint32_t)gq, (uint32_t)ts + dbase, 0x45505331u, (uint32_t)ar.seed,
                                  (uint32_t)(ar.seed >> 32)), z);
#pragma unroll
        for (int j = 0; j < 4; ++j) lds[zb + rr_ * 32 + 4 * gq + j] = z[j];
      }
    };
    auto draw_env = [&](int ts, int hid) {  // observation noise of the env phase
      if (!ROLL_ON(16) || !ROLL_ON(256)) return;   // (256: the draw alone, for timing)
      const uint32_t step_ = sbase + (uint32_t)ts;
      const int eb = L::EN + (ts & 1) * (32 * DP);
      for (int i = hid; i < R * per; i += FTHREADS) {
        const int rr_ = i / per, c = i - rr_ * per;
        if (row0 + rr_ < N) {
          float z[4];
          box_muller4(philox4x32_10((uint32_t)(row0 + rr_), (uint32_t)c, step_, kStreamEnvObs, EK0, EK1), z);
          *reinterpret_cast<f32x4*>(&lds[eb + rr_ * DP + 4 * c]) = f32x4{z[0], z[1], z[2], z[3]};
        }
      }
    };
    // KIND 3: what the host wrote for this tile's 32 rows -> LDS, by all eight waves with whole-wave contiguous 16-byte loads (the
    // tile's rows are one contiguous, 16-byte aligned range of the [N][D] host array: 32 D floats; per-row gathers of single floats
    // from uncached host memory cost a PCIe read each and made the step 25x slower).  Staging: the env-noise buffers this kind never
    // draws into -- observations at EN, terminal observations (only when the host reports a truncated row in the range) behind them;
    // rewards / done / truncated flags in the log-prob term buffer, free between barrier (4b) and the next sampling stage.
    auto host_pull = [&](int th) {
      if constexpr (KIND == 3) {
        const int nrow = min(R, N - row0);
        const int nq = nrow * D / 4, rem0 = 4 * nq, nfl = nrow * D;      // whole float4s, then a tail of single floats (nrow < 32 only)
        const f32x4* so = reinterpret_cast<const f32x4*>(ar.h_obs + (size_t)row0 * D);
        for (int i = th; i < nq; i += kRolloutThreads) {
          f32x4 v;
          asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(so + i) : "memory");
          *reinterpret_cast<f32x4*>(&lds[L::EN + 4 * i]) = v;
        }
        for (int i = rem0 + th; i < nfl; i += kRolloutThreads)
          lds[L::EN + i] = __hip_atomic_load(ar.h_obs + (size_t)row0 * D + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (cnt[2] != 0) {
          const f32x4* st = reinterpret_cast<const f32x4*>(ar.h_term + (size_t)row0 * D);
          for (int i = th; i < nq; i += kRolloutThreads) {
            f32x4 v;
            asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(st + i) : "memory");
            *reinterpret_cast<f32x4*>(&lds[L::EN + 32 * DP + 4 * i]) = v;
          }
          for (int i = rem0 + th; i < nfl; i += kRolloutThreads)
            lds[L::EN + 32 * DP + i] = __hip_atomic_load(ar.h_term + (size_t)row0 * D + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (th < nrow) {
          lds[L::TM + th] = __hip_atomic_load(ar.h_rew + row0 + th, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          reinterpret_cast<int*>(&lds[L::TM])[32 + th] = __hip_atomic_load(ar.h_done + row0 + th, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          reinterpret_cast<int*>(&lds[L::TM])[64 + th] = cnt[2] != 0 ? (int)__hip_atomic_load(ar.h_trunc + row0 + th, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0;
        }
      }
    };
    if (kRolloutStationary) {  // weights-stationary plan: the four waves draw their own numbers, first
      draw_sampling(t, tid);
      draw_env(t, tid);
    } else if (noise_wave) {
      // (every barrier of the step loop is LDS-only: the step's global stores -- observations, actions, log-probs, rewards -- are read by
      //  later kernels, and __syncthreads() made each of the three barriers behind them wait for their acknowledgement: 1.5 us per step)
      // Step t's numbers were drawn during step t - 1 (or in front of the loop); the noise waves keep the barriers' count and
      // draw step t + 1's numbers into the other buffer halves while the policy waves sample, step the env and update the state:
      // the half being written was last read in step t - 1, whose phases all ended before barrier (6) of that step.
      const int hid = tid - FTHREADS;
      if (t == t_begin) { draw_sampling(t, hid); if constexpr (KIND != 3) draw_env(t, hid); }
      if constexpr (!S8) {
        LDS_BARRIER(); ap_ = rollout_kernargs();  // (1) after layer 1
        LDS_BARRIER(); ap_ = rollout_kernargs();  // (2) after layer 2
      }
      LDS_BARRIER(); ap_ = rollout_kernargs();  // (3) after the head
      if (t + 1 < t_end) draw_sampling(t + 1, hid);
      LDS_BARRIER(); ap_ = rollout_kernargs();  // (4) after the sampling stage
      if constexpr (KIND == 3) {
        LDS_BARRIER(); ap_ = rollout_kernargs();  // (4b) the host has stepped the rows (or the wait was given up)
        if (cnt[1]) break;
        host_pull(tid);
        LDS_BARRIER(); ap_ = rollout_kernargs();  // (4c) the host's tile is in LDS
      } else {
        if (t + 1 < t_end) draw_env(t + 1, hid);
      }
#if MOBROB_ROLLOUT_BARRIER5
      LDS_BARRIER(); ap_ = rollout_kernargs();  // (5) after the env phase
#endif
      LDS_BARRIER(); ap_ = rollout_kernargs();  // (6) after the state update
    }
    if (!noise_wave) {
    if constexpr (!S8) {
    {  // layer 1
      f32x16 c0 = splat16(ar.pi.b1s[64 * wave + r]), c1 = splat16(ar.pi.b1s[64 * wave + 32 + r]);
      constexpr int nkg = DP / 8;
      if (ROLL_ON(2)) {
        if (kRolloutStationary) gemm_two_resident<ldx, NKG1>(LB::X, w1a, w1b, c0, c1, lane);
        else if (x3) gemm_x3_r32<ldx, DP / 16>(LB::X, W1x + (size_t)(2 * wave) * (DP / 16) * 192, W1x + (size_t)(2 * wave + 1) * (DP / 16) * 192, c0, c1, lane);
        else gemm_lds_packed_r32<ldx>(LB::X, ar.pi.W1f + (size_t)(2 * wave) * nkg * 64, ar.pi.W1f + (size_t)(2 * wave + 1) * nkg * 64, nkg,
                                      c0, c1, lane);
      }
      const int o = opaque(LB::H1 + 4 * h * FLDH + 64 * wave + r);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        lds[o + crc(i) * FLDH] = ROLL_TANH(c0[i]);
        lds[o + crc(i) * FLDH + 32] = ROLL_TANH(c1[i]);
      }
    }
    LDS_BARRIER(); ap_ = rollout_kernargs();
    {  // layer 2
      f32x16 c0 = splat16(ar.pi.b2s[64 * wave + r]), c1 = splat16(ar.pi.b2s[64 * wave + 32 + r]);
      constexpr int nkg = FH / 8;
      if (ROLL_ON(4)) {
        if (kRolloutStationary) gemm_two_resident<FLDH, 32>(LB::H1, w2a, w2b, c0, c1, lane);
        else if (x3) gemm_x3_r32<FLDH, FH / 16>(LB::H1, W2x + (size_t)(2 * wave) * (FH / 16) * 192, W2x + (size_t)(2 * wave + 1) * (FH / 16) * 192, c0, c1, lane);
        else gemm_lds_packed_r32_deep<FLDH>(LB::H1, ar.pi.W2f + (size_t)(2 * wave) * nkg * 64,
                                            ar.pi.W2f + (size_t)(2 * wave + 1) * nkg * 64, nkg, c0, c1, lane);
      }
      const int o = opaque(LB::H2 + 4 * h * FLDH + 64 * wave + r);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        lds[o + crc(i) * FLDH] = ROLL_TANH(c0[i]);
        lds[o + crc(i) * FLDH + 32] = ROLL_TANH(c1[i]);
      }
    }
    LDS_BARRIER(); ap_ = rollout_kernargs();
    }  // (!S8)
    {  // head: K split over the 4 waves (64 each); partial tiles side by side, summed in the sampling stage
      f32x16 acc = zero16(), acc2 = zero16();
      const int ab = 4 * opaque((LB::H2 + r * FLDH + wave * 64 + 4 * h) >> 2);
      const f32x4* bp = ar.pi.W3f + (size_t)(wave * 8) * 64;
      const unsigned bo = opaque_u((unsigned)lane * 16u);
      // fragments requested four at a time (they used to be fetched pair by pair inside the loop: four L2 round trips in a row;
      // all eight at once spilled the goal-env variant at 64 observation columns)
      f32x4 hb[4];
#pragma unroll
      for (int kg = 0; kg < (ROLL_ON(64) ? 8 : 0); kg += 2) {
        if ((kg & 3) == 0) {
#pragma unroll
          for (int q = 0; q < 4; ++q) hb[q] = kRolloutStationary ? w3s.f[kg + q] : ldg16(bp, bo + (kg + q) * 1024u);
        }
        const f32x4 b0 = hb[kg & 3];
        const f32x4 b1 = hb[(kg & 3) + 1];
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(&lds[ab + kg * 8]);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(&lds[ab + kg * 8 + 8]);
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
          acc = MFMA32(a0[s_], b0[s_], acc);
          acc2 = MFMA32(a1[s_], b1[s_], acc2);
        }
      }
      const int o = opaque(LB::DO + wave * R * FLDO + 4 * h * FLDO + r);
#pragma unroll
      for (int i = 0; i < 16; ++i) lds[o + crc(i) * FLDO] = acc[i] + acc2[i];
    }
    if (tid == 0) *cnt = 0;
    LDS_BARRIER(); ap_ = rollout_kernargs();
    // ---- Gaussian sample + log-prob.  Same expressions and Philox counters as k_fused_act, spread over the block:
    //      (row, action group) items draw the normals, (row, action) items form action and log-prob term, one lane
    //      per row adds the terms in action order (-> bit-identical log-probs) ----
    for (int i = tid; i < (ROLL_ON(8) ? R * A : 0); i += FTHREADS) {
      const int rr_ = i / A, k = i - rr_ * A;
      const int row = row0 + rr_;
      if (row < N) {
        const int db = LB::DO + rr_ * FLDO + k;
        const float m = ((lds[db] + lds[db + R * FLDO]) + (lds[db + 2 * R * FLDO] + lds[db + 3 * R * FLDO])) + lds[L::AC + 96 + k];
        const float sd = lds[L::AC + k];
        const float act = m + lds[L::ZN + (t & 1) * (32 * 32) + rr_ * 32 + k] * sd;
        const float d = act - m;
        lds[L::TM + rr_ * 33 + k] = -(d * d) / lds[L::AC + 32 + k] - lds[L::AC + 64 + k] - 0.91893853320467274178f;
        const float ac = fminf(fmaxf(act, ar.lo), ar.hi);
        if (ROLL_ON(128)) ar.actions[((size_t)t * N + row) * A + k] = act;
        if constexpr (KIND == 3) __hip_atomic_store(&ar.clip_act[(size_t)row * A + k], ac, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // pinned host memory, past the caches
        else if (ROLL_ON(128)) ar.clip_act[(size_t)row * A + k] = ac;
        lds[L::CA + rr_ * 33 + k] = ac;
      }
    }
    SERVE_STAMP(6)
    if constexpr (KIND == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave: its actions have left for host memory
    SERVE_STAMP(7)
    LDS_BARRIER(); ap_ = rollout_kernargs();
    if constexpr (KIND == 3) {
      // Hand-over: one lane tells the host "the actions of step t of these 32 rows are in your memory" and then waits until the host
      // has stepped the row range (part) these rows belong to.  Relaxed system-scope accesses past the caches (sc0 sc1), ordered by the
      // drained stores above and by the barriers around this block; bounded: a host that never answers raises the error word and this
      // launch (and, through abort_dev, the launches queued behind it) gives up -- the HOST then fails the rollout.
      SERVE_STAMP(0)   // actions drained, barrier passed
      if (tid == 64) {
        __hip_atomic_store(ar.h_gpu_flag + blockIdx.x, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        // The host's word of the row range: (truncated rows of the newest step) << 32 | steps finished -- ONE 8-byte load per poll.
        // Only the range's FIRST workgroup polls host memory (a PCIe round trip per poll); it relays what it saw through a device
        // word the other workgroups of the range poll (64 pollers per range on the PCIe link made every round trip ~4 us).
        const int part = row0 / ar.rows_per_part;
        const bool relay = row0 == part * ar.rows_per_part;
        const unsigned long long* hf = reinterpret_cast<const unsigned long long*>(ar.h_host_flag + 16 * part);
        unsigned long long* df = reinterpret_cast<unsigned long long*>(ar.abort_dev) + 1 + part;
        const long long w0 = wall_clock64();
        unsigned long long v;
        for (;;) {
          v = relay ? __hip_atomic_load(hf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : __hip_atomic_load(df, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((unsigned)v >= (unsigned)(t + 1)) break;
          if (wall_clock64() - w0 > ar.timeout_ticks) {
            v = 0xFFFFFFFFull;
            __hip_atomic_store(ar.h_error, t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
          }
          if (relay) __builtin_amdgcn_s_sleep(8); else __builtin_amdgcn_s_sleep(2);
        }
        if (relay) __hip_atomic_store(df, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // ONE system-scope acquire between the word's arrival and host_pull (buffer_inv sc0 sc1: this CU's L1 and the XCD L2's
        // non-coherent lines), completed (vmcnt) before barrier (4b) lets the other waves load: what the host wrote before it
        // raised its word is what every load behind the barrier sees, whatever the caching policy of the block.  (The polls
        // themselves stay relaxed: an acquire per poll is an invalidate per PCIe round trip.)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const bool dead = (unsigned)v == 0xFFFFFFFFu;
        if (dead) __hip_atomic_store(ar.abort_dev, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        SERVE_STAMP(1)   // the host's word arrived
        cnt[1] = dead ? 1 : 0;
        cnt[2] = dead ? 0 : (int)(v >> 32);
      }
    }
    if (tid < R && row0 + tid < N) {
      float lp = 0.f;
      for (int k = 0; k < A; ++k) lp += lds[L::TM + tid * 33 + k];
      if (ROLL_ON(128)) ar.logp[(size_t)t * N + row0 + tid] = lp;
    }
    if constexpr (KIND == 3) {
      LDS_BARRIER(); ap_ = rollout_kernargs();  // (4b)
      if (cnt[1]) break;
      SERVE_STAMP(2)
      host_pull(tid);
      SERVE_STAMP(3)   // this wave's share of the tile is in LDS
      LDS_BARRIER(); ap_ = rollout_kernargs();  // (4c)
      SERVE_STAMP(4)
    }
    // ---- env.step(clipped actions) + auto-reset: 8 threads per row, observation chunks sub and sub + 8 ----
    const int rr = tid >> 3, sub = tid & 7;
    const int n = row0 + rr;
    const bool live = n < N;
    const uint32_t step = sbase + (uint32_t)t;
    float* S = &lds[L::ST + rr * 16];
    float* xrow = &lds[LB::X + rr * ldx];
    float* trow = &lds[LB::H2 + rr * DP];  // terminal observation staging (h2 is dead after the head GEMM)
    const size_t onext = ((size_t)(t + 1) * N + (live ? n : 0)) * per;
    bool tr = false, done = false, reached = false;
    float reward = 0.f, ep_ret = 0.f;
    int ep_len_new = 0, ep_len_fin = 0;
    GoalState g{};
    if constexpr (KIND == 3) {
      if (live) {   // what the host wrote for this row (staged in LDS by host_pull): reward, done, truncated, next observation, terminal one
        done = reinterpret_cast<const int*>(&lds[L::TM])[32 + rr] != 0;
        tr = reinterpret_cast<const int*>(&lds[L::TM])[64 + rr] != 0;
        if (sub == 0) reward = lds[L::TM + rr];
        auto staged_chunk = [&](int base, int c) {   // columns 4 c .. 4 c + 3 of the tile's row rr ([32][D] floats at `base`)
          f32x4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int col = 4 * c + j;
            o[j] = col < D ? lds[base + rr * D + (col < D ? col : 0)] : 0.f;
          }
          return o;
        };
        for (int c = sub; c < per; c += 8) {
          const f32x4 o = staged_chunk(L::EN, c);
          if (tr) {
            const f32x4 to = staged_chunk(L::EN + 32 * DP, c);
            reinterpret_cast<f32x4*>(ar.term_obs)[(size_t)n * per + c] = to;
            *reinterpret_cast<f32x4*>(&trow[4 * c]) = to;
          }
          reinterpret_cast<f32x4*>(ar.obs)[onext + c] = o;
          if constexpr (S8 && MOBROB_S8_XPLANES) xplanes_store4<LB>(rr, c, o);
          else *reinterpret_cast<f32x4*>(&xrow[4 * c]) = o;
        }
      }
    } else if (live && ROLL_ON(16)) {
      if (KIND == 1) {
        const Philox4 mr = philox4x32_10((uint32_t)n, 0u, step, kStreamEnvMisc, EK0, EK1);
        const bool term = u32_to_unit_open(mr.x) < ar.p_term;
        const int len = reinterpret_cast<const int*>(S)[13] + 1;
        tr = (len >= ar.time_limit) && !term;
        done = term || tr;
        ep_len_new = done ? 0 : len;
        for (int c = sub; c < per; c += 8) {
          f32x4 z = *reinterpret_cast<const f32x4*>(&lds[L::EN + (t & 1) * (32 * DP) + rr * DP + 4 * c]);  // drawn by the noise waves
          f32x4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = (4 * c + j < D) ? z[j] : 0.f;
          if (tr) {
            reinterpret_cast<f32x4*>(ar.term_obs)[(size_t)n * per + c] = o;
            *reinterpret_cast<f32x4*>(&trow[4 * c]) = o;
            float zt[4];
            box_muller4(philox4x32_10((uint32_t)n, (uint32_t)c, step, kStreamEnvTerm, EK0, EK1), zt);
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (4 * c + j < D) ? zt[j] : 0.f;
          }
          if (ROLL_ON(128)) reinterpret_cast<f32x4*>(ar.obs)[onext + c] = o;
          if constexpr (S8 && MOBROB_S8_XPLANES) xplanes_store4<LB>(rr, c, o);
          else *reinterpret_cast<f32x4*>(&xrow[4 * c]) = o;
        }
        if (sub == 0) {
          float zz[4];
          box_muller4(Philox4{mr.y, mr.z, mr.w, mr.x ^ 0x9E3779B9u}, zz);
          reward = 0.03f + 0.1f * zz[0] + (term ? 5.0f : 0.f);
        }
      } else {
        g = goal_load(S);
        const GoalOutcome o = goal_advance(g, ar.goal, &lds[L::CA + rr * 33], A);
        tr = o.tr; done = o.done; reached = o.reached; reward = o.reward;
        ep_ret = g.ep_ret; ep_len_fin = g.ep_len;
        GoalState gn = g;
        if (done) goal_reset(gn, ar.goal, o.reached, (uint32_t)n, step, EK0, EK1);
        for (int c = sub; c < per; c += 8) {
          float z[4];
          {
            const f32x4 zz = *reinterpret_cast<const f32x4*>(&lds[L::EN + (t & 1) * (32 * DP) + rr * DP + 4 * c]);  // drawn by the noise waves
            z[0] = zz[0]; z[1] = zz[1]; z[2] = zz[2]; z[3] = zz[3];
          }
          f32x4 ob = goal_features(g, ar.goal.P, D, c, z, ar.goal.noise);
          if (done) {
            if (tr) {
              reinterpret_cast<f32x4*>(ar.term_obs)[(size_t)n * per + c] = ob;
              *reinterpret_cast<f32x4*>(&trow[4 * c]) = ob;
            }
            box_muller4(philox4x32_10((uint32_t)n, (uint32_t)c, step, kStreamEnvTerm, EK0, EK1), z);
            ob = goal_features(gn, ar.goal.P, D, c, z, ar.goal.noise);
          }
          if (ROLL_ON(128)) reinterpret_cast<f32x4*>(ar.obs)[onext + c] = ob;
          if constexpr (S8 && MOBROB_S8_XPLANES) xplanes_store4<LB>(rr, c, ob);
          else *reinterpret_cast<f32x4*>(&xrow[4 * c]) = ob;
        }
        g = gn;
      }
    }
    // The eight threads of a row (tid >> 3) sit in ONE wave: their reads of the row's old state above are issued before lane sub == 0's
    // writes below and a wave's LDS operations execute in order -- the workgroup barrier that used to stand here ("every thread of a
    // row has read the row's old state") ordered nothing else (MOBROB_ROLLOUT_BARRIER5=1 brings it back; bit-identical either way).
#if MOBROB_ROLLOUT_BARRIER5
    LDS_BARRIER();
#endif
    ap_ = rollout_kernargs();
    if (live && sub == 0) {
      const size_t so = (size_t)t * N + n;
      if (ROLL_ON(128)) ar.es[so] = S[12];
      S[12] = done ? 1.f : 0.f;
      if (ROLL_ON(128)) ar.trunc[n] = tr ? 1 : 0;
      if (KIND == 1) {
        reinterpret_cast<int*>(S)[13] = ep_len_new;
      } else if (KIND == 2) {
        goal_store(S, g);
        if (done) {  // Monitor statistics: accumulated per thread, flushed once per launch (see the kernel end)
          es_n += 1.0; es_ret += (double)ep_ret; es_len += (double)ep_len_fin; es_goal += reached ? 1.0 : 0.0;
          ep_ring_push(ar.ep_stats, ep_ret, (float)ep_len_fin);
        }
      }
      if (tr) {  // reward is written after the bootstrap below
        const int q = atomicAdd(cnt, 1);
        lrow[q] = rr;
        lrew[q] = reward;
      } else {
        if (ROLL_ON(128)) ar.rewards[so] = reward;
      }
    }
    LDS_BARRIER(); ap_ = rollout_kernargs();
    SERVE_STAMP(5)   // step over: state updated
    }  // (policy waves)
    // ---- time-limit bootstrap of the (rare) truncated rows: r += gamma * V(terminal_obs).  All eight waves (the value MLP
    //      of a row is a block-wide routine with its own barriers; the noise waves hold no hidden unit and add zeros) ----
    const int m = *cnt;
    for (int q = 0; q < m; ++q) {
      const int br = lrow[q];
      float* sc = &lds[LB::H1];  // h1 is dead after the layer-2 GEMM: scratch h1[G1] | h2[G2] | red[16]
      const float v = value_row_lds(&lds[LB::H2 + br * DP], sc, sc + ar.bt.G1, sc + ar.bt.G1 + ar.bt.G2, ar.bt.W1, ar.bt.b1,
                                    ar.bt.W2, ar.bt.b2, ar.bt.Wv, ar.bt.bv, D, ar.bt.G1, ar.bt.G2);
      if (tid == 0) {
        ar.bt.term_val[row0 + br] = v;
        ar.rewards[(size_t)t * N + row0 + br] = (float)((double)lrew[q] + (double)__fmul_rn(ar.bt.gamma, v));
      }
      __syncthreads();
    }
  }
  // ---- episode statistics: one set of atomics per wave and launch (order irrelevant: diagnostics) ----
  if (KIND == 2) {
    const double n = wave_sum_d(es_n), r = wave_sum_d(es_ret), l = wave_sum_d(es_len), g = wave_sum_d(es_goal);
    if ((tid0 & 63) == 0 && n > 0.0) {
      atomicAdd(&a.ep_stats[0], n); atomicAdd(&a.ep_stats[1], r); atomicAdd(&a.ep_stats[2], l); atomicAdd(&a.ep_stats[3], g);
    }
  }
  // ---- carried state back to global ----
  if (tid0 < R) {
    const int n = row0 + tid0;
    const float* S = &lds[L::ST + tid0 * 16];
    if (n < N) {
      if (KIND == 2) {
#pragma unroll
        for (int j = 0; j < kGoalStateFloats; ++j) a.gstate[(size_t)n * kGoalStateFloats + j] = S[j];
      } else if (KIND == 1) {
        a.ep_len[n] = reinterpret_cast<const int*>(S)[13];
      }
      a.prev_dones[n] = S[12];
    }
  }
}

#undef ar
#undef EK0
#undef EK1

// ------------------------------------------------------------------------------------------------
// Batched value forward: v[row] = V(X[row]) for `rows` contiguous observations.  Persistent 256-thread blocks loop
// over 64-row tiles with the forward code of the training kernel (tile_layers + the 16x16x4 head); the next tile's
// rows are in flight while the current tile computes.
// ------------------------------------------------------------------------------------------------
template <int DP>
__global__ __launch_bounds__(FTHREADS, 1) void k_value_batch(FusedNet W, const float* __restrict__ X, int rows,
                                                             float* __restrict__ v) {
  using L = Lay<DP>;
  constexpr int ldx = L::LDX, per = DP / 4;
  constexpr int NG = (FR * per + FTHREADS - 1) / FTHREADS;
  const int tid0 = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int ntiles = (rows + FR - 1) / FR;
  f32x4 xr[NG];
#pragma unroll
  for (int u = 0; u < NG; ++u) {
    const int i = tid0 + u * FTHREADS, rr = i / per, c = i - rr * per;
    xr[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    if ((int)blockIdx.x < ntiles && blockIdx.x * FR + rr < rows)
      xr[u] = ldg16(X, (unsigned)(blockIdx.x * FR + rr) * (unsigned)(DP * 4) + (unsigned)(c * 16));
  }
  const float bv = W.b3[0];
#ifdef MOBROB_STAMPS  // diagnostic build: the shared forward code stamps into a scratch array here
  unsigned long long tacc_[26] = {0};
  unsigned long long tprev_ = 0;
#endif
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int tid = opaque(tid0), lane = tid & 63;
    const Frag2 f1 = prefetch_frag(W.W1f + (size_t)(2 * wave) * (DP / 8) * 64,
                                   W.W1f + (size_t)(2 * wave + 1) * (DP / 8) * 64, lane);
#pragma unroll
    for (int u = 0; u < NG; ++u) {
      const int i = tid + u * FTHREADS, rr = i / per, c = i - rr * per;
      *reinterpret_cast<f32x4*>(&lds[L::X + rr * ldx + 4 * c]) = xr[u];
    }
    const int nt = tile + gridDim.x;
#pragma unroll
    for (int u = 0; u < NG; ++u) {
      const int i = tid + u * FTHREADS, rr = i / per, c = i - rr * per;
      xr[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (nt < ntiles && nt * FR + rr < rows)
        xr[u] = ldg16(X, (unsigned)(nt * FR + rr) * (unsigned)(DP * 4) + (unsigned)(c * 16));
    }
    __syncthreads();
    const Frag2 f3 = W.W2x != nullptr ? tile_layers_x3<DP>(W, wave, lane) : tile_layers<DP, true>(W, wave, lane, f1 STAMP_ARGS);
    tile_head16<DP>(W, wave, lane, f3);  // wave w writes head rows 16w..16w+15, read back by the same wave below
    if (lane < 16) {
      const int rr = 16 * wave + lane, row = tile * FR + rr;
      if (row < rows) v[row] = lds[L::DO + rr * FLDO] + bv;
    }
    __syncthreads();  // X / h1 / h2 / head tile are rewritten by the next tile
  }
}

// ------------------------------------------------------------------------------------------------
// Hidden width 64 (reference YAML shape): the same persistent rollout with ONE WAVE per 32-env tile and the
// policy's forward packs resident in LDS for the whole launch (no weight traffic, no workgroup barrier in the step
// loop: every phase of a tile is wave-synchronous, ordered by the wave's in-order DS queue).  A step is ~100 MFMAs
// plus sampling and the env rules -- a few microseconds; the per-step launches of the graph path cost more than that.
// ------------------------------------------------------------------------------------------------
template <int DP>
struct WtsF64 {  // forward packs + scaled biases as mirrored into LDS (a prefix and a suffix of Wts64<DP>)
  static constexpr int W1F = 0;
  static constexpr int W2F = W1F + 2 * (DP / 8) * 256;
  static constexpr int W3F = W2F + 2 * 8 * 256;
  static constexpr int B1S = W3F + 8 * 256;
  static constexpr int B2S = B1S + 64;
  static constexpr int TOTAL = B2S + 64;
};
template <int DP>
struct LayRo64 {
  using B = Lay64<DP>;
  static constexpr int CA = B::WAVE;          // [32][33] clipped actions
  static constexpr int ST = CA + 32 * 33;     // [32][16] row state
  static constexpr int ZN = ST + 32 * 16;     // [32][32] standard normals
  static constexpr int TM = ZN + 32 * 32;     // [32][33] log-prob terms
  static constexpr int WAVE = TM + 32 * 33;   // floats per wave
  static constexpr int AC = WtsF64<DP>::TOTAL;            // block: per-action constants [4][32]
  static constexpr int W0 = AC + 128;                     // first wave region
  static constexpr int NWV = (40960 - W0) / WAVE >= 4 ? 4 : (40960 - W0) / WAVE;  // waves per block (<= 160 KB)
  static constexpr int END = W0 + NWV * WAVE;
};
inline int rollout64_waves(int Dp) {
  const int w0 = 2 * (Dp / 8) * 256 + 4096 + 2048 + 128 + 128;
  const int wave = 32 * (Dp + 4) + 2 * 32 * GLDH + 32 * FLDO + 64 + 32 * 33 + 32 * 16 + 32 * 32 + 32 * 33;
  return std::min(4, (40960 - w0) / wave);
}
inline size_t rollout64_lds_bytes(int Dp) {
  const int w0 = 2 * (Dp / 8) * 256 + 4096 + 2048 + 128 + 128;
  const int wave = 32 * (Dp + 4) + 2 * 32 * GLDH + 32 * FLDO + 64 + 32 * 33 + 32 * 16 + 32 * 32 + 32 * 33;
  return (size_t)(w0 + rollout64_waves(Dp) * wave) * sizeof(float);
}

// V(x) for one row by one wave (x[D], h1[G1], h2[G2] in the wave's LDS).  Same per-unit fma chains as value_row_lds;
// for G2 <= 64 the final wave_sum equals its block_sum (the other waves contribute exact zeros).
template <class BT>   // BootArgs, or the same struct behind a kernel-argument reference (constant address space)
__device__ __forceinline__ float value_row_wave(const float* x, float* h1, float* h2, const BT& bt, int D, int lane) {
  for (int j = lane; j < bt.G1; j += 64) {
    float s = 0.f;
    for (int k = 0; k < D; ++k) s = fmaf(x[k], bt.W1[(size_t)j * D + k], s);
    h1[j] = tanhf(s + bt.b1[j]);
  }
  for (int j = lane; j < bt.G2; j += 64) {
    float s = 0.f;
    for (int k = 0; k < bt.G1; ++k) s = fmaf(h1[k], bt.W2[(size_t)j * bt.G1 + k], s);
    h2[j] = tanhf(s + bt.b2[j]);
  }
  float p = 0.f;
  for (int k = lane; k < bt.G2; k += 64) p += h2[k] * bt.Wv[k];
  return wave_sum(p) + bt.bv[0];
}

#define ar (*ap_)
#define EK0 ((uint32_t)ar.env_seed)
#define EK1 ((uint32_t)(ar.env_seed >> 32))
template <int DP>
__global__ __launch_bounds__(LayRo64<DP>::NWV * 64, 1) void k_rollout64_persistent(RolloutArgs a) {
  using L = LayRo64<DP>;
  using LB = Lay64<DP>;
  using WF = WtsF64<DP>;
  using WS = Wts64<DP>;
  constexpr int ldx = LB::LDX, per = DP / 4, R = 32;
  const int tid0 = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int N = a.N, A = a.A, D = a.D;
  // ---- forward packs of the policy network and the per-action constants -> LDS (whole block) ----
  {
    const f32x4* src = a.pi.W1f;  // W1F | W2F | W3F are contiguous in the per-network pack (Wts64 layout)
    for (int i = tid0; i < WF::B1S / 4; i += blockDim.x) reinterpret_cast<f32x4*>(lds)[i] = src[i];
    for (int i = tid0; i < 64; i += blockDim.x) {
      lds[WF::B1S + i] = a.pi.b1s[i];
      lds[WF::B2S + i] = a.pi.b2s[i];
    }
    if (tid0 < 32) {
      float sd = 1.f, bb = 0.f;
      if (tid0 < A) { sd = expf(a.log_std[tid0]); bb = a.pi.b3[tid0]; }
      lds[L::AC + tid0] = sd;
      lds[L::AC + 32 + tid0] = 2.0f * (sd * sd);
      lds[L::AC + 64 + tid0] = logf(sd);
      lds[L::AC + 96 + tid0] = bb;
    }
  }
  __syncthreads();  // the only workgroup barrier of the kernel
  (void)sizeof(WS);
  const int tile = blockIdx.x * L::NWV + wave;
  const int row0 = tile * R;
  if (row0 >= N) return;
  const int wb = L::W0 + wave * L::WAVE;
  const int lane0 = tid0 & 63;
  // ---- carried state and the observation tile of step t0 ----
  if (lane0 < R) {
    const int n = row0 + lane0;
    float* S = &lds[wb + L::ST + lane0 * 16];
    if (n < N) {
      if (a.kind == 2) {
#pragma unroll
        for (int j = 0; j < kGoalStateFloats; ++j) S[j] = a.gstate[(size_t)n * kGoalStateFloats + j];
      } else {
        reinterpret_cast<int*>(S)[13] = a.ep_len[n];
      }
      S[12] = a.prev_dones[n];
    }
  }
  for (int i = lane0; i < R * per; i += 64) {
    const int rr = i / per, c = i - rr * per;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row0 + rr < N)
      v = ldg16(a.obs, (unsigned)((size_t)a.t0 * N + row0 + rr) * (unsigned)(DP * 4) + (unsigned)(c * 16));
    *reinterpret_cast<f32x4*>(&lds[wb + LB::X + rr * ldx + 4 * c]) = v;
  }
  const uint32_t dbase = a.draw_base ? *a.draw_base : a.draw0, sbase = a.step_base ? *a.step_base : 0u;
  const uint32_t ek0 = (uint32_t)a.env_seed, ek1 = (uint32_t)(a.env_seed >> 32);

  double es_n = 0.0, es_ret = 0.0, es_len = 0.0, es_goal = 0.0;  // finished episodes of this thread's row
  const int t_begin = a.t0, t_end = a.t1;   // (kernel arguments and Philox keys inside the step loop: see k_rollout_persistent)
  for (int t = t_begin; t < t_end; ++t) {
    RolloutArgsK ap_ = rollout_kernargs();
    const int lane = opaque(lane0);
    {  // standard normals of this step
      const int ngrp = (A + 3) >> 2;
      for (int i = lane; i < R * ngrp; i += 64) {
        const int rr_ = i / ngrp, gq = i - rr_ * ngrp;
        float z[4];
        box_muller4(philox4x32_10((uint32_t)(row0 + rr_), (uint32_t)gq, (uint32_t)t + dbase, 0x45505331u, (uint32_t)ar.seed,
                                  (uint32_t)(ar.seed >> 32)), z);
#pragma unroll
        for (int j = 0; j < 4; ++j) lds[wb + L::ZN + rr_ * 32 + 4 * gq + j] = z[j];
      }
    }
    tile64_forward_ldsw<DP, WF>(wb, lane);  // h1, h2, raw head tile (policy) in the wave's region
    // ---- Gaussian sample + log-prob (expressions and Philox counters of k_fused64_act) ----
    for (int i = lane; i < R * A; i += 64) {
      const int rr_ = i / A, k = i - rr_ * A;
      const int row = row0 + rr_;
      if (row < N) {
        const float m = lds[wb + LB::DO + rr_ * FLDO + k] + lds[L::AC + 96 + k];
        const float sd = lds[L::AC + k];
        const float act = m + lds[wb + L::ZN + rr_ * 32 + k] * sd;
        const float d = act - m;
        lds[wb + L::TM + rr_ * 33 + k] = -(d * d) / lds[L::AC + 32 + k] - lds[L::AC + 64 + k] - 0.91893853320467274178f;
        const float ac = fminf(fmaxf(act, ar.lo), ar.hi);
        ar.actions[((size_t)t * N + row) * A + k] = act;
        ar.clip_act[(size_t)row * A + k] = ac;
        lds[wb + L::CA + rr_ * 33 + k] = ac;
      }
    }
    if (lane < R && row0 + lane < N) {
      float lp = 0.f;
      for (int k = 0; k < A; ++k) lp += lds[wb + L::TM + lane * 33 + k];
      ar.logp[(size_t)t * N + row0 + lane] = lp;
    }
    // ---- env.step + auto-reset: 2 lanes per row, observation chunks sub, sub + 2, ... ----
    const int rr = lane >> 1, sub = lane & 1;
    const int n = row0 + rr;
    const bool live = n < N;
    const uint32_t step = sbase + (uint32_t)t;
    float* S = &lds[wb + L::ST + rr * 16];
    float* xrow = &lds[wb + LB::X + rr * ldx];
    float* trow = &lds[wb + LB::H2 + rr * GLDH];  // terminal observation staging (h2 is dead after the head GEMM)
    const size_t onext = ((size_t)(t + 1) * N + (live ? n : 0)) * per;
    bool tr = false, done = false, reached = false;
    float reward = 0.f, ep_ret = 0.f;
    int ep_len_new = 0, ep_len_fin = 0;
    GoalState g{};
    if (live) {
      if (ar.kind == 1) {
        const Philox4 mr = philox4x32_10((uint32_t)n, 0u, step, kStreamEnvMisc, EK0, EK1);
        const bool term = u32_to_unit_open(mr.x) < ar.p_term;
        const int len = reinterpret_cast<const int*>(S)[13] + 1;
        tr = (len >= ar.time_limit) && !term;
        done = term || tr;
        ep_len_new = done ? 0 : len;
        for (int c = sub; c < per; c += 2) {
          float z[4];
          box_muller4(philox4x32_10((uint32_t)n, (uint32_t)c, step, kStreamEnvObs, EK0, EK1), z);
          f32x4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = (4 * c + j < D) ? z[j] : 0.f;
          if (tr) {
            reinterpret_cast<f32x4*>(ar.term_obs)[(size_t)n * per + c] = o;
            *reinterpret_cast<f32x4*>(&trow[4 * c]) = o;
            box_muller4(philox4x32_10((uint32_t)n, (uint32_t)c, step, kStreamEnvTerm, EK0, EK1), z);
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (4 * c + j < D) ? z[j] : 0.f;
          }
          reinterpret_cast<f32x4*>(ar.obs)[onext + c] = o;
          *reinterpret_cast<f32x4*>(&xrow[4 * c]) = o;
        }
        if (sub == 0) {
          float zz[4];
          box_muller4(Philox4{mr.y, mr.z, mr.w, mr.x ^ 0x9E3779B9u}, zz);
          reward = 0.03f + 0.1f * zz[0] + (term ? 5.0f : 0.f);
        }
      } else {
        g = goal_load(S);
        const GoalOutcome o = goal_advance(g, ar.goal, &lds[wb + L::CA + rr * 33], A);
        tr = o.tr; done = o.done; reached = o.reached; reward = o.reward;
        ep_ret = g.ep_ret; ep_len_fin = g.ep_len;
        GoalState gn = g;
        if (done) goal_reset(gn, ar.goal, o.reached, (uint32_t)n, step, EK0, EK1);
        for (int c = sub; c < per; c += 2) {
          float z[4];
          box_muller4(philox4x32_10((uint32_t)n, (uint32_t)c, step, kStreamEnvObs, EK0, EK1), z);
          f32x4 ob = goal_features(g, ar.goal.P, D, c, z, ar.goal.noise);
          if (done) {
            if (tr) {
              reinterpret_cast<f32x4*>(ar.term_obs)[(size_t)n * per + c] = ob;
              *reinterpret_cast<f32x4*>(&trow[4 * c]) = ob;
            }
            box_muller4(philox4x32_10((uint32_t)n, (uint32_t)c, step, kStreamEnvTerm, EK0, EK1), z);
            ob = goal_features(gn, ar.goal.P, D, c, z, ar.goal.noise);
          }
          reinterpret_cast<f32x4*>(ar.obs)[onext + c] = ob;
          *reinterpret_cast<f32x4*>(&xrow[4 * c]) = ob;
        }
        g = gn;
      }
    }
    // every lane of the wave has consumed the old row state (program order within the wave): commit
    const bool boot = live && sub == 0 && tr;
    if (live && sub == 0) {
      const size_t so = (size_t)t * N + n;
      ar.es[so] = S[12];
      S[12] = done ? 1.f : 0.f;
      ar.trunc[n] = tr ? 1 : 0;
      if (ar.kind == 1) {
        reinterpret_cast<int*>(S)[13] = ep_len_new;
      } else {
        goal_store(S, g);
        if (done) {  // Monitor statistics: accumulated per thread, flushed once per launch (see the kernel end)
          es_n += 1.0; es_ret += (double)ep_ret; es_len += (double)ep_len_fin; es_goal += reached ? 1.0 : 0.0;
          ep_ring_push(ar.ep_stats, ep_ret, (float)ep_len_fin);
        }
      }
      if (!tr) ar.rewards[so] = reward;
    }
    // ---- time-limit bootstrap of the (rare) truncated rows: the wave evaluates the value MLP row by row ----
    unsigned long long pending = __ballot(boot);
    while (pending) {
      const int src = __ffsll((long long)pending) - 1;  // lane that owns the truncated row
      pending &= pending - 1;
      const int br = src >> 1;
      const float rw = __shfl(reward, src, 64);
      float* sc = &lds[wb + LB::H1];  // h1 is dead: scratch h1[G1] | h2[G2]
      const float v = value_row_wave(&lds[wb + LB::H2 + br * GLDH], sc, sc + ar.bt.G1, ar.bt, D, lane);
      if (lane == 0) {
        ar.bt.term_val[row0 + br] = v;
        ar.rewards[(size_t)t * N + row0 + br] = (float)((double)rw + (double)__fmul_rn(ar.bt.gamma, v));
      }
    }
  }
  // ---- episode statistics: one set of atomics per wave and launch (order irrelevant: diagnostics) ----
  if (a.kind == 2) {
    const double n = wave_sum_d(es_n), r = wave_sum_d(es_ret), l = wave_sum_d(es_len), g = wave_sum_d(es_goal);
    if ((tid0 & 63) == 0 && n > 0.0) {
      atomicAdd(&a.ep_stats[0], n); atomicAdd(&a.ep_stats[1], r); atomicAdd(&a.ep_stats[2], l); atomicAdd(&a.ep_stats[3], g);
    }
  }
  // ---- carried state back to global ----
  if (lane0 < R) {
    const int n = row0 + lane0;
    const float* S = &lds[wb + L::ST + lane0 * 16];
    if (n < N) {
      if (a.kind == 2) {
#pragma unroll
        for (int j = 0; j < kGoalStateFloats; ++j) a.gstate[(size_t)n * kGoalStateFloats + j] = S[j];
      } else {
        a.ep_len[n] = reinterpret_cast<const int*>(S)[13];
      }
      a.prev_dones[n] = S[12];
    }
  }
}
#undef ar
#undef EK0
#undef EK1

// weight fragments of one 32-column block held in registers, and a single-chain 32-row GEMM over them (the k order of
// gemm_lds_packed_r32 / gemm_lds_lds_r32; see kernels_split64.h, which has the same pair for the training tile)
template <int NKG>
struct Frags {
  f32x4 f[NKG];
};
template <int NKG>
__device__ __forceinline__ Frags<NKG> load_frags_ro(const f32x4* __restrict__ Bp, int lane) {
  Frags<NKG> w;
  const unsigned bo = opaque_u((unsigned)lane * 16u);
#pragma unroll
  for (int kg = 0; kg < NKG; ++kg) w.f[kg] = ldg16(Bp, bo + (unsigned)kg * 1024u);
  return w;
}
template <int LDA, int NKG>
__device__ __forceinline__ void gemm_one_ro(int a_off, const Frags<NKG>& w, f32x16& c, int lane) {
  const int r = lane & 31, h = lane >> 5;
  const int ab = 4 * opaque((a_off + r * LDA + 4 * h) >> 2);
#pragma unroll
  for (int kg = 0; kg < NKG; ++kg) {
    const f32x4 u = *reinterpret_cast<const f32x4*>(&lds[ab + 8 * kg]);
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) c = MFMA32(u[s_], w.f[kg][s_], c);
  }
}

// ------------------------------------------------------------------------------------------------
// Hidden width 64, ONE WORKGROUP (four waves) per 32-env tile: the layout of k_rollout_persistent (8 threads per row in
// the env phase, block-wide sampling) with the 64-wide forward of k_rollout64_persistent split over two waves.
//
// k_rollout64_persistent gives a tile to one wave: 160 dependent MFMAs (4.3 us) plus Philox, sampling and the env rules
// on 64 lanes -> 11.8 us per step, and a 16-env rollout (data/configs/doggo-ppo.yaml) keeps ONE wave of the chip busy
// for 1000 steps.  Here waves 0 / 1 each own one 32-column block of both hidden layers and one of the two accumulation
// chains of the head, with their weight fragments held in REGISTERS for the whole launch (the weights are constant
// during a rollout: no LDS mirror, no B-operand reads in the step loop); waves 2 / 3 draw the step's normals meanwhile;
// sampling and env.step run on all 256 threads.  Every accumulator sees the MFMA sequence of tile64_forward, the Philox
// counters and the per-row expressions are those of the one-wave kernel -> bit-identical rollouts
// (tests/test_engine_gpu.py::test_rollout64_tile_kernel_is_bit_identical...).  Used while the tiles fit one workgroup
// per CU (N <= 8192); beyond that the one-wave kernel (four tiles per workgroup) has the better throughput.
// ------------------------------------------------------------------------------------------------
template <int DP>
struct LayRoT {
  static constexpr int LDX = DP + 4;
  static constexpr int X = 0;
  static constexpr int H1 = X + 32 * LDX;
  static constexpr int H2 = H1 + 32 * GLDH;
  static constexpr int DO = H2 + 32 * GLDH;   // head partial of wave 0 (even k-groups)
  static constexpr int DO2 = DO + 32 * FLDO;  // head partial of wave 1 (odd k-groups)
  static constexpr int CA = DO2 + 32 * FLDO;  // [32][33] clipped actions
  static constexpr int ST = CA + 32 * 33;     // [32][16] row state
  static constexpr int BL = ST + 32 * 16;     // bootstrap list: cnt[4] | row[32] | reward[32]
  static constexpr int AC = BL + 4 + 64;      // per-action constants [4][32]
  static constexpr int ZN = AC + 128;         // [32][32] standard normals
  static constexpr int TM = ZN + 32 * 32;     // [32][33] log-prob terms
  static constexpr int END = TM + 32 * 33;
  static constexpr int EN = END;              // KIND 0: [32][DP] standard normals of this step's env phase (observation noise), drawn by the noise waves;
                                              // KIND 3: [2][32][DP] what the host wrote for the tile (observations | terminal observations)
  static constexpr int MR = EN + 32 * DP;     // KIND 0, synthetic source: [32][2] per-row draws (termination word | reward normal)
};
inline size_t rollout64_tile_lds_bytes(int Dp, bool served = false) {
  return (size_t)(32 * (Dp + 4) + 2 * 32 * GLDH + 2 * 32 * FLDO + 32 * 33 + 32 * 16 + 68 + 128 + 32 * 32 + 32 * 33 + (served ? 2 * 32 * Dp : 32 * Dp + 64)) * sizeof(float);
}

// Step-loop barriers of the tile kernel: LDS hand-offs only (as in k_rollout_persistent: the step's global stores are read by later
// kernels; __syncthreads() also waits for their acknowledgements at every barrier).  Worth 1 % of config 2's rollout (7.90 -> 7.80 ms),
// nothing at 16 envs; bit-identical.  MOBROB_R64_SYNCTHREADS=1: the old form.
#ifndef MOBROB_R64_SYNCTHREADS
#define MOBROB_R64_SYNCTHREADS 0
#endif
#if MOBROB_R64_SYNCTHREADS
#define R64_BARRIER() __syncthreads()
#else
#define R64_BARRIER() LDS_BARRIER()
#endif
#define ar (*ap_)
#define EK0 ((uint32_t)ar.env_seed)
#define EK1 ((uint32_t)(ar.env_seed >> 32))
// KIND 0: the device env sources (a.kind 1 / 2, a runtime switch as before).  KIND 3 (round 6): HOST environments served by this launch,
// the hand-over of k_rollout_persistent<.., 3, true> for the 64-wide networks every reference config trains
// (/root/reference/data/configs/*-ppo.yaml:20-23): the tile's clipped actions go to the caller's pinned buffer with system-scope
// stores, one lane raises the workgroup's flag word and waits for the host's word of the row range, all four waves pull what the host
// wrote for the 32 rows into LDS, and the step goes on as on the device (storage, time-limit bootstrap, next observation tile).
template <int DP, int KIND = 0>
__global__ __launch_bounds__(256, 1) void k_rollout64_tile(RolloutArgs a) {
  if constexpr (KIND == 3) {   // a launch in front of this one gave up on the host: do nothing (uniform)
    if (__hip_atomic_load(a.abort_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
  }
  using L = LayRoT<DP>;
  constexpr int ldx = L::LDX, per = DP / 4, R = 32, NKG1 = DP / 8;
  const int tid0 = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const FusedNet W = a.pi;
  const int row0 = blockIdx.x * R;
  const int N = a.N, A = a.A, D = a.D;
  int* cnt = reinterpret_cast<int*>(&lds[L::BL]);
  int* lrow = cnt + 4;
  float* lrew = &lds[L::BL + 4 + 32];
  // ---- this wave's weight fragments: registers for the whole launch ----
  Frags<NKG1> f1;
  Frags<8> f2;
  Frags<4> fh;  // head k-groups wave, wave + 2, wave + 4, wave + 6
  float bias1 = 0.f, bias2 = 0.f;
  if (wave < 2) {
    const int l0 = tid0 & 63;
    f1 = load_frags_ro<NKG1>(W.W1f + (size_t)wave * NKG1 * 64, l0);
    f2 = load_frags_ro<8>(W.W2f + (size_t)wave * 8 * 64, l0);
#pragma unroll
    for (int j = 0; j < 4; ++j) fh.f[j] = ldg16(W.W3f, (unsigned)l0 * 16u + (unsigned)(2 * j + wave) * 1024u);
    bias1 = W.b1s[32 * wave + (l0 & 31)];
    bias2 = W.b2s[32 * wave + (l0 & 31)];
  }
  // ---- carried state and the observation tile of step t0 -> LDS ----
  if (tid0 < R) {
    const int n = row0 + tid0;
    float* S = &lds[L::ST + tid0 * 16];
    if (n < N) {
      if constexpr (KIND != 3) {
        if (a.kind == 2) {
#pragma unroll
          for (int j = 0; j < kGoalStateFloats; ++j) S[j] = a.gstate[(size_t)n * kGoalStateFloats + j];
        } else {
          reinterpret_cast<int*>(S)[13] = a.ep_len[n];
        }
      }
      S[12] = a.prev_dones[n];
    }
  }
  if (tid0 >= 64 && tid0 < 96) {  // per-action constants of the Gaussian head
    const int k = tid0 - 64;
    float sd = 1.f, bb = 0.f;
    if (k < A) { sd = expf(a.log_std[k]); bb = W.b3[k]; }
    lds[L::AC + k] = sd;
    lds[L::AC + 32 + k] = 2.0f * (sd * sd);
    lds[L::AC + 64 + k] = logf(sd);
    lds[L::AC + 96 + k] = bb;
  }
  for (int i = tid0; i < R * per; i += 256) {
    const int rr = i / per, c = i - rr * per;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row0 + rr < N)
      v = ldg16(a.obs, (unsigned)((size_t)a.t0 * N + row0 + rr) * (unsigned)(DP * 4) + (unsigned)(c * 16));
    *reinterpret_cast<f32x4*>(&lds[L::X + rr * ldx + 4 * c]) = v;
  }
  if (tid0 == 0) { cnt[0] = 0; cnt[1] = 0; cnt[2] = 0; }
  __syncthreads();
  const uint32_t dbase = a.draw_base ? *a.draw_base : a.draw0, sbase = a.step_base ? *a.step_base : 0u;
  const uint32_t ek0 = (uint32_t)a.env_seed, ek1 = (uint32_t)(a.env_seed >> 32);

  double es_n = 0.0, es_ret = 0.0, es_len = 0.0, es_goal = 0.0;  // finished episodes of this thread's row
  const int t_begin = a.t0, t_end = a.t1;   // (kernel arguments and Philox keys inside the step loop: see k_rollout_persistent)
  for (int t = t_begin; t < t_end; ++t) {
    RolloutArgsK ap_ = rollout_kernargs();
    const int tid = opaque(tid0), lane = tid & 63;
    const int r = lane & 31, h = lane >> 5;
    if (wave < 2) {  // layer 1: column block `wave`
      f32x16 c = splat16(bias1);
      gemm_one_ro<ldx, NKG1>(L::X, f1, c, lane);
      const int o = opaque(L::H1 + 4 * h * GLDH + 32 * wave + r);
#pragma unroll
      for (int i = 0; i < 16; ++i) lds[o + crc(i) * GLDH] = fast_tanh_scaled(c[i]);
    } else {  // standard normals of this step (consumed after the head)
      const int ngrp = (A + 3) >> 2;
      for (int i = tid - 128; i < R * ngrp; i += 128) {
        const int rr_ = i / ngrp, gq = i - rr_ * ngrp;
        float z[4];
        box_muller4(philox4x32_10((uint32_t)(row0 + rr_), (uint32_t)gq, (uint32_t)t + dbase, 0x45505331u, (uint32_t)ar.seed,
                                  (uint32_t)(ar.seed >> 32)), z);
#pragma unroll
        for (int j = 0; j < 4; ++j) lds[L::ZN + rr_ * 32 + 4 * gq + j] = z[j];
      }
    }
    R64_BARRIER(); ap_ = rollout_kernargs();
    if (wave < 2) {  // layer 2
      f32x16 c = splat16(bias2);
      gemm_one_ro<GLDH, 8>(L::H1, f2, c, lane);
      const int o = opaque(L::H2 + 4 * h * GLDH + 32 * wave + r);
#pragma unroll
      for (int i = 0; i < 16; ++i) lds[o + crc(i) * GLDH] = fast_tanh_scaled(c[i]);
    } else if constexpr (KIND == 0) {
      // Round 6: the env phase's observation noise -- one Philox4x32-10 + Box-Muller per (row, 4-column chunk), 100+ VALU instructions
      // that every thread of the env phase used to run in front of its stores -- is drawn HERE, by the two waves that idle through
      // layer 2 and the head.  Same counters (row, chunk, step), same functions: the same bits (the second draw of a row that ends an
      // episode, its reset observation, stays with the env phase: rare).
      const uint32_t step_ = sbase + (uint32_t)t;
      for (int i = tid - 128; i < R * per; i += 128) {
        const int rr_ = i / per, c = i - rr_ * per;
        if (row0 + rr_ < N) {
          float z[4];
          box_muller4(philox4x32_10((uint32_t)(row0 + rr_), (uint32_t)c, step_, kStreamEnvObs, EK0, EK1), z);
          *reinterpret_cast<f32x4*>(&lds[L::EN + rr_ * DP + 4 * c]) = f32x4{z[0], z[1], z[2], z[3]};
        }
      }
    }
    R64_BARRIER(); ap_ = rollout_kernargs();
    if constexpr (KIND == 0) {
      if (wave >= 2 && ar.kind == 1 && tid - 128 < R && row0 + (tid - 128) < N) {  // synthetic source: the row's termination word and reward normal
        const Philox4 mr = philox4x32_10((uint32_t)(row0 + tid - 128), 0u, sbase + (uint32_t)t, kStreamEnvMisc, EK0, EK1);
        float zz[4];
        box_muller4(Philox4{mr.y, mr.z, mr.w, mr.x ^ 0x9E3779B9u}, zz);
        reinterpret_cast<uint32_t*>(&lds[L::MR])[2 * (tid - 128)] = mr.x;
        lds[L::MR + 2 * (tid - 128) + 1] = zz[0];
      }
    }
    if (wave < 2) {  // head: wave 0 the even k-groups (tile64_forward's `acc`), wave 1 the odd ones (`acc2`)
      f32x16 acc = zero16();
      const int ab = 4 * opaque((L::H2 + r * GLDH + 4 * h) >> 2);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(&lds[ab + (2 * j + wave) * 8]);
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) acc = MFMA32(av[s_], fh.f[j][s_], acc);
      }
      const int o = opaque((wave == 0 ? L::DO : L::DO2) + 4 * h * FLDO + r);
#pragma unroll
      for (int i = 0; i < 16; ++i) lds[o + crc(i) * FLDO] = acc[i];
    }
    R64_BARRIER(); ap_ = rollout_kernargs();
    // ---- Gaussian sample + log-prob (expressions and Philox counters of k_fused64_act), then env.step(clipped actions) +
    //      auto-reset: the SAME 8 threads serve a row in both (actions sub, sub + 8, ...; observation chunks sub, sub + 8):
    //      they sit in one wave, so the row's clipped actions and log-prob terms are ordered by the wave's DS queue and no
    //      workgroup barrier separates sampling from the env step ----
    const int rr = tid >> 3, sub = tid & 7;
    if (row0 + rr < N) {
      for (int k = sub; k < A; k += 8) {
        const float m = (lds[L::DO + rr * FLDO + k] + lds[L::DO2 + rr * FLDO + k]) + lds[L::AC + 96 + k];
        const float sd = lds[L::AC + k];
        const float act = m + lds[L::ZN + rr * 32 + k] * sd;
        const float d = act - m;
        lds[L::TM + rr * 33 + k] = -(d * d) / lds[L::AC + 32 + k] - lds[L::AC + 64 + k] - 0.91893853320467274178f;
        const float ac = fminf(fmaxf(act, ar.lo), ar.hi);
        ar.actions[((size_t)t * N + row0 + rr) * A + k] = act;
        if constexpr (KIND == 3) __hip_atomic_store(&ar.clip_act[(size_t)(row0 + rr) * A + k], ac, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // pinned host memory, past the caches
        else ar.clip_act[(size_t)(row0 + rr) * A + k] = ac;
        lds[L::CA + rr * 33 + k] = ac;
      }
    }
    __builtin_amdgcn_wave_barrier();  // the row's 8 lanes have issued their LDS writes (same wave: in-order DS queue)
    if (sub == 0 && row0 + rr < N) {
      float lp = 0.f;
      for (int k = 0; k < A; ++k) lp += lds[L::TM + rr * 33 + k];
      ar.logp[(size_t)t * N + row0 + rr] = lp;
    }
    if constexpr (KIND == 3) {
      // Hand-over (the protocol of k_rollout_persistent<.., 3, true>): every storing wave drains its stores -- the actions have left for
      // host memory --, a barrier, then ONE lane tells the host "step t of these 32 rows" and waits for the host's word of the row range
      // ((truncated rows) << 32 | steps finished; 0xFFFFFFFF: give up).  Only the range's FIRST workgroup polls host memory and relays
      // what it saw through a device word; bounded; one system-scope acquire between the word's arrival and the pull.
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      R64_BARRIER(); ap_ = rollout_kernargs();
      if (tid == 0) {
        __hip_atomic_store(ar.h_gpu_flag + blockIdx.x, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const int part = row0 / ar.rows_per_part;
        const bool relay = row0 == part * ar.rows_per_part;
        const unsigned long long* hf = reinterpret_cast<const unsigned long long*>(ar.h_host_flag + 16 * part);
        unsigned long long* df = reinterpret_cast<unsigned long long*>(ar.abort_dev) + 1 + part;
        const long long w0 = wall_clock64();
        unsigned long long v;
        for (;;) {
          v = relay ? __hip_atomic_load(hf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : __hip_atomic_load(df, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((unsigned)v >= (unsigned)(t + 1)) break;
          if (wall_clock64() - w0 > ar.timeout_ticks) {
            v = 0xFFFFFFFFull;
            __hip_atomic_store(ar.h_error, t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
          }
          if (relay) __builtin_amdgcn_s_sleep(8); else __builtin_amdgcn_s_sleep(2);
        }
        if (relay) __hip_atomic_store(df, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const bool dead = (unsigned)v == 0xFFFFFFFFu;
        if (dead) __hip_atomic_store(ar.abort_dev, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        cnt[1] = dead ? 1 : 0;
        cnt[2] = dead ? 0 : (int)(v >> 32);
      }
      R64_BARRIER(); ap_ = rollout_kernargs();  // (4b) the host has stepped the rows (or the wait was given up)
      if (cnt[1]) break;
      {  // what the host wrote for this tile's 32 rows -> LDS: one contiguous, 16-byte aligned range of the [N][D] arrays, whole-wave
         // 16-byte loads past the caches; rewards / done / truncated flags into the log-prob term buffer (free until the next sampling)
        const int nrow = min(R, N - row0);
        const int nq = nrow * D / 4, rem0 = 4 * nq, nfl = nrow * D;
        const f32x4* so = reinterpret_cast<const f32x4*>(ar.h_obs + (size_t)row0 * D);
        for (int i = tid; i < nq; i += 256) {
          f32x4 v;
          asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(so + i) : "memory");
          *reinterpret_cast<f32x4*>(&lds[L::EN + 4 * i]) = v;
        }
        for (int i = rem0 + tid; i < nfl; i += 256)
          lds[L::EN + i] = __hip_atomic_load(ar.h_obs + (size_t)row0 * D + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (cnt[2] != 0) {
          const f32x4* st = reinterpret_cast<const f32x4*>(ar.h_term + (size_t)row0 * D);
          for (int i = tid; i < nq; i += 256) {
            f32x4 v;
            asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(st + i) : "memory");
            *reinterpret_cast<f32x4*>(&lds[L::EN + 32 * DP + 4 * i]) = v;
          }
          for (int i = rem0 + tid; i < nfl; i += 256)
            lds[L::EN + 32 * DP + i] = __hip_atomic_load(ar.h_term + (size_t)row0 * D + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (tid < nrow) {
          lds[L::TM + tid] = __hip_atomic_load(ar.h_rew + row0 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          reinterpret_cast<int*>(&lds[L::TM])[32 + tid] = __hip_atomic_load(ar.h_done + row0 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          reinterpret_cast<int*>(&lds[L::TM])[64 + tid] = cnt[2] != 0 ? (int)__hip_atomic_load(ar.h_trunc + row0 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0;
        }
      }
      R64_BARRIER(); ap_ = rollout_kernargs();  // (4c) the host's tile is in LDS
    }
    const int n = row0 + rr;
    const bool live = n < N;
    const uint32_t step = sbase + (uint32_t)t;
    float* S = &lds[L::ST + rr * 16];
    float* xrow = &lds[L::X + rr * ldx];
    float* trow = &lds[L::H2 + rr * GLDH];  // terminal observation staging (h2 is dead after the head GEMM)
    const size_t onext = ((size_t)(t + 1) * N + (live ? n : 0)) * per;
    bool tr = false, done = false, reached = false;
    float reward = 0.f, ep_ret = 0.f;
    int ep_len_new = 0, ep_len_fin = 0;
    GoalState g{};
    if constexpr (KIND == 3) {
      if (live) {   // what the host wrote for this row (staged above): reward, done, truncated, next observation, terminal one
        done = reinterpret_cast<const int*>(&lds[L::TM])[32 + rr] != 0;
        tr = reinterpret_cast<const int*>(&lds[L::TM])[64 + rr] != 0;
        if (sub == 0) reward = lds[L::TM + rr];
        auto staged_chunk = [&](int base, int c) {   // columns 4 c .. 4 c + 3 of the tile's row rr ([32][D] floats at `base`)
          f32x4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int col = 4 * c + j;
            o[j] = col < D ? lds[base + rr * D + (col < D ? col : 0)] : 0.f;
          }
          return o;
        };
        for (int c = sub; c < per; c += 8) {
          const f32x4 o = staged_chunk(L::EN, c);
          if (tr) {
            const f32x4 to = staged_chunk(L::EN + 32 * DP, c);
            reinterpret_cast<f32x4*>(ar.term_obs)[(size_t)n * per + c] = to;
            *reinterpret_cast<f32x4*>(&trow[4 * c]) = to;
          }
          reinterpret_cast<f32x4*>(ar.obs)[onext + c] = o;
          *reinterpret_cast<f32x4*>(&xrow[4 * c]) = o;
        }
      }
    } else if (live) {
      if (ar.kind == 1) {
        const bool term = u32_to_unit_open(reinterpret_cast<const uint32_t*>(&lds[L::MR])[2 * rr]) < ar.p_term;   // drawn by the noise waves
        const int len = reinterpret_cast<const int*>(S)[13] + 1;
        tr = (len >= ar.time_limit) && !term;
        done = term || tr;
        ep_len_new = done ? 0 : len;
        for (int c = sub; c < per; c += 8) {
          float z[4];
          {
            const f32x4 zq = *reinterpret_cast<const f32x4*>(&lds[L::EN + rr * DP + 4 * c]);   // drawn by the noise waves
            z[0] = zq[0]; z[1] = zq[1]; z[2] = zq[2]; z[3] = zq[3];
          }
          f32x4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = (4 * c + j < D) ? z[j] : 0.f;
          if (tr) {
            reinterpret_cast<f32x4*>(ar.term_obs)[(size_t)n * per + c] = o;
            *reinterpret_cast<f32x4*>(&trow[4 * c]) = o;
            box_muller4(philox4x32_10((uint32_t)n, (uint32_t)c, step, kStreamEnvTerm, EK0, EK1), z);
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (4 * c + j < D) ? z[j] : 0.f;
          }
          reinterpret_cast<f32x4*>(ar.obs)[onext + c] = o;
          *reinterpret_cast<f32x4*>(&xrow[4 * c]) = o;
        }
        if (sub == 0) reward = 0.03f + 0.1f * lds[L::MR + 2 * rr + 1] + (term ? 5.0f : 0.f);
      } else {
        g = goal_load(S);
        const GoalOutcome o = goal_advance(g, ar.goal, &lds[L::CA + rr * 33], A);
        tr = o.tr; done = o.done; reached = o.reached; reward = o.reward;
        ep_ret = g.ep_ret; ep_len_fin = g.ep_len;
        GoalState gn = g;
        if (done) goal_reset(gn, ar.goal, o.reached, (uint32_t)n, step, EK0, EK1);
        for (int c = sub; c < per; c += 8) {
          float z[4];
          {
            const f32x4 zq = *reinterpret_cast<const f32x4*>(&lds[L::EN + rr * DP + 4 * c]);   // drawn by the noise waves
            z[0] = zq[0]; z[1] = zq[1]; z[2] = zq[2]; z[3] = zq[3];
          }
          f32x4 ob = goal_features(g, ar.goal.P, D, c, z, ar.goal.noise);
          if (done) {
            if (tr) {
              reinterpret_cast<f32x4*>(ar.term_obs)[(size_t)n * per + c] = ob;
              *reinterpret_cast<f32x4*>(&trow[4 * c]) = ob;
            }
            box_muller4(philox4x32_10((uint32_t)n, (uint32_t)c, step, kStreamEnvTerm, EK0, EK1), z);
            ob = goal_features(gn, ar.goal.P, D, c, z, ar.goal.noise);
          }
          reinterpret_cast<f32x4*>(ar.obs)[onext + c] = ob;
          *reinterpret_cast<f32x4*>(&xrow[4 * c]) = ob;
        }
        g = gn;
      }
    }
    // the 8 threads of a row sit in ONE wave and have consumed the row's old state (program order): commit
    if (live && sub == 0) {
      const size_t so = (size_t)t * N + n;
      ar.es[so] = S[12];
      S[12] = done ? 1.f : 0.f;
      ar.trunc[n] = tr ? 1 : 0;
      if constexpr (KIND != 3) {
      if (ar.kind == 1) {
        reinterpret_cast<int*>(S)[13] = ep_len_new;
      } else {
        goal_store(S, g);
        if (done) {  // Monitor statistics: accumulated per thread, flushed once per launch (see the kernel end)
          es_n += 1.0; es_ret += (double)ep_ret; es_len += (double)ep_len_fin; es_goal += reached ? 1.0 : 0.0;
          ep_ring_push(ar.ep_stats, ep_ret, (float)ep_len_fin);
        }
      }
      }
      if (tr) {  // reward is written after the bootstrap below
        const int q = atomicAdd(cnt, 1);
        lrow[q] = rr;
        lrew[q] = reward;
      } else {
        ar.rewards[so] = reward;
      }
    }
    R64_BARRIER(); ap_ = rollout_kernargs();  // next observation tile, row state and the bootstrap list are complete
    // ---- time-limit bootstrap of the (rare) truncated rows: wave 0 evaluates the value MLP row by row ----
    const int m = *cnt;
    if (m > 0) {  // block-uniform
      if (wave == 0) {
        for (int q = 0; q < m; ++q) {
          const int br = lrow[q];
          float* sc = &lds[L::H1];  // h1 is dead: scratch h1[G1] | h2[G2]
          const float v = value_row_wave(&lds[L::H2 + br * GLDH], sc, sc + ar.bt.G1, ar.bt, D, lane);
          if (lane == 0) {
            ar.bt.term_val[row0 + br] = v;
            ar.rewards[(size_t)t * N + row0 + br] = (float)((double)lrew[q] + (double)__fmul_rn(ar.bt.gamma, v));
          }
        }
      }
      __syncthreads(); ap_ = rollout_kernargs();
      if (tid == 0) *cnt = 0;  // read again only after the next step's barriers
    }
  }
  // ---- episode statistics: one set of atomics per wave and launch (order irrelevant: diagnostics) ----
  if (KIND != 3 && a.kind == 2) {
    const double n = wave_sum_d(es_n), r = wave_sum_d(es_ret), l = wave_sum_d(es_len), g = wave_sum_d(es_goal);
    if ((tid0 & 63) == 0 && n > 0.0) {
      atomicAdd(&a.ep_stats[0], n); atomicAdd(&a.ep_stats[1], r); atomicAdd(&a.ep_stats[2], l); atomicAdd(&a.ep_stats[3], g);
    }
  }
  // ---- carried state back to global ----
  if (tid0 < R) {
    const int n = row0 + tid0;
    const float* S = &lds[L::ST + tid0 * 16];
    if (n < N) {
      if constexpr (KIND != 3) {
        if (a.kind == 2) {
#pragma unroll
          for (int j = 0; j < kGoalStateFloats; ++j) a.gstate[(size_t)n * kGoalStateFloats + j] = S[j];
        } else {
          a.ep_len[n] = reinterpret_cast<const int*>(S)[13];
        }
      }
      a.prev_dones[n] = S[12];
    }
  }
}
#undef ar
#undef EK0
#undef EK1

}  // namespace mobrob
