// Fused fast path for hidden width 256 (BASELINE configs 3-4: doggo 58/12, 2x256).
//
// One persistent 256-thread workgroup per CU (4 waves, one per SIMD, up to 512 unified VGPR/AGPR each), specialised on one of the two
// independent networks (policy / value).  For each 64-row tile of the minibatch the whole
// forward -> loss -> backward chain stays on chip:
//     LDS (160 KB):  X tile [64][Dp+4] | h1 [64][260] | h2 [64][260] | head/dout [64][36] | row scalars
//     activations never touch HBM; dz2 / dz1 overwrite h2 / h1 in place
//     weights are streamed from L2 in an MFMA-fragment-ordered packing (1 KiB per wave load)
//     dW2 (256x256) is accumulated in registers across all tiles of the workgroup (256 registers per lane),
//     dW1 / dW3 / biases likewise; one slab store per workgroup at the end, then a deterministic
//     slab reduction kernel (no float atomics -> run-to-run reproducible gradients).
// All contractions use v_mfma_f32_32x32x2_f32 (exact f32).  Fragment maps: see kernels_generic.h.
#pragma once
#include "device_utils.h"

namespace mobrob {

constexpr int FH = 256;          // hidden width of the fused path
constexpr int FR = 64;           // rows per tile
constexpr int FLDH = FH + 4;     // padded LDS row stride of h1/h2 (conflict-free ds_read_b128: 260 % 64 == 4)
constexpr int FLDO = 36;         // head tile [64][32] + pad
constexpr int FTHREADS = 256;   // 4 waves: wave w owns output columns [64w, 64w+64)

// packed weights of one network (device pointers)
struct FusedNet {
  const f32x4* W1f;  // [H/32][Dp/8][64]   fwd pack of W1 [H][Dp]
  const f32x4* W2f;  // [H/32][H/8][64]    fwd pack of W2 [H][H]
  const f32x4* W3f;  // [1][H/8][64]       fwd pack of head [32 (zero padded)][H]
  const f32x4* W2b;  // [H/32][H/8][64]    bwd pack: B[k=n][j] = W2[n][j]
  const f32x4* W3b;  // [H/32][32/8][64]   bwd pack: B[k=a][j] = head[a][j], a < 32 zero padded
  const float* b1; const float* b2; const float* b3;  // canonical biases
  int head;          // A for the policy net, 1 for the value net
};

struct FusedTrainArgs {
  FusedNet net[2];               // 0 = policy, 1 = value
  // rollout storage
  const float* obs; int Dp;
  const float* actions; int A;
  const float* old_logp; const float* adv; const float* ret;
  const int* rows;               // permuted row indices of this minibatch
  int count;                     // rows in this minibatch (local)
  const float* log_std;
  const double* advstat;         // (sum, sumsq, n, -) of the GLOBAL minibatch
  int normalize;
  float clip, vf_coef, ent_coef, inv_bg;
  float* slabs;                  // [gridDim.x][slab_floats]
  int slab_floats;
  float* sums;                   // [8] loss statistics (float atomics; diagnostics only)
};

// slab layout (floats): dW2 [H][H] | dW1 [H][Dp] | dW3 [32][H] | db2 [H] | db1 [H] | db3 [32] | dls [32]
__host__ __device__ inline int slab_off_w2() { return 0; }
__host__ __device__ inline int slab_off_w1() { return FH * FH; }
__host__ __device__ inline int slab_off_w3(int Dp) { return FH * FH + FH * Dp; }
__host__ __device__ inline int slab_off_b2(int Dp) { return slab_off_w3(Dp) + 32 * FH; }
__host__ __device__ inline int slab_off_b1(int Dp) { return slab_off_b2(Dp) + FH; }
__host__ __device__ inline int slab_off_b3(int Dp) { return slab_off_b1(Dp) + FH; }
__host__ __device__ inline int slab_off_ls(int Dp) { return slab_off_b3(Dp) + 32; }
__host__ __device__ inline int slab_size(int Dp) { return slab_off_ls(Dp) + 32; }

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// tanh(x) = 1 - 2 / (exp(2x) + 1) on the hardware exp2/rcp units (v_exp_f32, v_rcp_f32: ~1 ulp each).
// Absolute error <= ~2e-7 over the whole range (saturates correctly to +-1); the relative error grows for
// |x| < 1e-3 where tanh(x) ~ x, which is irrelevant at the 1e-4 parity tolerance of O(1) activations.
__device__ __forceinline__ float fast_tanh(float x) {
  const float t = __builtin_amdgcn_exp2f(x * 2.88539008177792681472f);  // exp(2x)
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(t + 1.0f);
}

// All LDS accesses go through ONE extern array with integer (float-unit) offsets, so that every address is
// "per-lane base + compile-time constant" and folds into the DS instructions' 16-bit immediate.  Per-lane bases
// are passed through opaque() inside the tile loop: without it LLVM's LICM hoists hundreds of loop-invariant
// address computations out of the tile loop and spills them.
extern __shared__ __attribute__((aligned(16))) float lds[];
__device__ __forceinline__ int opaque(int x) {
  asm volatile("" : "+v"(x));
  return x;
}
// C layout: element i of a 32x32 accumulator sits at row crc(i) + 4*h, column r of the tile
__device__ __forceinline__ constexpr int crc(int i) { return (i & 3) + 8 * (i >> 2); }

// acc[cb][rb] += A[rb*32 + 0..31][0..8*nkg) . Bpacked[cb]  for this wave's two 32-column blocks.
// a_off: LDS offset of the A tile (row stride lda floats); Bp0/Bp1: packed fragments [nkg][64] of the two blocks.
// B fragments are prefetched two k-groups ahead (global/L2 latency); A fragments one k-group ahead (LDS).
template <int LDA>
__device__ __forceinline__ void gemm_lds_packed(int a_off, const f32x4* __restrict__ Bp0,
                                                const f32x4* __restrict__ Bp1, int nkg, f32x16& c00, f32x16& c01,
                                                f32x16& c10, f32x16& c11, int lane) {
  const int r = lane & 31, h = lane >> 5;
  const int ab = opaque(a_off + r * LDA + 4 * h);
  Bp0 += lane;
  Bp1 += lane;
  f32x4 b0 = Bp0[0], b1 = Bp1[0];
  f32x4 b0n = nkg > 1 ? Bp0[64] : b0, b1n = nkg > 1 ? Bp1[64] : b1;
  f32x4 a0 = *reinterpret_cast<const f32x4*>(&lds[ab]);
  f32x4 a1 = *reinterpret_cast<const f32x4*>(&lds[ab + 32 * LDA]);
#pragma unroll 2
  for (int kg = 0; kg < nkg; ++kg) {
    const f32x4 p = b0, q = b1, u = a0, v = a1;
    b0 = b0n; b1 = b1n;
    if (kg + 2 < nkg) { b0n = Bp0[(kg + 2) * 64]; b1n = Bp1[(kg + 2) * 64]; }
    if (kg + 1 < nkg) {
      a0 = *reinterpret_cast<const f32x4*>(&lds[ab + (kg + 1) * 8]);
      a1 = *reinterpret_cast<const f32x4*>(&lds[ab + 32 * LDA + (kg + 1) * 8]);
    }
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) {
      c00 = MFMA32(u[s_], p[s_], c00);
      c01 = MFMA32(v[s_], p[s_], c01);
      c10 = MFMA32(u[s_], q[s_], c10);
      c11 = MFMA32(v[s_], q[s_], c11);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// dW2 accumulators: 16 tiles of 32x32 (this wave's 64 neurons x 256 inputs) pinned to the accumulator
// register file for the whole kernel.  The library is compiled with -mllvm -amdgpu-mfma-vgpr-form, so every
// compiler-selected MFMA keeps its accumulator in arch VGPRs (work tiles, <= 64 registers); the persistent
// tiles are only ever touched by the statement below, whose "+a" constraints make the register allocator keep
// all 256 AGPRs occupied by them (so it cannot park anything else there).  Tile t = ib*8 + jb.
// ------------------------------------------------------------------------------------------------
#define MFA(t, x, y) "v_mfma_f32_32x32x2_f32 %" #t ", %" #x ", %" #y ", %" #t "\n\t"
// one k-step (two batch rows) of dW2 += dz2^T . h1: 16 MFMAs; x = A operands (2 neuron blocks), y = B operands
__device__ __forceinline__ void dw2_kstep(f32x16 (&g)[16], float x0, float x1, float y0, float y1, float y2,
                                          float y3, float y4, float y5, float y6, float y7) {
  asm volatile(
      "s_nop 1\n\t"  // VALU-written operand -> MFMA
      MFA(0, 16, 18) MFA(8, 17, 18) MFA(1, 16, 19) MFA(9, 17, 19) MFA(2, 16, 20) MFA(10, 17, 20)
      MFA(3, 16, 21) MFA(11, 17, 21) MFA(4, 16, 22) MFA(12, 17, 22) MFA(5, 16, 23) MFA(13, 17, 23)
      MFA(6, 16, 24) MFA(14, 17, 24) MFA(7, 16, 25) MFA(15, 17, 25)
      "s_nop 1"
      : "+a"(g[0]), "+a"(g[1]), "+a"(g[2]), "+a"(g[3]), "+a"(g[4]), "+a"(g[5]), "+a"(g[6]), "+a"(g[7]), "+a"(g[8]),
        "+a"(g[9]), "+a"(g[10]), "+a"(g[11]), "+a"(g[12]), "+a"(g[13]), "+a"(g[14]), "+a"(g[15])
      : "v"(x0), "v"(x1), "v"(y0), "v"(y1), "v"(y2), "v"(y3), "v"(y4), "v"(y5), "v"(y6), "v"(y7));
}

// ------------------------------------------------------------------------------------------------
// epilogue helpers for the wave's 64x64 output block (2 column blocks x 2 row blocks, C layout)
// ------------------------------------------------------------------------------------------------
// lds[dst][row][col] = tanh(acc + bias[col])
__device__ __forceinline__ void store_tanh(int dst_off, const float* __restrict__ bias, int wave, int lane,
                                           const f32x16& c00, const f32x16& c01, const f32x16& c10,
                                           const f32x16& c11) {
  const int r = lane & 31, h = lane >> 5;
  const float bz0 = bias[64 * wave + r], bz1 = bias[64 * wave + 32 + r];
  const int o = opaque(dst_off + 4 * h * FLDH + 64 * wave + r);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    lds[o + crc(i) * FLDH] = fast_tanh(c00[i] + bz0);
    lds[o + (32 + crc(i)) * FLDH] = fast_tanh(c01[i] + bz0);
    lds[o + crc(i) * FLDH + 32] = fast_tanh(c10[i] + bz1);
    lds[o + (32 + crc(i)) * FLDH + 32] = fast_tanh(c11[i] + bz1);
  }
}
// lds[hs][row][col] <- acc * (1 - hs^2) in place; accumulates the column sums (both lane halves hold partials)
__device__ __forceinline__ void dtanh_inplace(int hs_off, int wave, int lane, const f32x16& c00, const f32x16& c01,
                                              const f32x16& c10, const f32x16& c11, float& cs0, float& cs1) {
  const int r = lane & 31, h = lane >> 5;
  const int o = opaque(hs_off + 4 * h * FLDH + 64 * wave + r);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    float hv, z;
    hv = lds[o + crc(i) * FLDH];             z = c00[i] * (1.0f - hv * hv); lds[o + crc(i) * FLDH] = z;             cs0 += z;
    hv = lds[o + (32 + crc(i)) * FLDH];      z = c01[i] * (1.0f - hv * hv); lds[o + (32 + crc(i)) * FLDH] = z;      cs0 += z;
    hv = lds[o + crc(i) * FLDH + 32];        z = c10[i] * (1.0f - hv * hv); lds[o + crc(i) * FLDH + 32] = z;        cs1 += z;
    hv = lds[o + (32 + crc(i)) * FLDH + 32]; z = c11[i] * (1.0f - hv * hv); lds[o + (32 + crc(i)) * FLDH + 32] = z; cs1 += z;
  }
}

// LDS carve-up (float offsets) for a given padded observation width
template <int DP>
struct Lay {
  static constexpr int LDX = DP + 4;
  static constexpr int X = 0;
  static constexpr int H1 = X + FR * LDX;
  static constexpr int H2 = H1 + FR * FLDH;
  static constexpr int DO = H2 + FR * FLDH;
  static constexpr int END = DO + FR * FLDO;
};

// ------------------------------------------------------------------------------------------------
// forward of one 64-row tile through one network; leaves h1, h2 in LDS and the raw head tile
// (without bias) in the head tile [64][FLDO].  All 4 waves participate; ends with a barrier.
// ------------------------------------------------------------------------------------------------
template <int DP>
__device__ __forceinline__ void tile_forward(const FusedNet& W, int wave, int lane) {
  using L = Lay<DP>;
  const int r = lane & 31, h = lane >> 5;
  {  // layer 1: K = DP
    f32x16 c00 = zero16(), c01 = zero16(), c10 = zero16(), c11 = zero16();
    constexpr int nkg = DP / 8;
    gemm_lds_packed<L::LDX>(L::X, W.W1f + (size_t)(2 * wave) * nkg * 64, W.W1f + (size_t)(2 * wave + 1) * nkg * 64,
                            nkg, c00, c01, c10, c11, lane);
    store_tanh(L::H1, W.b1, wave, lane, c00, c01, c10, c11);
  }
  __syncthreads();
  {  // layer 2: K = H
    f32x16 c00 = zero16(), c01 = zero16(), c10 = zero16(), c11 = zero16();
    constexpr int nkg = FH / 8;
    gemm_lds_packed<FLDH>(L::H1, W.W2f + (size_t)(2 * wave) * nkg * 64, W.W2f + (size_t)(2 * wave + 1) * nkg * 64, nkg,
                          c00, c01, c10, c11, lane);
    store_tanh(L::H2, W.b2, wave, lane, c00, c01, c10, c11);
  }
  __syncthreads();
  {  // head: [64 x 32] = h2 . W3^T, K split in two halves; wave = (khalf << 1) | rowblock
    const int rb = wave & 1, ks = wave >> 1;
    f32x16 acc = zero16();
    const int ab = opaque(L::H2 + (rb * 32 + r) * FLDH + ks * 128 + 4 * h);
    const f32x4* bp = W.W3f + (size_t)(ks * 16) * 64 + lane;
#pragma unroll 4
    for (int kg = 0; kg < 16; ++kg) {
      const f32x4 b = bp[kg * 64];
      const f32x4 a = *reinterpret_cast<const f32x4*>(&lds[ab + kg * 8]);
      acc = MFMA32(a[0], b[0], acc);
      acc = MFMA32(a[1], b[1], acc);
      acc = MFMA32(a[2], b[2], acc);
      acc = MFMA32(a[3], b[3], acc);
    }
    // deterministic cross-wave reduction through the head tile: one K-half per round
    const int o = opaque(L::DO + (rb * 32 + 4 * h) * FLDO + r);
#pragma unroll
    for (int round = 0; round < 2; ++round) {
      if (ks == round) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          if (round == 0) lds[o + crc(i) * FLDO] = acc[i];
          else lds[o + crc(i) * FLDO] += acc[i];
        }
      }
      __syncthreads();
    }
  }
}

// ------------------------------------------------------------------------------------------------
// The training kernel.
// ------------------------------------------------------------------------------------------------
template <int DP>
__global__ __launch_bounds__(FTHREADS, 1) void k_fused_train(FusedTrainArgs a) {
  using L = Lay<DP>;
  constexpr int ldx = L::LDX, per = DP / 4;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 31, h = lane >> 5;
  const int net = blockIdx.x & 1, wg = blockIdx.x >> 1, nwg = gridDim.x >> 1;
  const FusedNet W = a.net[net];
  const int ntiles = (a.count + FR - 1) / FR;

  // dW2 lives in the 256 accumulator registers for the whole kernel; dW1 / dW3 are small and are accumulated
  // per tile into the workgroup's private slab (read-modify-write by the owning lane, L2 resident).
  f32x16 gW2[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) gW2[t] = zero16();
  float* slab = a.slabs + (size_t)blockIdx.x * a.slab_floats;
  float* slab_w1 = slab + slab_off_w1();
  float* slab_w3 = slab + slab_off_w3(DP);
  bool first = true;
  float gb2a = 0.f, gb2b = 0.f, gb1a = 0.f, gb1b = 0.f;  // bias-gradient partials for cols 64w+r, 64w+32+r
  float gb3 = 0.f, gls = 0.f;                            // wave 0, lane < head
  float s_pl = 0.f, s_vl = 0.f, s_kl = 0.f, s_cf = 0.f;  // wave 0 loss statistics

  float adv_mean = 0.f, adv_sd = 1.f;
  bool adv_on = false;
  {
    const double n = a.advstat[2];
    adv_on = n > 1.0;
    const double m = a.advstat[0] / (n > 0 ? n : 1.0);
    double var = adv_on ? (a.advstat[1] - n * m * m) / (n - 1.0) : 0.0;
    if (var < 0.0) var = 0.0;
    adv_mean = (float)m;
    adv_sd = (float)sqrt(var);
  }

  for (int tile = wg; tile < ntiles; tile += nwg) {
    const int row0 = tile * FR;
    // ---- gather the observation rows of this tile (zero rows beyond the minibatch) ----
#pragma unroll
    for (int i = tid; i < FR * per; i += FTHREADS) {
      const int rr = i / per, c = i - rr * per;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (row0 + rr < a.count) {
        const int src = a.rows[row0 + rr];
        v = reinterpret_cast<const f32x4*>(a.obs)[(size_t)src * per + c];
      }
      *reinterpret_cast<f32x4*>(&lds[L::X + rr * ldx + 4 * c]) = v;
    }
    __syncthreads();
    tile_forward<DP>(W, wave, lane);

    // ---- loss: one lane per row (wave 0); writes dL/d(head output) into the head tile, zero padded ----
    if (wave == 0) {
      const int rr = lane;
      const bool live = row0 + rr < a.count;
      const int src = live ? a.rows[row0 + rr] : 0;
      float* drow = &lds[L::DO + rr * FLDO];
      if (net == 0) {
        float g_logp = 0.f;
        if (live) {
          float adv = a.adv[src];
          if (a.normalize && adv_on) adv = (adv - adv_mean) / (adv_sd + 1e-8f);
          float lp = 0.f;
          for (int k = 0; k < a.A; ++k) {
            const float sd = expf(a.log_std[k]);
            const float d = a.actions[(size_t)src * a.A + k] - (drow[k] + W.b3[k]);
            lp += -(d * d) / (2.0f * (sd * sd)) - logf(sd) - 0.91893853320467274178f;
          }
          const float log_ratio = lp - a.old_logp[src];
          const float ratio = expf(log_ratio);
          const float lo = 1.0f - a.clip, hi = 1.0f + a.clip;
          const float s1 = adv * ratio, s2 = adv * fminf(fmaxf(ratio, lo), hi);
          s_pl += fminf(s1, s2);
          s_cf += (fabsf(ratio - 1.0f) > a.clip) ? 1.f : 0.f;
          s_kl += (ratio - 1.0f) - log_ratio;
          const float in_range = (ratio >= lo && ratio <= hi) ? 1.f : 0.f;
          const float w1 = (s1 < s2) ? 1.f : ((s1 > s2) ? 0.f : 0.5f);
          g_logp = -(w1 * adv + (1.0f - w1) * adv * in_range) * a.inv_bg * ratio;
        }
        for (int k = 0; k < 32; ++k) {
          float gm = 0.f, gl = 0.f;
          if (k < a.A && live) {
            const float sd = expf(a.log_std[k]);
            const float var = sd * sd;
            const float d = a.actions[(size_t)src * a.A + k] - (drow[k] + W.b3[k]);
            gm = g_logp * d / var;
            gl = g_logp * (d * d / var - 1.0f);
          }
          drow[k] = gm;
          if (k < a.A) {  // wave-uniform
            const float t1 = wave_sum(gm), t2 = wave_sum(gl);
            if (lane == k) { gb3 += t1; gls += t2; }
          }
        }
      } else {
        float dv = 0.f;
        if (live) {
          const float v = drow[0] + W.b3[0], rt = a.ret[src];
          s_vl += (rt - v) * (rt - v);
          dv = a.vf_coef * 2.0f * (v - rt) * a.inv_bg;
        }
        drow[0] = dv;
        for (int k = 1; k < 32; ++k) drow[k] = 0.f;
        const float t = wave_sum(dv);
        if (lane == 0) gb3 += t;
      }
    }
    __syncthreads();

    // ---- dW3 (+)= dout^T . h2  (M = 32 head rows, this wave's 64 columns, K = 64 rows) ----
    {
      f32x16 t0 = zero16(), t1 = zero16();
      const int ao = opaque(L::DO + h * FLDO + r);              // A[i=a][k=row] = dout[row][a]
      const int bo = opaque(L::H2 + h * FLDH + 64 * wave + r);  // B[k=row][j]  = h2[row][j]
#pragma unroll 4
      for (int k = 0; k < FR; k += 2) {
        const float x = lds[ao + k * FLDO];
        t0 = MFMA32(x, lds[bo + k * FLDH], t0);
        t1 = MFMA32(x, lds[bo + k * FLDH + 32], t1);
      }
      const int so = opaque(4 * h * FH + 64 * wave + r);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float* p = slab_w3 + so + crc(i) * FH;
        p[0] = first ? t0[i] : p[0] + t0[i];
        p[32] = first ? t1[i] : p[32] + t1[i];
      }
    }
    // ---- dh2 = dout . W3 (K = 32), then dz2 = dh2 * (1 - h2^2) in place ----
    {
      f32x16 c00 = zero16(), c01 = zero16(), c10 = zero16(), c11 = zero16();
      gemm_lds_packed<FLDO>(L::DO, W.W3b + (size_t)(2 * wave) * 4 * 64, W.W3b + (size_t)(2 * wave + 1) * 4 * 64, 4, c00,
                            c01, c10, c11, lane);
      __syncthreads();  // every wave is done reading h2 (dW3) before it is overwritten
      dtanh_inplace(L::H2, wave, lane, c00, c01, c10, c11, gb2a, gb2b);
    }
    __syncthreads();
    // ---- dW2 += dz2^T . h1  (this wave: 64 neurons x 256 inputs, K = 64 rows) ----
    {
      const int ao = opaque(L::H2 + h * FLDH + 64 * wave + r);
      const int bo = opaque(L::H1 + h * FLDH + r);
#pragma unroll 2
      for (int k = 0; k < FR; k += 2) {
        const float* bk = &lds[bo + k * FLDH];
        dw2_kstep(gW2, lds[ao + k * FLDH], lds[ao + k * FLDH + 32], bk[0], bk[32], bk[64], bk[96], bk[128], bk[160], bk[192],
                  bk[224]);
      }
    }
    // ---- dh1 = dz2 . W2 (K = 256), then dz1 = dh1 * (1 - h1^2) in place ----
    {
      f32x16 c00 = zero16(), c01 = zero16(), c10 = zero16(), c11 = zero16();
      constexpr int nkg = FH / 8;
      gemm_lds_packed<FLDH>(L::H2, W.W2b + (size_t)(2 * wave) * nkg * 64, W.W2b + (size_t)(2 * wave + 1) * nkg * 64, nkg,
                            c00, c01, c10, c11, lane);
      __syncthreads();  // dW2 reads of h1 complete everywhere
      dtanh_inplace(L::H1, wave, lane, c00, c01, c10, c11, gb1a, gb1b);
    }
    __syncthreads();
    // ---- dW1 (+)= dz1^T . X  (this wave: 64 neurons x DP inputs, K = 64 rows) ----
    {
      f32x16 t00 = zero16(), t10 = zero16(), t01 = zero16(), t11 = zero16();
      const int ao = opaque(L::H1 + h * FLDH + 64 * wave + r);
      constexpr bool two = DP > 32;
      const int c0 = (r < DP) ? r : 0;
      const int c1 = (32 + r < DP) ? 32 + r : c0;  // clamped columns are never stored
      const int b0o = opaque(L::X + h * ldx + c0), b1o = opaque(L::X + h * ldx + c1);
#pragma unroll 4
      for (int k = 0; k < FR; k += 2) {
        const float x0 = lds[ao + k * FLDH], x1 = lds[ao + k * FLDH + 32];
        const float y0 = lds[b0o + k * ldx];
        t00 = MFMA32(x0, y0, t00);
        t10 = MFMA32(x1, y0, t10);
        if (two) {
          const float y1 = lds[b1o + k * ldx];
          t01 = MFMA32(x0, y1, t01);
          t11 = MFMA32(x1, y1, t11);
        }
      }
      const int so = opaque((64 * wave + 4 * h) * DP + r);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float* p0 = slab_w1 + so + crc(i) * DP;
        float* p1 = p0 + 32 * DP;
        if (r < DP) {
          p0[0] = first ? t00[i] : p0[0] + t00[i];
          p1[0] = first ? t10[i] : p1[0] + t10[i];
        }
        if (two && 32 + r < DP) {
          p0[32] = first ? t01[i] : p0[32] + t01[i];
          p1[32] = first ? t11[i] : p1[32] + t11[i];
        }
      }
    }
    first = false;
    __syncthreads();  // X / h1 / h2 are rewritten by the next tile
  }

  // ---- store this workgroup's partial gradients to its slab ----
  {
    asm volatile("s_nop 15\n\ts_nop 3");  // last asm MFMA's D -> v_accvgpr_read (16-pass XDL)
    float* w2 = slab + slab_off_w2() + (64 * wave + 4 * h) * FH + r;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
#pragma unroll
      for (int i = 0; i < 16; ++i) w2[(32 * (t / 8) + crc(i)) * FH + 32 * (t % 8)] = gW2[t][i];
      __builtin_amdgcn_sched_barrier(0);  // keep at most one tile of read-backs live
    }
    const float b2a = gb2a + __shfl_xor(gb2a, 32, 64), b2b = gb2b + __shfl_xor(gb2b, 32, 64);
    const float b1a = gb1a + __shfl_xor(gb1a, 32, 64), b1b = gb1b + __shfl_xor(gb1b, 32, 64);
    if (h == 0) {
      slab[slab_off_b2(DP) + 64 * wave + r] = b2a;
      slab[slab_off_b2(DP) + 64 * wave + 32 + r] = b2b;
      slab[slab_off_b1(DP) + 64 * wave + r] = b1a;
      slab[slab_off_b1(DP) + 64 * wave + 32 + r] = b1b;
    }
    if (wave == 0 && lane < 32) {
      slab[slab_off_b3(DP) + lane] = gb3;
      slab[slab_off_ls(DP) + lane] = gls;
    }
  }
  if (wave == 0) {
    const float t0 = wave_sum(s_pl), t1 = wave_sum(s_vl), t2 = wave_sum(s_kl), t3 = wave_sum(s_cf);
    if (lane == 0) {
      if (net == 0) {
        atomicAdd(&a.sums[0], t0);
        atomicAdd(&a.sums[2], t2);
        atomicAdd(&a.sums[3], t3);
      } else {
        atomicAdd(&a.sums[1], t1);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Deterministic slab reduction into the canonical gradient vector.
// grid.x covers the P canonical elements; each thread sums its element over the slabs of its network.
// ------------------------------------------------------------------------------------------------
struct SlabReduceArgs {
  const float* slabs; int slab_floats; int nslabs;  // nslabs = gridDim of the train kernel (even: pi, odd: vf)
  float* grads; int P;
  int offs[14];   // canonical offsets
  int D, Dp, A;
  float ent_coef, b_local, inv_bg;
  float* sums;    // sums[4] = rows (for the stats finaliser)
};

__global__ __launch_bounds__(256) void k_slab_reduce(SlabReduceArgs s) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) s.sums[4] = s.b_local;
  if (i >= s.P) return;
  // locate tensor
  int t = 0;
#pragma unroll
  for (int k = 1; k < 13; ++k) t += (i >= s.offs[k]) ? 1 : 0;
  const int e = i - s.offs[t];
  int net, off;
  switch (t) {
    case 0: net = 0; off = slab_off_ls(s.Dp) + e; break;                                   // log_std
    case 1: net = 0; off = slab_off_w1() + (e / s.D) * s.Dp + (e % s.D); break;            // pi W1 [H][D]
    case 2: net = 0; off = slab_off_b1(s.Dp) + e; break;
    case 3: net = 0; off = slab_off_w2() + e; break;
    case 4: net = 0; off = slab_off_b2(s.Dp) + e; break;
    case 5: net = 1; off = slab_off_w1() + (e / s.D) * s.Dp + (e % s.D); break;            // vf W1
    case 6: net = 1; off = slab_off_b1(s.Dp) + e; break;
    case 7: net = 1; off = slab_off_w2() + e; break;
    case 8: net = 1; off = slab_off_b2(s.Dp) + e; break;
    case 9: net = 0; off = slab_off_w3(s.Dp) + e; break;                                   // action_net.weight [A][H]
    case 10: net = 0; off = slab_off_b3(s.Dp) + e; break;
    case 11: net = 1; off = slab_off_w3(s.Dp) + e; break;                                  // value_net.weight [1][H]
    default: net = 1; off = slab_off_b3(s.Dp) + e; break;
  }
  float acc = 0.f;
  for (int w = net; w < s.nslabs; w += 2) acc += s.slabs[(size_t)w * s.slab_floats + off];
  if (t == 0) acc += s.ent_coef * (-s.b_local) * s.inv_bg;  // entropy bonus gradient on log_std
  s.grads[i] = acc;
}

// ------------------------------------------------------------------------------------------------
// Weight packing: canonical row-major W[N][K] (ld) -> MFMA fragment order.
//   fwd pack  Pf[nb][kg][lane][s] = W[nb*32 + r][kg*8 + 4h + s]      (B[k][n] = W[n][k])
//   bwd pack  Pb[jb][kg][lane][s] = W[kg*8 + 4h + s][jb*32 + r]      (B[k][j] = W[k][j])
// rows/cols outside the source matrix are zero.
// ------------------------------------------------------------------------------------------------
__global__ void k_pack_fwd(const float* __restrict__ W, int N, int K, int ld, float* __restrict__ out, int NB,
                           int KG) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= NB * KG * 256) return;
  const int s = i & 3, lane = (i >> 2) & 63, kg = (i >> 8) % KG, nb = (i >> 8) / KG;
  const int n = nb * 32 + (lane & 31), k = kg * 8 + 4 * (lane >> 5) + s;
  out[i] = (n < N && k < K) ? W[(size_t)n * ld + k] : 0.f;
}
__global__ void k_pack_bwd(const float* __restrict__ W, int N, int K, int ld, float* __restrict__ out, int JB,
                           int KG) {
  // W is [N rows = k index][K cols = j index]
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= JB * KG * 256) return;
  const int s = i & 3, lane = (i >> 2) & 63, kg = (i >> 8) % KG, jb = (i >> 8) / KG;
  const int k = kg * 8 + 4 * (lane >> 5) + s, j = jb * 32 + (lane & 31);
  out[i] = (k < N && j < K) ? W[(size_t)k * ld + j] : 0.f;
}

// ------------------------------------------------------------------------------------------------
// Rollout-time fused forward (+ sampling epilogue for the policy network).
// grid = 2 * ceil(rows / 64); block b: net = b & 1, tile = b >> 1.
// ------------------------------------------------------------------------------------------------
struct FusedActArgs {
  FusedNet net[2];
  const float* X; int Dp; int rows;
  int want_pi, want_v;
  float* mu; int ldmu;     // optional raw mean output [rows][ldmu]
  float* v;                // [rows]
  // sampling (policy net) -- any of the outputs may be null
  int sample; int A;
  const float* log_std; const float* eps; uint64_t seed; uint32_t draw; float lo, hi;
  float* act_raw; float* act_clip; float* logp;
};

template <int DP>
__global__ __launch_bounds__(FTHREADS, 1) void k_fused_act(FusedActArgs a) {
  using L = Lay<DP>;
  constexpr int ldx = L::LDX, per = DP / 4;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int net = blockIdx.x & 1, tile = blockIdx.x >> 1;
  if ((net == 0 && !a.want_pi) || (net == 1 && !a.want_v)) return;
  const FusedNet W = a.net[net];
  const int row0 = tile * FR;
#pragma unroll
  for (int i = tid; i < FR * per; i += FTHREADS) {
    const int rr = i / per, c = i - rr * per;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row0 + rr < a.rows) v = reinterpret_cast<const f32x4*>(a.X)[(size_t)(row0 + rr) * per + c];
    *reinterpret_cast<f32x4*>(&lds[L::X + rr * ldx + 4 * c]) = v;
  }
  __syncthreads();
  tile_forward<DP>(W, wave, lane);
  if (wave != 0) return;
  const int row = row0 + lane;
  if (row >= a.rows) return;
  const float* drow = &lds[L::DO + lane * FLDO];
  if (net == 1) {
    a.v[row] = drow[0] + W.b3[0];
    return;
  }
  float lp = 0.f;
  float z0 = 0.f, z1 = 0.f, z2 = 0.f, z3 = 0.f;
  for (int k = 0; k < a.A; ++k) {
    const float m = drow[k] + W.b3[k];
    if (a.mu) a.mu[(size_t)row * a.ldmu + k] = m;
    if (a.sample) {
      float e;
      if (a.eps != nullptr) {
        e = a.eps[(size_t)row * a.A + k];
      } else {
        if ((k & 3) == 0) {
          float z[4];
          box_muller4(philox4x32_10((uint32_t)row, (uint32_t)(k >> 2), a.draw, 0x45505331u, (uint32_t)a.seed,
                                    (uint32_t)(a.seed >> 32)), z);
          z0 = z[0]; z1 = z[1]; z2 = z[2]; z3 = z[3];
        }
        const int q = k & 3;
        e = q == 0 ? z0 : (q == 1 ? z1 : (q == 2 ? z2 : z3));
      }
      const float sd = expf(a.log_std[k]);
      const float act = m + e * sd;
      const float d = act - m;
      lp += -(d * d) / (2.0f * (sd * sd)) - logf(sd) - 0.91893853320467274178f;
      if (a.act_raw) a.act_raw[(size_t)row * a.A + k] = act;
      if (a.act_clip) a.act_clip[(size_t)row * a.A + k] = fminf(fmaxf(act, a.lo), a.hi);
    }
  }
  if (a.sample && a.logp) a.logp[row] = lp;
}

// ------------------------------------------------------------------------------------------------
// host-side state of the fused path
// ------------------------------------------------------------------------------------------------
struct FusedState {
  bool enabled = false;
  int D = 0, Dp = 0, A = 0;
  float* packed = nullptr;      // all packed weights
  size_t packed_floats = 0;
  FusedNet net[2];
  float* slabs = nullptr;
  int slab_floats = 0, max_grid = 0;
  size_t lds_bytes = 0;
};

inline size_t fused_lds_bytes(int Dp) { return (size_t)(FR * (Dp + 4) + 2 * FR * FLDH + FR * FLDO) * sizeof(float); }

inline bool fused_shape_ok(int D, int A, int H1, int H2, int G1, int G2) {
  const int Dp = (D + 7) / 8 * 8;
  return H1 == FH && H2 == FH && G1 == FH && G2 == FH && A <= 32 && (Dp == 16 || Dp == 32 || Dp == 48 || Dp == 64);
}

// (re)build the packed weight copies from the canonical parameter vector
inline void fused_repack(FusedState& f, const float* params, const int* offs, hipStream_t st) {
  if (!f.enabled) return;
  const int Dp = f.Dp, D = f.D, A = f.A;
  auto fwd = [&](const float* W, int N, int K, int ld, const f32x4* out, int NB, int KG) {
    hipLaunchKernelGGL(k_pack_fwd, dim3((NB * KG * 256 + 255) / 256), dim3(256), 0, st, W, N, K, ld, (float*)out, NB, KG);
  };
  auto bwd = [&](const float* W, int N, int K, int ld, const f32x4* out, int JB, int KG) {
    hipLaunchKernelGGL(k_pack_bwd, dim3((JB * KG * 256 + 255) / 256), dim3(256), 0, st, W, N, K, ld, (float*)out, JB, KG);
  };
  // tensor ids: 0 log_std, 1 pW1, 2 pb1, 3 pW2, 4 pb2, 5 vW1, 6 vb1, 7 vW2, 8 vb2, 9 aW, 10 ab, 11 vW, 12 vb
  const int w1[2] = {1, 5}, w2[2] = {3, 7}, w3[2] = {9, 11}, heads[2] = {A, 1};
  for (int n = 0; n < 2; ++n) {
    fwd(params + offs[w1[n]], FH, D, D, f.net[n].W1f, FH / 32, Dp / 8);
    fwd(params + offs[w2[n]], FH, FH, FH, f.net[n].W2f, FH / 32, FH / 8);
    fwd(params + offs[w3[n]], heads[n], FH, FH, f.net[n].W3f, 1, FH / 8);
    bwd(params + offs[w2[n]], FH, FH, FH, f.net[n].W2b, FH / 32, FH / 8);
    bwd(params + offs[w3[n]], heads[n], FH, FH, f.net[n].W3b, FH / 32, 4);
  }
}

#define FUSED_DISPATCH_DP(dp, CALL)                 \
  switch (dp) {                                     \
    case 16: { constexpr int DPc = 16; CALL; } break; \
    case 32: { constexpr int DPc = 32; CALL; } break; \
    case 48: { constexpr int DPc = 48; CALL; } break; \
    default: { constexpr int DPc = 64; CALL; } break; \
  }

inline void fused_launch_act(FusedState& f, FusedActArgs& a, hipStream_t st) {
  a.net[0] = f.net[0];
  a.net[1] = f.net[1];
  a.Dp = f.Dp;
  const int tiles = (a.rows + FR - 1) / FR;
  FUSED_DISPATCH_DP(f.Dp, hipLaunchKernelGGL((k_fused_act<DPc>), dim3(2 * tiles), dim3(FTHREADS), f.lds_bytes, st, a));
}
inline void fused_launch_train(FusedState& f, FusedTrainArgs& a, int grid, hipStream_t st) {
  FUSED_DISPATCH_DP(f.Dp, hipLaunchKernelGGL((k_fused_train<DPc>), dim3(grid), dim3(FTHREADS), f.lds_bytes, st, a));
}
inline hipError_t fused_set_lds_attr(FusedState& f) {
  hipError_t e = hipSuccess;
  FUSED_DISPATCH_DP(f.Dp, {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_fused_train<DPc>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)f.lds_bytes);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_fused_act<DPc>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)f.lds_bytes);
  });
  return e;
}

inline bool fused_forward(FusedState& f, const float* X, int rows, bool want_pi, float* mu_out, int ldmu, bool want_v,
                          float* v_out, hipStream_t st) {
  if (!f.enabled) return false;
  FusedActArgs a{};
  a.X = X; a.rows = rows; a.want_pi = want_pi; a.want_v = want_v; a.mu = mu_out; a.ldmu = ldmu; a.v = v_out;
  a.sample = 0; a.A = f.A;
  fused_launch_act(f, a, st);
  return true;
}

}  // namespace mobrob
