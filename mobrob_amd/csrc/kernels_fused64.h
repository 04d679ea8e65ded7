// Fused fast path for hidden width 64 -- the shape of every reference config (data/configs/*.yaml: net_arch
// pi=[64,64], vf=[64,64]; SB3 default) and of BASELINE config 2 (point, 1024 envs, 2x64).
//
// With 64-wide layers a whole network tile fits one wave: every wave owns a private 32-row tile in LDS and runs
// the complete forward -> loss -> backward chain by itself.  There is NO workgroup barrier in the tile loop: the
// LDS round trips between layers (C layout -> A-operand layout) are ordered by the in-order DS queue of the
// wave.  Weights (<= 64 KB per network, fragment-packed as in kernels_fused.h) stream from L2; all weight
// gradients (dW2 4 tiles, dW1 <= 4 tiles, dW3 2 tiles = 160 registers) stay in registers for every tile of the
// wave and are written once, as a per-wave slab, for the deterministic reduction kernel.
#pragma once
#include "kernels_fused.h"

namespace mobrob {

constexpr int GH = 64;         // hidden width
constexpr int GR = 32;         // rows per tile = one MFMA row block
constexpr int GLDH = GH + 4;   // 68 floats: conflict-free ds_read_b128 (68 % 64 == 4)
// independent waves per block: as many as fit 160 KB of LDS -> 5 or 6 waves per CU, i.e. two waves on some SIMDs
// (each wave needs <= 256 registers), which hides most of the phase-boundary latency of a lone wave
__host__ __device__ constexpr int g_waves(int DP) { return DP <= 32 ? 6 : 5; }

// packed weights of ONE network as laid out in FusedState::packed (and mirrored into LDS by the train kernel)
template <int DP>
struct Wts64 {
  static constexpr int W1F = 0;
  static constexpr int W2F = W1F + 2 * (DP / 8) * 256;
  static constexpr int W3F = W2F + 2 * 8 * 256;
  static constexpr int W2B = W3F + 8 * 256;
  static constexpr int W3B = W2B + 2 * 8 * 256;
  static constexpr int B1S = W3B + 2 * 4 * 256;
  static constexpr int B2S = B1S + 64;
  static constexpr int TOTAL = B2S + 64;
};
template <int DP>
struct Lay64 {
  static constexpr int LDX = DP + 4;
  static constexpr int X = 0;
  static constexpr int H1 = X + GR * LDX;
  static constexpr int H2 = H1 + GR * GLDH;
  static constexpr int DO = H2 + GR * GLDH;    // head tile [32][FLDO]
  static constexpr int GACC = DO + GR * FLDO;  // [2][32] head-bias / log_std gradient sums of this wave
  static constexpr int WAVE = GACC + 64;       // floats per wave
  static constexpr int NWV = g_waves(DP);
  static constexpr int CST = NWV * WAVE;       // block-level [3][32] per-action constants
  static constexpr int END = CST + 96;
  // training kernel: the network's packed weights are LDS resident in front of the wave regions
  static constexpr int TW = Wts64<DP>::TOTAL;
  static constexpr int TNWV = (40960 - TW - 96) / WAVE;   // waves per block that fit 160 KB (4 for DP=16, else 3)
  static constexpr int TCST = TW + TNWV * WAVE;
  static constexpr int TEND = TCST + 96;
};
__host__ __device__ constexpr int g_train_waves(int DP) {
  return (40960 - (2 * (DP / 8) * 256 + 2 * 4096 + 2048 + 4096 / 2 + 128) - 96) / (GR * (DP + 4) + 2 * GR * GLDH + GR * FLDO + 64);
}
static_assert(g_train_waves(16) <= 4 && g_train_waves(64) >= 1, "k_slab64_reduce folds groups of at most four tiles");
inline size_t fused64_train_lds_bytes(int Dp) {
  const int tw = 2 * (Dp / 8) * 256 + 2 * 4096 + 2048 + 2048 + 128;
  return (size_t)(tw + g_train_waves(Dp) * (GR * (Dp + 4) + 2 * GR * GLDH + GR * FLDO + 64) + 96) * sizeof(float);
}
inline size_t fused64_lds_bytes(int Dp) {
  return (size_t)(g_waves(Dp) * (GR * (Dp + 4) + 2 * GR * GLDH + GR * FLDO + 64) + 96) * sizeof(float);
}

// per-BLOCK slab (floats; the waves of a block are summed through LDS at kernel end), fragment order: dW2 [4 tiles: ib*2+jb] | dW1 [4 tiles: ib*2+jb] | dW3 [2 tiles: jb]
//                                         | db2 [64] | db1 [64] | db3 [32] | dls [32]
__host__ __device__ inline int s64_w2() { return 0; }
__host__ __device__ inline int s64_w1() { return 4 * 1024; }
__host__ __device__ inline int s64_w3() { return 8 * 1024; }
__host__ __device__ inline int s64_b2() { return 10 * 1024; }
__host__ __device__ inline int s64_b1() { return s64_b2() + 64; }
__host__ __device__ inline int s64_b3() { return s64_b1() + 64; }
__host__ __device__ inline int s64_ls() { return s64_b3() + 32; }
__host__ __device__ inline int s64_st() { return s64_ls() + 32; }  // loss sums: pl, vl, kl, clip count
__host__ __device__ inline int s64_size() { return s64_st() + 8; }

// 32-row GEMM with BOTH operands in LDS (weights resident): c{0,1} += A[32 x 8*nkg] . Bpacked{0,1}
template <int LDA>
__device__ __forceinline__ void gemm_lds_lds_r32(int a_off, int b_off0, int b_off1, int nkg, f32x16& c0, f32x16& c1,
                                                 int lane) {
  const int r = lane & 31, h = lane >> 5;
  const int ab = 4 * opaque((a_off + r * LDA + 4 * h) >> 2);
  const int b0 = 4 * opaque((b_off0 >> 2) + lane), b1 = 4 * opaque((b_off1 >> 2) + lane);
  f32x4 uA = *reinterpret_cast<const f32x4*>(&lds[ab]);
  f32x4 pA = *reinterpret_cast<const f32x4*>(&lds[b0]), qA = *reinterpret_cast<const f32x4*>(&lds[b1]);
  f32x4 uB, pB, qB;
  int ao = ab, bo = 0;
#pragma unroll 1
  for (int kg = 0; kg < nkg - 2; kg += 2) {
    uB = *reinterpret_cast<const f32x4*>(&lds[ao + 8]);
    pB = *reinterpret_cast<const f32x4*>(&lds[b0 + bo + 256]);
    qB = *reinterpret_cast<const f32x4*>(&lds[b1 + bo + 256]);
    MFMA_KG1(uA, pA, qA)
    uA = *reinterpret_cast<const f32x4*>(&lds[ao + 16]);
    pA = *reinterpret_cast<const f32x4*>(&lds[b0 + bo + 512]);
    qA = *reinterpret_cast<const f32x4*>(&lds[b1 + bo + 512]);
    MFMA_KG1(uB, pB, qB)
    ao += 16;
    bo += 512;
  }
  uB = *reinterpret_cast<const f32x4*>(&lds[ao + 8]);
  pB = *reinterpret_cast<const f32x4*>(&lds[b0 + bo + 256]);
  qB = *reinterpret_cast<const f32x4*>(&lds[b1 + bo + 256]);
  MFMA_KG1(uA, pA, qA)
  MFMA_KG1(uB, pB, qB)
}

// forward layers of one 32-row tile held at LDS offset wb (per-wave region); leaves h1, h2 and the raw head tile
template <int DP>
__device__ __forceinline__ void tile64_forward(const FusedNet& W, int wb, int lane) {
  using L = Lay64<DP>;
  const int r = lane & 31, h = lane >> 5;
  {  // layer 1
    f32x16 c0 = splat16(W.b1s[r]), c1 = splat16(W.b1s[32 + r]);
    constexpr int nkg = DP / 8;
    gemm_lds_packed_r32<L::LDX>(wb + L::X, W.W1f, W.W1f + (size_t)nkg * 64, nkg, c0, c1, lane);
    const int o = opaque(wb + L::H1 + 4 * h * GLDH + r);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      lds[o + crc(i) * GLDH] = fast_tanh_scaled(c0[i]);
      lds[o + crc(i) * GLDH + 32] = fast_tanh_scaled(c1[i]);
    }
  }
  {  // layer 2
    f32x16 c0 = splat16(W.b2s[r]), c1 = splat16(W.b2s[32 + r]);
    constexpr int nkg = GH / 8;
    gemm_lds_packed_r32<GLDH>(wb + L::H1, W.W2f, W.W2f + (size_t)nkg * 64, nkg, c0, c1, lane);
    const int o = opaque(wb + L::H2 + 4 * h * GLDH + r);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      lds[o + crc(i) * GLDH] = fast_tanh_scaled(c0[i]);
      lds[o + crc(i) * GLDH + 32] = fast_tanh_scaled(c1[i]);
    }
  }
  {  // head: [32 x 32] = h2 . W3^T, K = 64, two independent accumulation chains
    f32x16 acc = zero16(), acc2 = zero16();
    const int ab = 4 * opaque((wb + L::H2 + r * GLDH + 4 * h) >> 2);
    const unsigned bo = opaque_u((unsigned)lane * 16u);
#pragma unroll
    for (int kg = 0; kg < 8; kg += 2) {
      const f32x4 b0 = ldg16(W.W3f, bo + kg * 1024u), b1 = ldg16(W.W3f, bo + (kg + 1) * 1024u);
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(&lds[ab + kg * 8]);
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(&lds[ab + kg * 8 + 8]);
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) {
        acc = MFMA32(a0[s_], b0[s_], acc);
        acc2 = MFMA32(a1[s_], b1[s_], acc2);
      }
    }
    const int o = opaque(wb + L::DO + 4 * h * FLDO + r);
#pragma unroll
    for (int i = 0; i < 16; ++i) lds[o + crc(i) * FLDO] = acc[i] + acc2[i];
  }
}

// the same forward with the network's packed weights resident in LDS at offset 0 (training kernel)
template <int DP, class Wt = Wts64<DP>>
__device__ __forceinline__ void tile64_forward_ldsw(int wb, int lane) {
  using L = Lay64<DP>;
  const int r = lane & 31, h = lane >> 5;
  {
    f32x16 c0 = splat16(lds[Wt::B1S + r]), c1 = splat16(lds[Wt::B1S + 32 + r]);
    constexpr int nkg = DP / 8;
    gemm_lds_lds_r32<L::LDX>(wb + L::X, Wt::W1F, Wt::W1F + nkg * 256, nkg, c0, c1, lane);
    const int o = opaque(wb + L::H1 + 4 * h * GLDH + r);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      lds[o + crc(i) * GLDH] = fast_tanh_scaled(c0[i]);
      lds[o + crc(i) * GLDH + 32] = fast_tanh_scaled(c1[i]);
    }
  }
  {
    f32x16 c0 = splat16(lds[Wt::B2S + r]), c1 = splat16(lds[Wt::B2S + 32 + r]);
    gemm_lds_lds_r32<GLDH>(wb + L::H1, Wt::W2F, Wt::W2F + 8 * 256, 8, c0, c1, lane);
    const int o = opaque(wb + L::H2 + 4 * h * GLDH + r);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      lds[o + crc(i) * GLDH] = fast_tanh_scaled(c0[i]);
      lds[o + crc(i) * GLDH + 32] = fast_tanh_scaled(c1[i]);
    }
  }
  {
    f32x16 acc = zero16(), acc2 = zero16();
    const int ab = 4 * opaque((wb + L::H2 + r * GLDH + 4 * h) >> 2);
    const int bb = 4 * opaque((Wt::W3F >> 2) + lane);
#pragma unroll
    for (int kg = 0; kg < 8; kg += 2) {
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(&lds[bb + kg * 256]);
      const f32x4 b1 = *reinterpret_cast<const f32x4*>(&lds[bb + (kg + 1) * 256]);
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(&lds[ab + kg * 8]);
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(&lds[ab + kg * 8 + 8]);
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) {
        acc = MFMA32(a0[s_], b0[s_], acc);
        acc2 = MFMA32(a1[s_], b1[s_], acc2);
      }
    }
    const int o = opaque(wb + L::DO + 4 * h * FLDO + r);
#pragma unroll
    for (int i = 0; i < 16; ++i) lds[o + crc(i) * FLDO] = acc[i] + acc2[i];
  }
}

struct Fused64TrainArgs {
  FusedNet net[2];
  const float* wpack[2];  // per-network packed weight block (Wts64 layout), mirrored into LDS
  const float* obs;
  const float* actions; int A;
  const float* old_logp; const float* adv; const float* ret;
  const int* rows; int count;
  const float* log_std;
  const double* advstat;
  int normalize;
  float clip, vf_coef, ent_coef, inv_bg;
  float clip_vf; const float* old_values;  // clip_range_vf (< 0: none) and the rollout's value predictions
  float* slabs;   // [gridDim.x][s64_size()]
  float* sums;
  unsigned long long* stamps;  // diagnostic builds (-DMOBROB_PAIR_STAMPS): per-phase cycle sums
};

// Per-wave gradient accumulators of one network (registers): dW2 4 tiles [ib][jb] = 00, 10, 01, 11; dW1 <= 4 tiles;
// dW3 2 tiles; bias gradients of hidden column `lane`; loss sums.
struct Grad64 {
  f32x16 W2a, W2b, W2c, W2d, W1a, W1b, W1c, W1d, W3a, W3b;
  float b2, b1, pl, vl, kl, cf;
  __device__ __forceinline__ void zero() {
    W2a = W2b = W2c = W2d = W1a = W1b = W1c = W1d = W3a = W3b = zero16();
    b2 = b1 = pl = vl = kl = cf = 0.f;
  }
};

// One 32-row tile of one network by ONE wave: rows -> forward -> loss -> backward, gradients added to `g`.
//   WL         the network's packed weights are LDS resident at offset 0 (Wts64 layout; k_fused64_train) | streamed
//              from L2 through W's fragment packs (k_train_small: the packs are rewritten by Adam every step)
//   wb / cst   LDS offsets of this wave's tile region and of the block's [3][32] per-action constants
//   rows, cnt  permuted row indices and row count of the minibatch this tile belongs to; the tile starts at row0
//   xr         observation rows of THIS tile (fetched earlier); on return it holds those of the wave's next tile:
//              rows nrows[nrow0 ...] of a minibatch of ncnt rows (nrows == nullptr: none)
template <int DP, bool WL>
__device__ __forceinline__ void tile64_train(const Fused64TrainArgs& a, const FusedNet& W, int net, int wb, int cst, int lane0,
                                             const int* rows, int cnt, int row0, const int* nrows, int ncnt, int nrow0,
                                             f32x4 (&xr)[GR * (DP / 4) / 64], float adv_mean, float adv_sd, bool adv_on,
                                             Grad64& g) {
  using L = Lay64<DP>;
  using Wt = Wts64<DP>;
  constexpr int ldx = L::LDX, per = DP / 4;
  constexpr int NGW = GR * per / 64;
  int nsrc[NGW];
  const int lane = opaque(lane0) & 63;
  const int r = lane & 31, h = lane >> 5;
  // ---- the observation rows of this tile were fetched during the previous tile ----
#pragma unroll
  for (int u = 0; u < NGW; ++u) {
    const int i = lane + u * 64, rr = i / per, c = i - rr * per;
    *reinterpret_cast<f32x4*>(&lds[wb + L::X + rr * ldx + 4 * c]) = xr[u];
  }
#pragma unroll
  for (int u = 0; u < NGW; ++u) {
    const int rr = (lane + u * 64) / per;
    nsrc[u] = (nrows != nullptr && nrow0 + rr < ncnt) ? nrows[nrow0 + rr] : -1;
  }
  // operands of the loss stage (two lanes per row), in flight while the forward pass runs
  const bool llive = row0 + r < cnt;
  float l_adv = 0.f, l_old = 0.f, l_act[16];
  {
    const unsigned src = llive ? (unsigned)rows[row0 + r] : 0u;
    if (net == 0) {
      const float* arow = a.actions + (size_t)src * a.A + h;
#pragma unroll
      for (int j = 0; j < 16; ++j) l_act[j] = (2 * j + h < a.A && llive) ? arow[2 * j] : 0.f;
      if (llive) { l_adv = a.adv[src]; l_old = a.old_logp[src]; }
    } else {
#pragma unroll
      for (int j = 0; j < 16; ++j) l_act[j] = 0.f;
      if (llive) {
        l_old = a.ret[src];
        if (a.clip_vf >= 0.f) l_adv = a.old_values[src];
      }
    }
  }
  if constexpr (WL) tile64_forward_ldsw<DP>(wb, lane);
  else tile64_forward<DP>(W, wb, lane);

  // ---- loss: two lanes per row (q = action parity); dL/d(head) -> head tile, zero padded ----
  {
    const int rr = r, q = h;
    const bool live = llive;
    const int db = opaque(wb + L::DO + rr * FLDO + q);
    const int cb = opaque(cst + q);
    const int gb = opaque(wb + L::GACC + q);
    const int A = a.A;
    if (net == 0) {
      float lp = 0.f;
      float dk[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        float d = 0.f;
        if (2 * j + q < A && live) {
          d = l_act[j] - (lds[db + 2 * j] + lds[cb + 64 + 2 * j]);
          lp += -(d * d) * (0.5f * lds[cb + 2 * j]) - lds[cb + 32 + 2 * j];
        }
        dk[j] = d;
      }
      lp += __shfl_xor(lp, 32, 64);
      float g_logp = 0.f;
      if (live) {
        float adv = l_adv;
        if (a.normalize && adv_on) adv = (adv - adv_mean) / (adv_sd + 1e-8f);
        const float log_ratio = lp - l_old;
        const float ratio = expf(log_ratio);
        const float lo = 1.0f - a.clip, hi = 1.0f + a.clip;
        const float s1 = adv * ratio, s2 = adv * fminf(fmaxf(ratio, lo), hi);
        if (q == 0) {
          g.pl += fminf(s1, s2);
          g.cf += (fabsf(ratio - 1.0f) > a.clip) ? 1.f : 0.f;
          g.kl += (ratio - 1.0f) - log_ratio;
        }
        const float in_range = (ratio >= lo && ratio <= hi) ? 1.f : 0.f;
        const float w1 = (s1 < s2) ? 1.f : ((s1 > s2) ? 0.f : 0.5f);
        g_logp = -(w1 * adv + (1.0f - w1) * adv * in_range) * a.inv_bg * ratio;
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {  // k = 2j + q; wave-uniform trip count
        const int k = 2 * j + q;
        float gm = 0.f, gl = 0.f;
        if (k < A && live) {
          const float iv = lds[cb + 2 * j];
          const float d = dk[j];
          gm = g_logp * d * iv;
          gl = g_logp * (d * d * iv - 1.0f);
        }
        lds[db + 2 * j] = gm;
        if (2 * j < A) {  // wave-uniform: sum over the 32 rows (lanes with equal q)
#pragma unroll
          for (int o = 1; o < 32; o <<= 1) {
            gm += xor_lane(gm, o);
            gl += xor_lane(gl, o);
          }
          if (r == 0) {
            lds[gb + 2 * j] += gm;
            lds[gb + 32 + 2 * j] += gl;
          }
        }
      }
    } else {
      float dv = 0.f;
      if (live && q == 0) {
        float sq, gv_;
        value_loss_terms(lds[db] + lds[cb + 64], l_old, l_adv, a.clip_vf, sq, gv_);
        g.vl += sq;
        dv = a.vf_coef * gv_ * a.inv_bg;
      }
      for (int j = 0; j < 16; ++j) lds[db + 2 * j] = (j == 0) ? dv : 0.f;
      const float t = wave_sum(dv);
      if (lane == 0) lds[gb] += t;
    }
  }

  // ---- dW3 += dout^T . h2  (K = 32 rows) ----
  {
    const int ao = opaque(wb + L::DO + h * FLDO + r);
    const int bo = opaque(wb + L::H2 + h * GLDH + r);
#pragma unroll 4
    for (int k = 0; k < GR; k += 2)
      mfma_x1y2(g.W3a, g.W3b, lds[ao + k * FLDO], lds[bo + k * GLDH], lds[bo + k * GLDH + 32]);
  }
  // ---- dh2 = dout . W3 (K = 32) ; dz2 = dh2 * (1 - h2^2) in place ----
  {
    f32x16 c0 = zero16(), c1 = zero16();
    const int nkh = W.head <= 16 ? 2 : 4;  // k-groups of 8 head columns; those beyond `head` are zero
    if constexpr (WL) gemm_lds_lds_r32<FLDO>(wb + L::DO, Wt::W3B, Wt::W3B + 4 * 256, nkh, c0, c1, lane);
    else gemm_lds_packed_r32<FLDO>(wb + L::DO, W.W3b, W.W3b + 4 * 64, nkh, c0, c1, lane);
    const int o = opaque(wb + L::H2 + 4 * h * GLDH + r);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float hv;
      hv = lds[o + crc(i) * GLDH];      lds[o + crc(i) * GLDH] = c0[i] * (1.0f - hv * hv);
      hv = lds[o + crc(i) * GLDH + 32]; lds[o + crc(i) * GLDH + 32] = c1[i] * (1.0f - hv * hv);
    }
  }
  // ---- bias gradient of layer 2: column sums of dz2 (lane c <-> column c) ----
  {
    const int o = opaque(wb + L::H2 + lane);
    float s0 = 0.f, s1 = 0.f;
#pragma unroll 4
    for (int rr = 0; rr < GR; rr += 2) {
      s0 += lds[o + rr * GLDH];
      s1 += lds[o + (rr + 1) * GLDH];
    }
    g.b2 += s0 + s1;
  }
#pragma unroll
  for (int u = 0; u < NGW; ++u) {  // next tile's observation rows: in flight during dW2 / dh1 / dW1
    const int c = (lane + u * 64) % per;
    xr[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (nsrc[u] >= 0) xr[u] = ldg16(a.obs, (unsigned)nsrc[u] * (unsigned)(DP * 4) + (unsigned)(c * 16));
  }
  // ---- dW2 += dz2^T . h1  (64 x 64, K = 32 rows) ----
  {
    const int ao = opaque(wb + L::H2 + h * GLDH + r);
    const int bo = opaque(wb + L::H1 + h * GLDH + r);
#pragma unroll 4
    for (int k = 0; k < GR; k += 2)
      mfma_x2y2(g.W2a, g.W2b, g.W2c, g.W2d, lds[ao + k * GLDH], lds[ao + k * GLDH + 32], lds[bo + k * GLDH],
                lds[bo + k * GLDH + 32]);
  }
  // ---- dh1 = dz2 . W2 (K = 64) ; dz1 = dh1 * (1 - h1^2) in place ----
  {
    f32x16 c0 = zero16(), c1 = zero16();
    constexpr int nkg = GH / 8;
    if constexpr (WL) gemm_lds_lds_r32<GLDH>(wb + L::H2, Wt::W2B, Wt::W2B + nkg * 256, nkg, c0, c1, lane);
    else gemm_lds_packed_r32<GLDH>(wb + L::H2, W.W2b, W.W2b + nkg * 64, nkg, c0, c1, lane);
    const int o = opaque(wb + L::H1 + 4 * h * GLDH + r);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float hv;
      hv = lds[o + crc(i) * GLDH];      lds[o + crc(i) * GLDH] = c0[i] * (1.0f - hv * hv);
      hv = lds[o + crc(i) * GLDH + 32]; lds[o + crc(i) * GLDH + 32] = c1[i] * (1.0f - hv * hv);
    }
  }
  {
    const int o = opaque(wb + L::H1 + lane);
    float s0 = 0.f, s1 = 0.f;
#pragma unroll 4
    for (int rr = 0; rr < GR; rr += 2) {
      s0 += lds[o + rr * GLDH];
      s1 += lds[o + (rr + 1) * GLDH];
    }
    g.b1 += s0 + s1;
  }
  // ---- dW1 += dz1^T . X  (64 x DP, K = 32 rows) ----
  {
    constexpr bool two = DP > 32;
    const int ao = opaque(wb + L::H1 + h * GLDH + r);
    const int c0 = (r < DP) ? r : 0;
    const int c1 = (32 + r < DP) ? 32 + r : c0;
    const int b0o = opaque(wb + L::X + h * ldx + c0), b1o = opaque(wb + L::X + h * ldx + c1);
#pragma unroll 4
    for (int k = 0; k < GR; k += 2) {
      if (two)
        mfma_x2y2(g.W1a, g.W1b, g.W1c, g.W1d, lds[ao + k * GLDH], lds[ao + k * GLDH + 32], lds[b0o + k * ldx],
                  lds[b1o + k * ldx]);
      else
        mfma_x2y1(g.W1a, g.W1b, lds[ao + k * GLDH], lds[ao + k * GLDH + 32], lds[b0o + k * ldx]);
    }
  }
}

// grid: even number of blocks; block b works for network b & 1; its NWV waves take tiles (b>>1)*NWV + wave, stride.
template <int DP>
__global__ __launch_bounds__(Lay64<DP>::TNWV * 64, 1) void k_fused64_train(Fused64TrainArgs a) {
  using L = Lay64<DP>;
  using Wt = Wts64<DP>;
  constexpr int ldx = L::LDX, per = DP / 4, NWV = L::TNWV;
  const int tid0 = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int net = blockIdx.x & 1;
  const int widx = (blockIdx.x >> 1) * NWV + wave, nw = (gridDim.x >> 1) * NWV;
  const int wb = L::TW + wave * L::WAVE;
  const FusedNet W = a.net[net];
  const int ntiles = (a.count + GR - 1) / GR;
  // mirror this network's packed weights (fragment order, scaled biases) into LDS: every GEMM operand of the
  // tile loop then comes from LDS and no phase waits on L2
  {  // eight 16-byte loads in flight per thread (one load -> one LDS store at a time is a chain of L2 round trips:
     // 22 of them for the 66 KB of a 58-dim network)
    constexpr int NV = Wt::TOTAL / 4, NT = NWV * 64;
    const f32x4* src = reinterpret_cast<const f32x4*>(a.wpack[net]);
    for (int base = tid0; base < NV; base += 8 * NT) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = base + u * NT;
        v[u] = i < NV ? src[i] : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = base + u * NT;
        if (i < NV) reinterpret_cast<f32x4*>(lds)[i] = v[u];
      }
    }
  }

  Grad64 g;
  g.zero();

  if (tid0 < 32) {  // per-action constants (block level)
    const int k = tid0;
    float iv = 0.f, lc = 0.f, bb = 0.f;
    if (net == 0 && k < a.A) {
      const float sd = expf(a.log_std[k]);
      iv = 1.0f / (sd * sd);
      lc = logf(sd) + 0.91893853320467274178f;
    }
    if (k < W.head) bb = W.b3[k];
    lds[L::TCST + k] = iv;
    lds[L::TCST + 32 + k] = lc;
    lds[L::TCST + 64 + k] = bb;
  }
  lds[wb + L::GACC + (tid0 & 63)] = 0.f;
  __syncthreads();  // the only workgroup barrier: constants visible to the four waves

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

  // software-pipelined gather (one wave has nothing else to hide global latency with): xr holds the observation
  // rows of the tile about to be processed, nsrc the row indices of the tile after it
  constexpr int NGW = GR * per / 64;
  static_assert((GR * per) % 64 == 0, "gather assumes a whole number of 16-byte chunks per lane");
  f32x4 xr[NGW];
#pragma unroll
  for (int u = 0; u < NGW; ++u) {
    const int i = (tid0 & 63) + u * 64, rr = i / per, c = i - rr * per;
    xr[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (widx < ntiles && widx * GR + rr < a.count)
      xr[u] = ldg16(a.obs, (unsigned)a.rows[widx * GR + rr] * (unsigned)(DP * 4) + (unsigned)(c * 16));
  }
#ifdef MOBROB64_EMPTY
  for (int tile = widx; tile < 0; tile += nw) {
#else
  for (int tile = widx; tile < ntiles; tile += nw) {
#endif
    tile64_train<DP, true>(a, W, net, wb, L::TCST, tid0, a.rows, a.count, tile * GR,
                           tile + nw < ntiles ? a.rows : nullptr, a.count, (tile + nw) * GR, xr, adv_mean, adv_sd, adv_on, g);
  }

  // ---- sum the block's waves through LDS (fixed order), then wave 0 writes the block slab ----
  const int lane = tid0 & 63;
  asm volatile("s_nop 15\n\ts_nop 3");  // last asm MFMA's D -> VALU read
  __syncthreads();                          // every wave has left its tile loop: the tile regions are free
  // Two rounds of five accumulator tiles: waves 1.. stage theirs in LDS ([wave-1][slot][16][64] floats; the tile
  // regions and the weight mirror are dead by now), wave 0 adds them to its own in wave order.  (Ten separate
  // store / barrier / add / barrier rounds, one per tile, cost 20 us of a 31 us launch at small minibatches.)
  auto stage = [&](int slot, const f32x16& acc) {
    if (wave > 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) lds[((wave - 1) * 5 + slot) * 1024 + i * 64 + lane] = acc[i];
    }
  };
  auto fold = [&](int slot, f32x16& acc) {
    if (wave == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float x[NWV - 1];
#pragma unroll
        for (int w = 0; w < NWV - 1; ++w) x[w] = lds[(w * 5 + slot) * 1024 + i * 64 + lane];
        float v = acc[i];
#pragma unroll
        for (int w = 0; w < NWV - 1; ++w) v += x[w];
        acc[i] = v;
      }
    }
  };
  // head-bias / log_std sums of all waves: read before the staging overwrites the GACC regions
  float b3s = 0.f, lss = 0.f;
  if (wave == 0 && lane < 32) {
#pragma unroll
    for (int w = 0; w < NWV; ++w) {
      b3s += lds[L::TW + w * L::WAVE + L::GACC + lane];
      lss += lds[L::TW + w * L::WAVE + L::GACC + 32 + lane];
    }
  }
  __syncthreads();
  stage(0, g.W2a); stage(1, g.W2b); stage(2, g.W2c); stage(3, g.W2d); stage(4, g.W1a);
  __syncthreads();
  fold(0, g.W2a); fold(1, g.W2b); fold(2, g.W2c); fold(3, g.W2d); fold(4, g.W1a);
  __syncthreads();
  stage(0, g.W1b); stage(1, g.W3a); stage(2, g.W3b);
  if (DP > 32) { stage(3, g.W1c); stage(4, g.W1d); }
  __syncthreads();
  fold(0, g.W1b); fold(1, g.W3a); fold(2, g.W3b);
  if (DP > 32) { fold(3, g.W1c); fold(4, g.W1d); }
  __syncthreads();
  {
    if (wave > 0) {
      lds[(wave - 1) * 128 + lane] = g.b2;
      lds[(wave - 1) * 128 + 64 + lane] = g.b1;
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int w = 0; w < NWV - 1; ++w) {
        g.b2 += lds[w * 128 + lane];
        g.b1 += lds[w * 128 + 64 + lane];
      }
    }
  }
  {  // loss statistics through the slab (fixed order; no contended atomics)
    const float t0 = wave_sum(g.pl), t1 = wave_sum(g.vl), t2 = wave_sum(g.kl), t3 = wave_sum(g.cf);
    __syncthreads();  // the bias partials above have been consumed
    if (lane == 0) {
      lds[wave * 4 + 0] = t0; lds[wave * 4 + 1] = t1; lds[wave * 4 + 2] = t2; lds[wave * 4 + 3] = t3;
    }
    __syncthreads();
  }
  if (wave != 0) return;
  float* slab = a.slabs + (size_t)blockIdx.x * s64_size();
  if (lane < 4) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NWV; ++w) t += lds[w * 4 + lane];
    slab[s64_st() + lane] = t;
  }
  auto put = [&](int region, int t, const f32x16& acc) {
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = acc[4 * qd + e];
      stg16(slab + region, (unsigned)((t * 4 + qd) * 64 + lane) * 16u, v);
    }
  };
  put(s64_w2(), 0, g.W2a); put(s64_w2(), 2, g.W2b); put(s64_w2(), 1, g.W2c); put(s64_w2(), 3, g.W2d);  // t = ib*2 + jb
  put(s64_w1(), 0, g.W1a); put(s64_w1(), 2, g.W1b);
  if (DP > 32) { put(s64_w1(), 1, g.W1c); put(s64_w1(), 3, g.W1d); }
  put(s64_w3(), 0, g.W3a); put(s64_w3(), 1, g.W3b);
  slab[s64_b2() + lane] = g.b2;
  slab[s64_b1() + lane] = g.b1;
  if (lane < 32) {  // head-bias / log_std sums of all waves (summed before the reduction reused their LDS)
    slab[s64_b3() + lane] = b3s;
    slab[s64_ls() + lane] = lss;
  }
}

// ---- slab reduction for the 64-wide path: thread p sums slab position p over the waves of its network ----
struct Slab64ReduceArgs {
  const float* slabs; int nblocks;  // train grid size; slab index = block, network = block & 1
  int group;                        // <= 1: one slab per block of k_fused64_train.  G > 1: one slab per TILE
                                    // (k_split64_train); G consecutive tiles are folded first, in order -- the wave
                                    // fold of a k_fused64_train block -- and the folds are then summed like block slabs
  float* grads; int P;
  int offs[14];
  int D, A;
  float ent_coef, b_local, inv_bg;
  float* sums;
  double* rec_sum; int* rec_t;  // optional norm records (block_norm_records, kernels_fused.h)
};
template <class TS>
__host__ __device__ __forceinline__ int slab64_to_canonical(const TS& s, int net, int p) {
  const int T_W1 = net == 0 ? 1 : 5, T_B1 = net == 0 ? 2 : 6, T_W2 = net == 0 ? 3 : 7, T_B2 = net == 0 ? 4 : 8;
  const int T_W3 = net == 0 ? 9 : 11, T_B3 = net == 0 ? 10 : 12;
  const int head = net == 0 ? s.A : 1;
  auto frag = [](int q, int* t, int* i, int* lane) {
    *lane = (q >> 2) & 63;
    *i = (q & 3) + 4 * ((q >> 8) & 3);
    *t = q >> 10;
  };
  int t, i, lane;
  if (p < s64_w1()) {  // dW2[n][j]
    frag(p, &t, &i, &lane);
    return s.offs[T_W2] + (32 * (t >> 1) + crc(i) + 4 * (lane >> 5)) * GH + 32 * (t & 1) + (lane & 31);
  }
  if (p < s64_w3()) {  // dW1[n][j]
    frag(p - s64_w1(), &t, &i, &lane);
    const int j = 32 * (t & 1) + (lane & 31);
    return j < s.D ? s.offs[T_W1] + (32 * (t >> 1) + crc(i) + 4 * (lane >> 5)) * s.D + j : -1;
  }
  if (p < s64_b2()) {  // dW3[a][j]
    frag(p - s64_w3(), &t, &i, &lane);
    const int a_ = crc(i) + 4 * (lane >> 5);
    return a_ < head ? s.offs[T_W3] + a_ * GH + 32 * t + (lane & 31) : -1;
  }
  if (p < s64_b1()) return s.offs[T_B2] + (p - s64_b2());
  if (p < s64_b3()) return s.offs[T_B1] + (p - s64_b1());
  if (p < s64_ls()) {
    const int k = p - s64_b3();
    return k < head ? s.offs[T_B3] + k : -1;
  }
  if (p < s64_st()) {
    const int k = p - s64_ls();
    return (net == 0 && k < s.A) ? s.offs[0] + k : -1;
  }
  const int k = p - s64_st();  // loss sums behind the gradient vector (sums = grads + P)
  const bool mine = net == 0 ? (k == 0 || k == 2 || k == 3) : k == 1;
  return mine ? s.P + k : -1;
}
// One 256-position block of the reduction: block bx of gx along the slab, network `net`.  COH: the slabs were written and the results
// are read by OTHER workgroups of the same launch (k_epoch64): every access of handed-off bytes at agent scope.
// (nblocks_ / b_local_ / inv_bg_: the per-minibatch fields of `s`, passed beside it: see split64_tile)
template <bool COH, class TS>   // TS: Slab64ReduceArgs, or the same struct behind a kernel-argument reference (constant address space)
__device__ __forceinline__ void slab64_reduce_block(const TS& s, int bx, int net, int gx, int nblocks_, float b_local_, float inv_bg_, int& dst_o, float& acc_o) {
  const int p = bx * 256 + (int)threadIdx.x;
  if (p == 0 && net == 0) stc<COH>(s.sums + 4, b_local_);
  const int dst = p < s64_size() ? slab64_to_canonical(s, net, p) : -1;
  float acc = 0.f;
  if (dst >= 0) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;  // blocks of this network: net, net+2, ...
    const float* src = s.slabs + (size_t)net * s64_size() + p;
    const size_t stride = 2 * (size_t)s64_size();
    const int nslabs = (nblocks_ - net + 1) / 2;
    if (s.group <= 1) {
      const int n = nslabs;
      int w = 0;
      for (; w + 16 <= n; w += 16) {  // sixteen loads in flight; the adds keep the four-accumulator order below
        float x[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) x[u] = ldc<COH>(src + (size_t)(w + u) * stride);
#pragma unroll
        for (int u = 0; u < 16; u += 4) {
          a0 += x[u];
          a1 += x[u + 1];
          a2 += x[u + 2];
          a3 += x[u + 3];
        }
      }
      for (; w + 4 <= n; w += 4) {
        a0 += ldc<COH>(src + (size_t)w * stride);
        a1 += ldc<COH>(src + (size_t)(w + 1) * stride);
        a2 += ldc<COH>(src + (size_t)(w + 2) * stride);
        a3 += ldc<COH>(src + (size_t)(w + 3) * stride);
      }
      for (; w < n; ++w) a0 += ldc<COH>(src + (size_t)w * stride);
    } else {
      const int G = s.group, n = (nslabs + G - 1) / G;
      auto fold = [&](int gi) {  // tiles gi*G .. gi*G + G-1 in order (an idle wave of the block kernel added 0.f)
        const int t0 = gi * G;
        float x[4];                // G <= 4 (g_train_waves); all loads of a fold are independent
#pragma unroll
        for (int u = 0; u < 4; ++u) x[u] = (u < G && t0 + u < nslabs) ? ldc<COH>(src + (size_t)(t0 + u) * stride) : 0.f;
        return ((x[0] + x[1]) + x[2]) + x[3];
      };
      int w = 0;
      for (; w + 4 <= n; w += 4) {
        a0 += fold(w);
        a1 += fold(w + 1);
        a2 += fold(w + 2);
        a3 += fold(w + 3);
      }
      for (; w < n; ++w) a0 += fold(w);
    }
    acc = (a0 + a1) + (a2 + a3);
    if (dst < s.offs[1]) acc = __fadd_rn(acc, __fmul_rn(__fmul_rn(s.ent_coef, -b_local_), inv_bg_));   // (explicit: no fma contraction, the same bits from every kernel)
    // (inside a co-operative launch the gradient itself goes on in this thread's registers: only the loss sums behind it are read by
    //  another workgroup -- the statistics row)
    if (dst < s.P) s.grads[dst] = acc;
    else stc<COH>(s.grads + dst, acc);
  }
  dst_o = dst; acc_o = acc;   // (what this thread reduced: k_epoch64 goes on with it in registers)
  if (s.rec_sum != nullptr) {
    const size_t b = ((size_t)net * gx + bx) * kNormRec;
    block_norm_records<COH>(tensor_of_canonical(s.offs, s.P, dst), acc, s.rec_sum + b, s.rec_t + b);
  }
}
__global__ __launch_bounds__(256) void k_slab64_reduce(Slab64ReduceArgs s) {
  int dst; float acc;
  slab64_reduce_block<false>(s, blockIdx.x, blockIdx.y, gridDim.x, s.nblocks, s.b_local, s.inv_bg, dst, acc);
}

// The same reduction for MANY slabs per network (k_pair64_train: up to 256 per network).  82 workgroups cover the slab, so
// what matters is the number of BYTES each of them keeps in flight: sixteen groups of 64 threads split the slabs sixteen ways,
// every thread owns four consecutive positions and issues 16-byte loads (round 2: four groups of 256 threads with 4-byte
// loads, 2.2 TB/s = the latency bound of 64 outstanding 256-byte wave loads per wave; now 4x the bytes per load).  The
// sixteen partial sums of a position are added in a fixed order -- groups g and g + 8 first, then the eight pairs in
// group order -- then threads 0..255 finish as k_slab64_reduce does (entropy term, store, norm records: same 256 positions
// per block, same record table).  8 KB of LDS, not 16: beside two k_pair64_train workgroups of ANOTHER fleet segment
// (152 KB of a CU's 160) a 16 KB block does not fit, and the fleet's three update streams stopped overlapping (+5 %).
__global__ __launch_bounds__(1024) void k_slab64_reduce_wide(Slab64ReduceArgs s) {
  __shared__ float part[8][256];
  const int l = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int net = blockIdx.y;
  if (blockIdx.x == 0 && threadIdx.x == 0 && net == 0) s.sums[4] = s.b_local;
  {
    const int p4 = blockIdx.x * 256 + 4 * l;  // s64_size() is a multiple of 4: a quad is inside the slab or outside it
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
    if (p4 < s64_size()) {
      const float* src = s.slabs + (size_t)net * s64_size() + p4;
      const size_t stride = 2 * (size_t)s64_size();
      const int nslabs = (s.nblocks - net + 1) / 2;
      const int per = (nslabs + 15) / 16;
      int w = g * per;
      const int n = min(nslabs, w + per);
      for (; w + 8 <= n; w += 8) {
        f32x4 x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = *reinterpret_cast<const f32x4*>(src + (size_t)(w + u) * stride);
#pragma unroll
        for (int u = 0; u < 8; u += 2) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { a0[e] += x[u][e]; a1[e] += x[u + 1][e]; }
        }
      }
      for (; w < n; ++w) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(src + (size_t)w * stride);
#pragma unroll
        for (int e = 0; e < 4; ++e) a0[e] += x[e];
      }
    }
    if (g >= 8) {
#pragma unroll
      for (int e = 0; e < 4; ++e) part[g - 8][4 * l + e] = a0[e] + a1[e];
    }
    __syncthreads();
    if (g < 8) {
#pragma unroll
      for (int e = 0; e < 4; ++e) part[g][4 * l + e] = (a0[e] + a1[e]) + part[g][4 * l + e];
    }
  }
  __syncthreads();
  if (threadIdx.x >= 256) return;  // waves 4..15 are done; the barrier inside block_norm_records counts the four that remain
  const int tp = threadIdx.x, p = blockIdx.x * 256 + tp;
  const int dst = p < s64_size() ? slab64_to_canonical(s, net, p) : -1;
  float acc = 0.f;
  if (dst >= 0) {
#pragma unroll
    for (int q = 0; q < 8; ++q) acc += part[q][tp];
    if (dst < s.offs[1]) acc = __fadd_rn(acc, __fmul_rn(__fmul_rn(s.ent_coef, -s.b_local), s.inv_bg));
    s.grads[dst] = acc;
  }
  if (s.rec_sum != nullptr) {
    const size_t b = ((size_t)net * gridDim.x + blockIdx.x) * kNormRec;
    block_norm_records(tensor_of_canonical(s.offs, s.P, dst), acc, s.rec_sum + b, s.rec_t + b);
  }
}

// ---- rollout-time forward + sampling for 64-wide nets: one wave per 32-row tile and network ----
template <int DP>
__global__ __launch_bounds__(g_waves(DP) * 64, 2) void k_fused64_act(FusedActArgs a) {
  using L = Lay64<DP>;
  constexpr int ldx = L::LDX, per = DP / 4;
  const int tid0 = threadIdx.x, lane = tid0 & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int gw = blockIdx.x * L::NWV + wave;  // global wave: network = gw & 1, tile = gw >> 1
  const int net = gw & 1, tile = gw >> 1;
  const int wb = wave * L::WAVE;
  const int row0 = tile * GR;
  if (row0 >= a.rows) return;
  if ((net == 0 && !a.want_pi) || (net == 1 && !a.want_v)) return;
  const FusedNet W = a.net[net];
#pragma unroll
  for (int i = lane; i < GR * per; i += 64) {
    const int rr = i / per, c = i - rr * per;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row0 + rr < a.rows) v = ldg16(a.X, (unsigned)(row0 + rr) * (unsigned)(DP * 4) + (unsigned)(c * 16));
    *reinterpret_cast<f32x4*>(&lds[wb + L::X + rr * ldx + 4 * c]) = v;
  }
  tile64_forward<DP>(W, wb, lane);
  if (lane >= GR) return;
  const int row = row0 + lane;
  if (row >= a.rows) return;
  const int db = opaque(wb + L::DO + lane * FLDO);
  if (net == 1) {
    a.v[row] = lds[db] + W.b3[0];
    return;
  }
  float lp = 0.f;
  float z0 = 0.f, z1 = 0.f, z2 = 0.f, z3 = 0.f;
  for (int k = 0; k < a.A; ++k) {
    const float m = lds[db + k] + W.b3[k];
    if (a.mu) a.mu[(size_t)row * a.ldmu + k] = m;
    if (a.sample) {
      float e;
      if (a.eps != nullptr) {
        e = a.eps[(size_t)row * a.A + k];
      } else {
        if ((k & 3) == 0) {
          float z[4];
          box_muller4(philox4x32_10((uint32_t)(row + a.row0), (uint32_t)(k >> 2), a.draw + (a.draw_base ? *a.draw_base : 0u),
                                    0x45505331u, (uint32_t)a.seed, (uint32_t)(a.seed >> 32)), z);
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

}  // namespace mobrob
