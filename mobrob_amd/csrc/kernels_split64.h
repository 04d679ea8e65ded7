// Minibatch gradient of 64-wide networks for SMALL minibatches: one workgroup (four waves) per 32-row tile.
//
// k_fused64_train gives every 32-row tile to ONE wave: 400 dependent 32x32x2 MFMAs = 10.7 us at the MFMA rate of one
// SIMD, 18 us measured, plus 12 us of fixed cost (LDS weight mirror, wave fold).  That is the right shape when a
// minibatch has thousands of tiles; the reference's own configurations (data/configs/*.yaml: batch_size 100 -> four
// tiles) run 800-1600 dependent optimizer steps of that size per train(), and the 31 us launch was two thirds of a
// step.  Here the four waves of a workgroup share ONE tile of one network:
//
//   phase      waves 0 / 1                                   waves 2 / 3
//   L1, L2     one 32-column block of the layer each           idle
//   head       wave 0 / 1: one of the head GEMM's two accumulation chains each (even / odd k-groups)   idle
//   loss       wave 0: adds the two partial tiles (the block kernel's `acc + acc2`) and runs the loss stage       idle
//   dh2 | dW3  dz2 block 0 / 1 (own buffer, h2 stays intact)   dW3 tile 0 / 1
//   dh1 | dW2  dz1 block 0 / 1                                 dW2 tiles (a, b) / (c, d)
//   dW1        tile a / b                                      tile c / d (observations wider than 32), bias sums
//
// Every accumulator sees exactly the MFMA sequence (operands, k order, start value) it sees in tile64_train, the
// loss stage is the same code, and the slab written per TILE is folded by k_slab64_reduce in the grouping the
// wave fold of k_fused64_train used -> gradients, norms, parameters are BIT-IDENTICAL to the one-wave-per-tile
// kernel (tests/test_engine_gpu.py::test_split_tile_kernel_is_bit_identical...).  Weights come from L2 through the
// fragment packs (they change every optimizer step, so an LDS mirror would be reloaded per launch anyway); each wave
// fetches ALL fragments of its next phases before the barrier in front of them.
#pragma once
#include "kernels_fused64.h"
#include "kernels_rollout.h"  // Frags<NKG>

namespace mobrob {

template <int DP>
struct LayS64 {
  static constexpr int LDX = DP + 4;
  static constexpr int X = 0;
  static constexpr int H1 = X + GR * LDX;
  static constexpr int H2 = H1 + GR * GLDH;
  static constexpr int Z1 = H2 + GR * GLDH;   // dz1 / dz2 in buffers of their own: dW2 / dW3 still read h1 / h2
  static constexpr int Z2 = Z1 + GR * GLDH;
  static constexpr int DO = Z2 + GR * GLDH;    // head partial of wave 0 (even k-groups); after the loss stage: dL/d(head)
  static constexpr int DO2 = DO + GR * FLDO;   // head partial of wave 1 (odd k-groups)
  static constexpr int GACC = DO2 + GR * FLDO; // [2][32] head-bias / log_std gradient sums
  static constexpr int CST = GACC + 64;        // [3][32] per-action constants
  static constexpr int END = CST + 96;
};
inline size_t split64_lds_bytes(int Dp) {
  return (size_t)(GR * (Dp + 4) + 4 * GR * GLDH + 2 * GR * FLDO + 64 + 96) * sizeof(float);
}

template <int NKG, bool COH = false>
__device__ __forceinline__ Frags<NKG> load_frags(const f32x4* __restrict__ Bp, int lane, int nkg = NKG) {
  Frags<NKG> w;
  const unsigned bo = opaque_u((unsigned)lane * 16u);
#pragma unroll
  for (int kg = 0; kg < NKG; ++kg) w.f[kg] = kg < nkg ? ldg16c<COH>(Bp, bo + (unsigned)kg * 1024u) : f32x4{0.f, 0.f, 0.f, 0.f};
  return w;
}
// c += A[32 x 8*nkg] (LDS, row stride LDA) . B (one 32-column block, fragments in registers): ONE accumulation chain,
// k order of gemm_lds_packed_r32
template <int LDA, int NKG>
__device__ __forceinline__ void gemm_one(int a_off, const Frags<NKG>& w, f32x16& c, int lane, int nkg = NKG) {
  const int r = lane & 31, h = lane >> 5;
  const int ab = 4 * opaque((a_off + r * LDA + 4 * h) >> 2);
#pragma unroll
  for (int kg = 0; kg < NKG; ++kg) {
    if (kg < nkg) {
      const f32x4 u = *reinterpret_cast<const f32x4*>(&lds[ab + 8 * kg]);
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) c = MFMA32(u[s_], w.f[kg][s_], c);
    }
  }
}

// slab of tile (blockIdx.x >> 1) of network (blockIdx.x & 1): layout of k_fused64_train's block slab
// NJ: action pairs the loss stage is unrolled over (2 NJ >= A).  The head tile's columns beyond A are exact +0 from the
// zero-padded head pack -- the very values tile64_train writes there -- so the shorter loops change no bit.
// The body as a device function of the block index `vb` (= blockIdx.x of k_split64_train; k_epoch64 runs it per minibatch inside one
// launch).  COH: the weight packs / parameters were written and the slab is read by OTHER workgroups of the same launch: those accesses
// at agent scope (kernels_fused.h ldc / stc / ldg16c / stg16c); the rollout arrays are constant during an update: plain loads.
// (rows_ / count_ / advstat_ / inv_bg_: the per-minibatch fields of `a`, passed beside it -- k_epoch64 changes them per step, and a
//  modified local COPY of the argument struct, with its runtime-indexed net[2], would live in scratch memory)
template <int DP, int NJ, bool COH, class TA>   // TA: Fused64TrainArgs, or the same struct behind a kernel-argument reference (constant address space)
__device__ __forceinline__ void split64_tile(const TA& a, int vb, const int* __restrict__ rows_, int count_, const double* advstat_, float inv_bg_) {
  using L = LayS64<DP>;
  constexpr int ldx = L::LDX, per = DP / 4, NKG1 = DP / 8;
  constexpr bool two = DP > 32;
  const int tid0 = threadIdx.x;
  const int lane = tid0 & 63, r = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int net = vb & 1, tile = vb >> 1;
  const int row0 = tile * GR, cnt = count_;
  // (the fields of a.net[net] this kernel uses, copied one by one: `a` may sit in the constant address space)
  const struct { const f32x4 *W1f, *W2f, *W3f, *W2b, *W3b; const float *b1s, *b2s, *b3; int head; } W = {
      a.net[net].W1f, a.net[net].W2f, a.net[net].W3f, a.net[net].W2b, a.net[net].W3b, a.net[net].b1s, a.net[net].b2s, a.net[net].b3, a.net[net].head};
  float* slab = a.slabs + (size_t)vb * s64_size();

  // ---- weight fragments of the forward phases (waves 0 / 1: column block `wave`) ----
  Frags<NKG1> f1;
  Frags<8> f2;
  Frags<4> fh;  // head k-groups wave, wave + 2, wave + 4, wave + 6: the chain `acc` (wave 0) / `acc2` (wave 1) of tile64_train's head GEMM
  if (wave < 2) {
    f1 = load_frags<NKG1, COH>(W.W1f + (size_t)wave * NKG1 * 64, lane);
    f2 = load_frags<8, COH>(W.W2f + (size_t)wave * 8 * 64, lane);
#pragma unroll
    for (int j = 0; j < 4; ++j) fh.f[j] = ldg16c<COH>(W.W3f, (unsigned)lane * 16u + (unsigned)(2 * j + wave) * 1024u);
  }
  // ---- observation rows of the tile -> LDS ----
#pragma unroll
  for (int i0 = 0; i0 < GR * per; i0 += 256) {
    const int i = i0 + tid0;
    if (i < GR * per) {
      const int rr = i / per, c = i - rr * per;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (row0 + rr < cnt) v = ldg16(a.obs, (unsigned)rows_[row0 + rr] * (unsigned)(DP * 4) + (unsigned)(c * 16));
      *reinterpret_cast<f32x4*>(&lds[L::X + rr * ldx + 4 * c]) = v;
    }
  }
  // ---- operands of the loss stage (wave 0, two lanes per row), in flight during the forward pass ----
  const bool llive = row0 + r < cnt;
  float l_adv = 0.f, l_old = 0.f, l_act[NJ];
  float adv_mean = 0.f, adv_sd = 1.f;
  bool adv_on = false;
  if (wave == 0) {
    const unsigned src = llive ? (unsigned)rows_[row0 + r] : 0u;
    if (net == 0) {
      const float* arow = a.actions + (size_t)src * a.A + h;
#pragma unroll
      for (int j = 0; j < NJ; ++j) l_act[j] = (2 * j + h < a.A && llive) ? arow[2 * j] : 0.f;
      if (llive) { l_adv = a.adv[src]; l_old = a.old_logp[src]; }
    } else {
#pragma unroll
      for (int j = 0; j < NJ; ++j) l_act[j] = 0.f;
      if (llive) {
        l_old = a.ret[src];
        if (a.clip_vf >= 0.f) l_adv = a.old_values[src];
      }
    }
    const double n = advstat_[2];
    adv_on = n > 1.0;
    const double m = advstat_[0] / (n > 0 ? n : 1.0);
    double var = adv_on ? (advstat_[1] - n * m * m) / (n - 1.0) : 0.0;
    if (var < 0.0) var = 0.0;
    adv_mean = (float)m;
    adv_sd = (float)sqrt(var);
  } else {
#pragma unroll
    for (int j = 0; j < NJ; ++j) l_act[j] = 0.f;
  }
  if (tid0 >= 64 && tid0 < 96) {  // per-action constants
    const int k = tid0 - 64;
    float iv = 0.f, lc = 0.f, bb = 0.f;
    if (net == 0 && k < a.A) {
      const float sd = expf(ldc<COH>(a.log_std + k));
      iv = 1.0f / (sd * sd);
      lc = logf(sd) + 0.91893853320467274178f;
    }
    if (k < W.head) bb = ldc<COH>(W.b3 + k);
    lds[L::CST + k] = iv;
    lds[L::CST + 32 + k] = lc;
    lds[L::CST + 64 + k] = bb;
  }
  if (wave == 2) lds[L::GACC + lane] = 0.f;
  __syncthreads();

  // ---- layer 1, layer 2: waves 0 / 1 take one column block each ----
  if (wave < 2) {
    f32x16 c = splat16(ldc<COH>(W.b1s + 32 * wave + r));
    gemm_one<ldx, NKG1>(L::X, f1, c, lane);
    const int o = opaque(L::H1 + 4 * h * GLDH + 32 * wave + r);
#pragma unroll
    for (int i = 0; i < 16; ++i) lds[o + crc(i) * GLDH] = fast_tanh_scaled(c[i]);
  }
  __syncthreads();
  // backward weight fragments (dh2: block `wave` of W3b, dh1: block `wave` of W2b), in flight from here on
  Frags<4> b3;
  Frags<8> b2;
  const int nkh = W.head <= 16 ? 2 : 4;  // k-groups of 8 head columns; those beyond `head` are zero
  if (wave < 2) {
    b3 = load_frags<4, COH>(W.W3b + (size_t)wave * 4 * 64, lane, nkh);
    b2 = load_frags<8, COH>(W.W2b + (size_t)wave * 8 * 64, lane);
    f32x16 c = splat16(ldc<COH>(W.b2s + 32 * wave + r));
    gemm_one<GLDH, 8>(L::H1, f2, c, lane);
    const int o = opaque(L::H2 + 4 * h * GLDH + 32 * wave + r);
#pragma unroll
    for (int i = 0; i < 16; ++i) lds[o + crc(i) * GLDH] = fast_tanh_scaled(c[i]);
  }
  __syncthreads();

  // ---- head: the two accumulation chains of tile64_train's head GEMM on waves 0 / 1 (round 6; one wave ran both: 64 dependent-issue
  //      MFMAs = 1.7 us of a 12 us launch).  Each chain sees its own MFMA sequence; the loss stage adds the partial tiles as `acc + acc2`
  //      did: the same bits. ----
  if (wave < 2) {
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
  __syncthreads();
  // ---- loss: wave 0 ----
  float s_pl = 0.f, s_vl = 0.f, s_kl = 0.f, s_cf = 0.f;
  if (wave == 0) {
    // loss: two lanes per row (q = action parity); dL/d(head) -> head tile, zero padded   (tile64_train's stage)
    const int rr = r, q = h;
    const bool live = llive;
    const int db = opaque(L::DO + rr * FLDO + q);
    const int d2 = opaque(L::DO2 + rr * FLDO + q);
    const int cb = opaque(L::CST + q);
    const int gb = opaque(L::GACC + q);
    const int A = a.A;
    if (net == 0) {
      // per-action constants of this lane's actions k = 2j + q (registers), masks by multiplication: the same values
      // as the predicated form (x * 1 and s + 0 are exact), without per-lane branches
      float c_iv[NJ], c_lc[NJ], c_bb[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        c_iv[j] = lds[cb + 2 * j];
        c_lc[j] = lds[cb + 32 + 2 * j];
        c_bb[j] = lds[cb + 64 + 2 * j];
      }
      float lp = 0.f;
      float dk[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const float on = (2 * j + q < A && live) ? 1.f : 0.f;
        const float d = on * (l_act[j] - ((lds[db + 2 * j] + lds[d2 + 2 * j]) + c_bb[j]));
        lp += on * (-(d * d) * (0.5f * c_iv[j]) - c_lc[j]);
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
          s_pl += fminf(s1, s2);
          s_cf += (fabsf(ratio - 1.0f) > a.clip) ? 1.f : 0.f;
          s_kl += (ratio - 1.0f) - log_ratio;
        }
        const float in_range = (ratio >= lo && ratio <= hi) ? 1.f : 0.f;
        const float w1 = (s1 < s2) ? 1.f : ((s1 > s2) ? 0.f : 0.5f);
        g_logp = -(w1 * adv + (1.0f - w1) * adv * in_range) * inv_bg_ * ratio;
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) {  // k = 2j + q
        const float on = (2 * j + q < A && live) ? 1.f : 0.f;
        const float d = dk[j];
        float gm = on * (g_logp * d * c_iv[j]);
        float gl = on * (g_logp * (d * d * c_iv[j] - 1.0f));
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
        value_loss_terms((lds[db] + lds[d2]) + lds[cb + 64], l_old, l_adv, a.clip_vf, sq, gv_);
        s_vl += sq;
        dv = a.vf_coef * gv_ * inv_bg_;
      }
      lds[db] = dv;  // column q of the row: dv (q = 0) or 0; the columns beyond are +0 already
      const float t = wave_sum(dv);
      if (lane == 0) lds[gb] += t;
    }
  }
  __syncthreads();

  // ---- dh2 -> dz2 (waves 0 / 1)  |  dW3 tile 0 / 1 (waves 2 / 3) ----
  f32x16 gA = zero16(), gB = zero16();  // this wave's weight-gradient tiles of the phase at hand
  if (wave < 2) {
    f32x16 c = zero16();
    gemm_one<FLDO, 4>(L::DO, b3, c, lane, nkh);
    const int oh = opaque(L::H2 + 4 * h * GLDH + 32 * wave + r), oz = opaque(L::Z2 + 4 * h * GLDH + 32 * wave + r);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float hv = lds[oh + crc(i) * GLDH];
      lds[oz + crc(i) * GLDH] = c[i] * (1.0f - hv * hv);
    }
  } else {
    const int ao = opaque(L::DO + h * FLDO + r);
    const int bo = opaque(L::H2 + h * GLDH + 32 * (wave - 2) + r);
#pragma unroll 4
    for (int k = 0; k < GR; k += 2) gA = MFMA32(lds[ao + k * FLDO], lds[bo + k * GLDH], gA);
    const unsigned lb = (unsigned)lane * 16u;  // put(s64_w3(), wave - 2, gA)
#pragma unroll
    for (int qd = 0; qd < 4; ++qd)
      stg16c<COH>(slab + s64_w3(), (unsigned)(((wave - 2) * 4 + qd) * 64) * 16u + lb, f32x4{gA[4 * qd], gA[4 * qd + 1], gA[4 * qd + 2], gA[4 * qd + 3]});
  }
  __syncthreads();

  // ---- dh1 -> dz1 (waves 0 / 1)  |  dW2 tiles (a, b) / (c, d) (waves 2 / 3) ----
  if (wave < 2) {
    f32x16 c = zero16();
    gemm_one<GLDH, 8>(L::Z2, b2, c, lane);
    const int oh = opaque(L::H1 + 4 * h * GLDH + 32 * wave + r), oz = opaque(L::Z1 + 4 * h * GLDH + 32 * wave + r);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float hv = lds[oh + crc(i) * GLDH];
      lds[oz + crc(i) * GLDH] = c[i] * (1.0f - hv * hv);
    }
  } else {
    gA = zero16();
    const int ao = opaque(L::Z2 + h * GLDH + r);
    const int bo = opaque(L::H1 + h * GLDH + 32 * (wave - 2) + r);
#pragma unroll 4
    for (int k = 0; k < GR; k += 2) {
      const float y = lds[bo + k * GLDH];
      gA = MFMA32(lds[ao + k * GLDH], y, gA);
      gB = MFMA32(lds[ao + k * GLDH + 32], y, gB);
    }
    // slab tiles t = ib*2 + jb: (a, b) = (0, 2) of column block 0, (c, d) = (1, 3) of column block 1
    const unsigned lb = (unsigned)lane * 16u;
    const int ta = wave - 2, tb = wave;
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      stg16c<COH>(slab + s64_w2(), (unsigned)((ta * 4 + qd) * 64) * 16u + lb, f32x4{gA[4 * qd], gA[4 * qd + 1], gA[4 * qd + 2], gA[4 * qd + 3]});
      stg16c<COH>(slab + s64_w2(), (unsigned)((tb * 4 + qd) * 64) * 16u + lb, f32x4{gB[4 * qd], gB[4 * qd + 1], gB[4 * qd + 2], gB[4 * qd + 3]});
    }
  }
  __syncthreads();

  // ---- dW1: tile a / b on waves 0 / 1, c / d on waves 2 / 3 (observations wider than 32); bias sums ----
  {
    const int jb = wave >> 1, ib = wave & 1;  // tile (neuron block ib, input block jb): a=00 b=10 c=01 d=11
    if (two || jb == 0) {
      f32x16 gw = zero16();
      const int ao = opaque(L::Z1 + h * GLDH + 32 * ib + r);
      const int c0 = (r < DP) ? r : 0;
      const int c1 = (32 + r < DP) ? 32 + r : c0;
      const int bo = opaque(L::X + h * ldx + (jb == 0 ? c0 : c1));
#pragma unroll 4
      for (int k = 0; k < GR; k += 2) gw = MFMA32(lds[ao + k * GLDH], lds[bo + k * ldx], gw);
      const int t = ib * 2 + jb;
      const unsigned lb = (unsigned)lane * 16u;
#pragma unroll
      for (int qd = 0; qd < 4; ++qd)
        stg16c<COH>(slab + s64_w1(), (unsigned)((t * 4 + qd) * 64) * 16u + lb, f32x4{gw[4 * qd], gw[4 * qd + 1], gw[4 * qd + 2], gw[4 * qd + 3]});
    }
    if (wave >= 2) {  // column sums of dz2 (wave 2) / dz1 (wave 3): lane c <-> column c
      const int o = opaque((wave == 2 ? L::Z2 : L::Z1) + lane);
      float s0 = 0.f, s1 = 0.f;
#pragma unroll 4
      for (int rr = 0; rr < GR; rr += 2) {
        s0 += lds[o + rr * GLDH];
        s1 += lds[o + (rr + 1) * GLDH];
      }
      stc<COH>(slab + (wave == 2 ? s64_b2() : s64_b1()) + lane, s0 + s1);
    }
  }
  if (wave == 0) {
    const float t0 = wave_sum(s_pl), t1 = wave_sum(s_vl), t2 = wave_sum(s_kl), t3 = wave_sum(s_cf);
    if (lane < 4) stc<COH>(slab + s64_st() + lane, lane == 0 ? t0 : (lane == 1 ? t1 : (lane == 2 ? t2 : t3)));
    if (lane < 32) {
      stc<COH>(slab + s64_b3() + lane, lds[L::GACC + lane]);
      stc<COH>(slab + s64_ls() + lane, lds[L::GACC + 32 + lane]);
    }
  }
}
template <int DP, int NJ>
__global__ __launch_bounds__(256, 1) void k_split64_train(Fused64TrainArgs a) { split64_tile<DP, NJ, false>(a, (int)blockIdx.x, a.rows, a.count, a.advstat, a.inv_bg); }

}  // namespace mobrob
