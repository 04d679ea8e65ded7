// Minibatch gradient of 64-wide networks for LARGE minibatches: a 32-row tile belongs to a PAIR of waves; two pairs
// make a workgroup (one wave per SIMD), two workgroups share a CU (two waves per SIMD).
//
// k_fused64_train (one wave per tile, four waves per CU, weights mirrored in LDS) keeps the matrix pipe busy about half
// of the time: a lone wave per SIMD runs gather -> 8 GEMM phases -> epilogues -> loss as one dependent chain, and 160 KB
// of LDS (54 KB weight mirror + 25 KB per tile) leave no room for a second wave per SIMD.  Here each wave of a pair owns
// one 32-column block of L1, L2, dh2, dh1, one of the two accumulation chains of the head and half of the
// weight-gradient tiles (dW2 a,b | c,d; dW3 a | b; dW1 a(,c) | b(,d)).  Column blocks are private to their wave in every
// in-place epilogue (dz2 over h2, dz1 over h1) and in the weight-gradient phases that read them, so a tile needs seven
// workgroup barriers and no extra buffers; a pair needs 30-36 KB of LDS.
//
// Weights: up to 32 observation columns everything is resident for the whole launch -- layer-1, layer-2, head and dh2
// fragments in registers (256 per wave is the budget of two waves per SIMD), the dh1 operand (backward pack of W2,
// 16 KB) in LDS, shared by the workgroup's two pairs; wider observations fetch layer-1 / layer-2 / backward fragments
// from L2 in every tile just ahead of their phase.
//
// Loss stage (role 0 only; the partner waits): measured with cycle stamps (-DMOBROB_PAIR_STAMPS, scratch/pair_stamps.py)
// it was a third of a tile.  It is unrolled over NJ = ceil(A / 2) action pairs instead of 16 (template parameter; the
// head tile's columns beyond A are exact zeros from the zero-padded head pack, nothing rewrites them), masks by
// multiplication instead of per-lane branches, and for narrow heads keeps the per-action constants in registers and sums
// dL/d(mean), dL/d(log_std) per lane over all tiles, reducing over the 32 rows once at the end.
//
// Per tile and accumulator the MFMA sequence is the one of tile64_train; the accumulators run over the tiles of the pair
// and are written once, the two pairs of a workgroup added through LDS into ONE slab (k_slab64_reduce_wide sums up to 256
// slabs per network in fixed order).  The
// summation order over TILES therefore differs from k_fused64_train's, so the two kernels agree to rounding, not bit for
// bit (tests: 1e-5 of each tensor's scale, both against the oracle); run to run the kernel is deterministic.
// Measured (MI355X, 65 536-row minibatch, us per launch, block kernel -> this one): 14/2 76 -> 65, 26/2 106 -> 68,
// 43/2 120 -> 77, 58/12 131 -> 82; BASELINE config 2 53.4 -> 57.5 M env-steps/s, mixed fleet 64.2 -> 80.5 M.
#pragma once
#include "kernels_split64.h"

namespace mobrob {

// timing-only ablation (never in the product build): -DPAIR_SKIP=<mask>
//   1 loss operands are constants (no actions / advantage / old log-prob gathers)   2 no exp / division in the loss stage
//   4 no loss stage at all (dL/d(head) = head)   8 no exp / rcp in the tanh epilogues   16 no observation gathers
//   32 no weight-gradient MFMAs (dW3, dW2, dW1)   64 no bias-gradient column sums   128 no dtanh read-modify-write
//   256 no tile loop (fixed cost of a launch)   512 no slab staging / stores at the end
#ifndef PAIR_SKIP
#define PAIR_SKIP 0
#endif
#define PAIR_ON(bit) (!((PAIR_SKIP) & (bit)))

// Head fragments resident in registers for the launch?  256 registers per wave is the budget of two waves per SIMD; the
// exceptions are the (observation width, action pairs) combinations the compiler spilled on with them resident
// (__graft_entry__.build() fails on a spill, so this table is checked by every build).
constexpr bool pair64_fh_resident(int DP, int NJ) {
  if (DP >= 64) return false;
  if (DP == 48) return NJ == 1 || NJ == 4 || NJ >= 10;
  if (DP == 32) return NJ <= 4;
  return NJ <= 10;
}

template <int DP>
struct LayP64 {
  static constexpr int LDX = DP + 4;
  static constexpr int X = 0;
  static constexpr int H1 = X + GR * LDX;
  static constexpr int H2 = H1 + GR * GLDH;
  static constexpr int DO = H2 + GR * GLDH;    // head partial of wave 0; after the loss stage: dL/d(head)
  static constexpr int DO2 = DO + GR * FLDO;   // head partial of wave 1
  static constexpr int GACC = DO2 + GR * FLDO; // [2][32] head-bias / log_std gradient sums (wide heads only: see kAccRegs)
  static constexpr int CST = GACC + 64;        // [3][32] per-action constants (wide heads only: see kCstRegs)
  static constexpr int END = CST + 96;
};
inline size_t pair64_lds_bytes(int Dp) {  // per WORKGROUP: two pairs (+ the backward pack of W2 where two workgroups still fit a CU)
  return (2 * (size_t)(GR * (Dp + 4) + 2 * GR * GLDH + 2 * GR * FLDO + 64 + 96) + (Dp <= 32 ? 4096 : 0)) * sizeof(float);
}
constexpr int kPairsPerCu = 4;  // two workgroups of two pairs: two waves per SIMD

// A workgroup is TWO pairs (waves 0,1 and 2,3 -> one wave per SIMD; 128-thread workgroups of one pair were placed on
// SIMD 0/1 only and left half of every CU idle).  The pairs of a workgroup share its barriers and nothing else.
// grid: groups of 16 blocks -- 8 policy workgroups for block sequences 8g..8g+7, then 8 value workgroups for the SAME
// sequences (blockIdx mod 8 = XCD: the workgroups that gather the same observation rows share an L2).  Pair p of block
// sequence s owns tile sequence 2s + p: tiles 2s+p, 2s+p+nseq, ...  One slab per workgroup: index 2 * block sequence + network.
// NJ: action pairs the loss stage loops over (2 NJ >= A; the head tile's columns beyond A are exact zeros from the zero-padded
// head pack, so nothing has to rewrite them): the loops are unrolled NJ times instead of 16 with 16 - NJ dead predicates.
template <int DP, int NJ>
__global__ __launch_bounds__(256, 2) void k_pair64_train(Fused64TrainArgs a, int nseq) {
  using L = LayP64<DP>;
  constexpr int ldx = L::LDX, per = DP / 4, NKG1 = DP / 8;
  constexpr bool two = DP > 32;
  constexpr int NG = (GR * per + 127) / 128;
  const int tid0 = threadIdx.x & 127;  // thread of the pair
  const int lane0 = tid0 & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6) & 1);   // role in the pair
  const int pr = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 7));          // pair of the workgroup
  const int lb = pr * L::END;                                                       // the pair's LDS region
  const int grp = blockIdx.x >> 4, j16 = blockIdx.x & 15;
  const int net = j16 >> 3, bseq = 8 * grp + (j16 & 7);
  if (2 * bseq >= nseq) return;  // whole workgroup
  const int seq = 2 * bseq + pr;
  const bool pair_on = seq < nseq;  // an odd nseq leaves the last workgroup's second pair without tiles: it only keeps the barriers
  const FusedNet W = a.net[net];
  const int cnt = pair_on ? a.count : 0;  // a pair without tiles sees an empty minibatch
  const int ntiles = (a.count + GR - 1) / GR;
  const int niter = (ntiles - 2 * bseq + nseq - 1) / nseq;  // tiles of the workgroup's first pair: both pairs make this many rounds
  // ---- forward weight fragments of this wave's column block: registers for the whole launch ----
  // What a wave fetches from L2 in EVERY tile is what paces this kernel: a CU sustains ~14 GB/s of L1 misses (its
  // outstanding-miss window over the L2 latency), and 8 waves x 13 KB of fragments per tile round were 7 us of a 24 us
  // round.  Up to 32 observation columns everything is resident: layer-1, layer-2, head and dh2 fragments in registers,
  // the dh1 operand (backward pack of W2, 16 KB) in LDS, shared by the workgroup's two pairs.
  constexpr bool kSmall = DP <= 32;
  constexpr bool kF2Resident = DP <= 32;  // wider observations need the registers (next tile's rows, a fifth accumulator tile)
  Frags<8> f2;
  if (kF2Resident) f2 = load_frags<8>(W.W2f + (size_t)wave * 8 * 64, lane0);
  constexpr bool kFhResident = pair64_fh_resident(DP, NJ);
  // Heads of <= 12 outputs (NJ <= 6; the value head has one): head GEMM and dW3 on v_mfma_f32_16x16x4_f32 -- wave w
  // computes head rows 16w..16w+15 over the full K = 64 (no partial tile of the partner to add), dW3 is two 16x16 tiles
  // per wave: half the matrix cycles of the zero-padded 32-wide tiles (16 of a wave's 160 MFMA32-equivalents per tile).
  constexpr bool kH16 = NJ <= 6;
  Frags<4> fh;  // kH16: the four 16-wide k-groups of the 16x16x4 head pack; else head k-groups wave, wave + 2, wave + 4, wave + 6
  auto load_fh = [&](int ln) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      fh.f[j] = kH16 ? ldg16(W.W3h, (unsigned)ln * 16u + (unsigned)j * 1024u)
                     : ldg16(W.W3f, (unsigned)ln * 16u + (unsigned)(2 * j + wave) * 1024u);
  };
  if (kFhResident) load_fh(lane0);
  const float bias1 = W.b1s[32 * wave + (lane0 & 31)], bias2 = W.b2s[32 * wave + (lane0 & 31)];
  const int nkh = W.head <= 16 ? 2 : 4;
  Frags<NKG1> f1;
  Frags<4> b3;
  constexpr int W2B_LDS = 2 * L::END;  // [2 blocks][8 k-groups][64 lanes][4]
  constexpr bool kB3Resident = kSmall && !(DP == 32 && NJ > 10);
  if (kSmall) {
    f1 = load_frags<NKG1>(W.W1f + (size_t)wave * NKG1 * 64, lane0);
    if (kB3Resident) b3 = load_frags<4>(W.W3b + (size_t)wave * 4 * 64, lane0, nkh);
#pragma unroll
    for (int u = 0; u < 4; ++u)
      reinterpret_cast<f32x4*>(&lds[W2B_LDS])[threadIdx.x + u * 256] = W.W2b[threadIdx.x + u * 256];
  }

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

  // weight-gradient accumulators of this wave: dW2 (x0, x1) x y_wave | dW3 x y_wave | dW1 x_wave x (y0, y1)
  f32x16 gW2a = zero16(), gW2b = zero16(), gW3 = zero16(), gW1a = zero16(), gW1c = zero16();
  f32x4 gW3h0 = {0.f, 0.f, 0.f, 0.f}, gW3h1 = gW3h0;  // kH16: dW3 [16 head rows][own 32 columns] as two 16x16 tiles
  float gb2 = 0.f, gb1 = 0.f, s_pl = 0.f, s_vl = 0.f, s_kl = 0.f, s_cf = 0.f;
  // loss stage (role 0, lane = (row r, action parity q)): per-action constants of the lane's actions k = 2j + q, and its
  // running sums of dL/d(mean) / dL/d(log_std) over the rows it has seen (reduced over the 32 rows at the end)
  // Narrow heads keep both in registers; wide ones (the register file is full at two waves per SIMD) keep the constants
  // in LDS and sum dL/d(mean), dL/d(log_std) over the rows of every tile with a butterfly into LDS, as tile64_train does.
  constexpr bool kCstRegs = NJ <= 2, kAccRegs = NJ <= 6;
  constexpr int NC = kCstRegs ? NJ : 1, NA = kAccRegs ? NJ : 1;
  float c_iv[NC], c_lc[NC], c_bb[NC], acc_gm[NA], acc_gl[NA];
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    const int k = 2 * j + (lane0 >> 5);
    float iv = 0.f, lc = 0.f, bb = 0.f;
    if (net == 0 && k < a.A) {
      const float sd = expf(a.log_std[k]);
      iv = 1.0f / (sd * sd);
      lc = logf(sd) + 0.91893853320467274178f;
    }
    if (k < W.head) bb = W.b3[k];
    c_iv[j] = iv; c_lc[j] = lc; c_bb[j] = bb;
  }
#pragma unroll
  for (int j = 0; j < NA; ++j) acc_gm[j] = acc_gl[j] = 0.f;
  if (tid0 < 32) {  // LDS copies (read by the wide-head variants only)
    const int k = tid0;
    float iv = 0.f, lc = 0.f, bb = 0.f;
    if (net == 0 && k < a.A) {
      const float sd = expf(a.log_std[k]);
      iv = 1.0f / (sd * sd);
      lc = logf(sd) + 0.91893853320467274178f;
    }
    if (k < W.head) bb = W.b3[k];
    lds[lb + L::CST + k] = iv;
    lds[lb + L::CST + 32 + k] = lc;
    lds[lb + L::CST + 64 + k] = bb;
  }
  if (tid0 >= 64) lds[lb + L::GACC + (tid0 - 64)] = 0.f;

  // software-pipelined gather: xr = observation chunks of the tile about to be processed
  f32x4 xr[NG];
  {
    const int row0 = seq * GR;
#pragma unroll
    for (int u = 0; u < NG; ++u) {
      const int i = tid0 + u * 128, rr = i / per, c = i - rr * per;
      xr[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (i < GR * per && row0 + rr < cnt)
        xr[u] = ldg16(a.obs, (unsigned)a.rows[row0 + rr] * (unsigned)(DP * 4) + (unsigned)(c * 16));
    }
  }
  int lsrc = -1;  // loss stage: storage row of this lane's minibatch row (wave 0), fetched one tile ahead
  if (wave == 0 && seq * GR + (lane0 & 31) < cnt) lsrc = a.rows[seq * GR + (lane0 & 31)];
  __syncthreads();

#ifdef MOBROB_PAIR_STAMPS  // diagnostic build: cycles per phase, summed over the tiles of every wave of role `wave`
  unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_prev = __builtin_readcyclecounter();
  const unsigned long long st_c0 = st_prev, st_r0 = __builtin_amdgcn_s_memrealtime();
#define PSTAMP(k) { const unsigned long long now_ = __builtin_readcyclecounter(); st_acc[k] += now_ - st_prev; st_prev = now_; }
#else
#define PSTAMP(k)
#endif
  for (int it = 0, tile = seq; it < (PAIR_ON(256) ? niter : 0); ++it, tile += nseq) {  // a tile >= ntiles has no live row: it adds exact zeros
    const int lane = opaque(lane0) & 63;
    const int r = lane & 31, h = lane >> 5;
    const int row0 = tile * GR, nrow0 = (tile + nseq) * GR;
    // ---- this tile's observation rows -> LDS; row indices of the next tile ----
    int nsrc[NG];
#pragma unroll
    for (int u = 0; u < NG; ++u) {
      const int i = tid0 + u * 128, rr = i / per, c = i - rr * per;
      if (i < GR * per) *reinterpret_cast<f32x4*>(&lds[lb + L::X + rr * ldx + 4 * c]) = xr[u];
      nsrc[u] = (i < GR * per && nrow0 + rr < cnt) ? a.rows[nrow0 + rr] : -1;
    }
    // layer-1 fragments of this tile (L2; the other pairs of the CU cover the latency)
    if (!kSmall) f1 = load_frags<NKG1>(W.W1f + (size_t)wave * NKG1 * 64, lane);
    if (!kF2Resident) f2 = load_frags<8>(W.W2f + (size_t)wave * 8 * 64, lane);
    const bool llive = row0 + r < cnt;
    __syncthreads();
    PSTAMP(0)

    {  // layer 1: column block `wave`
      f32x16 c = splat16(bias1);
      gemm_one<ldx, NKG1>(lb + L::X, f1, c, lane);
      const int o = opaque(lb + L::H1 + 4 * h * GLDH + 32 * wave + r);
#pragma unroll
      for (int i = 0; i < 16; ++i) lds[o + crc(i) * GLDH] = PAIR_ON(8) ? fast_tanh_scaled(c[i]) : 0.25f * c[i];
    }
    PSTAMP(12)  // layer 1 alone
    // operands of the loss stage (wave 0, two lanes per row; the row index came one tile ahead): in flight during layer 2 and the head
    // All of them are UNCONDITIONAL loads from in-bounds addresses (row 0 for a dead lane, the last action for a padded
    // one): a select on a loaded value would make the wave wait for HBM right here; the loss stage masks by itself.
    float l_adv = 0.f, l_old = 0.f, l_act[NJ];
    int lsrc_next = 0;
    if (wave == 0 && PAIR_ON(1)) {
      const unsigned src = llive ? (unsigned)lsrc : 0u;
      if (net == 0) {
        const float* arow = a.actions + (size_t)src * a.A;
        const int amax = a.A - 1;
#pragma unroll
        for (int j = 0; j < NJ; ++j) l_act[j] = (2 * j < a.A) ? arow[min(2 * j + h, amax)] : 0.f;  // wave-uniform condition
        l_adv = a.adv[src];
        l_old = a.old_logp[src];
      } else {
#pragma unroll
        for (int j = 0; j < NJ; ++j) l_act[j] = 0.f;
        l_old = a.ret[src];
        if (a.clip_vf >= 0.f) l_adv = a.old_values[src];
      }
      lsrc_next = a.rows[min(nrow0 + r, a.count - 1)];
    } else {
#pragma unroll
      for (int j = 0; j < NJ; ++j) l_act[j] = 0.f;
    }
    if (!kFhResident) load_fh(lane);
    PSTAMP(1)
    __syncthreads();
    PSTAMP(2)
    {  // layer 2
      f32x16 c = splat16(bias2);
      gemm_one<GLDH, 8>(lb + L::H1, f2, c, lane);
      const int o = opaque(lb + L::H2 + 4 * h * GLDH + 32 * wave + r);
#pragma unroll
      for (int i = 0; i < 16; ++i) lds[o + crc(i) * GLDH] = PAIR_ON(8) ? fast_tanh_scaled(c[i]) : 0.25f * c[i];
    }
    __syncthreads();
    PSTAMP(3)  // barrier after layer 2
    if constexpr (kH16) {  // head rows 16 wave .. 16 wave + 15, full K, two accumulation chains
      f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = acc;
      const int i16 = lane & 15, g = lane >> 4;
      const int ab = 4 * opaque((lb + L::H2 + (16 * wave + i16) * GLDH + 4 * g) >> 2);
#pragma unroll
      for (int kg = 0; kg < 4; kg += 2) {
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(&lds[ab + 16 * kg]);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(&lds[ab + 16 * kg + 16]);
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
          acc = MFMA16(a0[s_], fh.f[kg][s_], acc);
          acc2 = MFMA16(a1[s_], fh.f[kg + 1][s_], acc2);
        }
      }
      const int o = opaque(lb + L::DO + (16 * wave + 4 * g) * FLDO + i16);
#pragma unroll
      for (int e = 0; e < 4; ++e) lds[o + e * FLDO] = acc[e] + acc2[e];
    } else {  // head: wave 0 the even k-groups, wave 1 the odd ones; the loss stage adds the two partial tiles
      f32x16 acc = zero16();
      const int ab = 4 * opaque((lb + L::H2 + r * GLDH + 4 * h) >> 2);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(&lds[ab + (2 * j + wave) * 8]);
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) acc = MFMA32(av[s_], fh.f[j][s_], acc);
      }
      const int o = opaque((wave == 0 ? lb + L::DO : lb + L::DO2) + 4 * h * FLDO + r);
#pragma unroll
      for (int i = 0; i < 16; ++i) lds[o + crc(i) * FLDO] = acc[i];
    }
    PSTAMP(4)  // head
    __syncthreads();
    PSTAMP(5)
    if (wave == 0 && PAIR_ON(4)) {  // loss: two lanes per row (q = action parity); dL/d(head) -> DO, zero padded   (tile64_train's stage)
      const int rr = r, q = h;
      const bool live = llive;
      const int db = opaque(lb + L::DO + rr * FLDO + q);
      const int d2 = opaque(lb + L::DO2 + rr * FLDO + q);
      const int cb = opaque(lb + L::CST + q);
      const int gb = opaque(lb + L::GACC + q);
      auto IV = [&](int j) { return kCstRegs ? c_iv[kCstRegs ? j : 0] : lds[cb + 2 * j]; };
      auto LC = [&](int j) { return kCstRegs ? c_lc[kCstRegs ? j : 0] : lds[cb + 32 + 2 * j]; };
      auto BB = [&](int j) { return kCstRegs ? c_bb[kCstRegs ? j : 0] : lds[cb + 64 + 2 * j]; };
      const int A = a.A;
      if (net == 0) {
        // Branch-free over the lane's NJ actions: a dead lane (row beyond the minibatch, or the padded action of an odd A)
        // has weight 0, reads valid LDS words and contributes exact zeros.
        float lp = 0.f;
        float dk[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const float on = (2 * j + q < A && live) ? 1.f : 0.f;
          const float d = on * (l_act[j] - ((kH16 ? lds[db + 2 * j] : lds[db + 2 * j] + lds[d2 + 2 * j]) + BB(j)));
          lp += on * (-(d * d) * (0.5f * IV(j)) - LC(j));
          dk[j] = d;
        }
        lp += __shfl_xor(lp, 32, 64);
        PSTAMP(13)  // log-prob of the stored action
        float g_logp = 0.f;
        if (live) {
          float adv = l_adv;
          if (a.normalize && adv_on && PAIR_ON(2)) adv = (adv - adv_mean) / (adv_sd + 1e-8f);
          const float log_ratio = lp - l_old;
          const float ratio = PAIR_ON(2) ? expf(log_ratio) : 1.0f + log_ratio;
          const float lo = 1.0f - a.clip, hi = 1.0f + a.clip;
          const float s1 = adv * ratio, s2 = adv * fminf(fmaxf(ratio, lo), hi);
          if (q == 0) {
            s_pl += fminf(s1, s2);
            s_cf += (fabsf(ratio - 1.0f) > a.clip) ? 1.f : 0.f;
            s_kl += (ratio - 1.0f) - log_ratio;
          }
          const float in_range = (ratio >= lo && ratio <= hi) ? 1.f : 0.f;
          const float w1 = (s1 < s2) ? 1.f : ((s1 > s2) ? 0.f : 0.5f);
          g_logp = -(w1 * adv + (1.0f - w1) * adv * in_range) * a.inv_bg * ratio;
        }
        PSTAMP(14)  // ratio / clip / g_logp
#pragma unroll
        for (int j = 0; j < NJ; ++j) {  // k = 2j + q; the sums over rows are taken once, after the last tile
          const float on = (2 * j + q < A && live) ? 1.f : 0.f;
          const float d = dk[j];
          float gm = on * (g_logp * d * IV(j));
          float gl = on * (g_logp * (d * d * IV(j) - 1.0f));
          lds[db + 2 * j] = gm;
          if (kAccRegs) {
            acc_gm[kAccRegs ? j : 0] += gm;
            acc_gl[kAccRegs ? j : 0] += gl;
          } else if (2 * j < A) {  // wave-uniform: sum over the 32 rows (lanes with equal q) of this tile
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
          value_loss_terms((kH16 ? lds[db] : lds[db] + lds[d2]) + BB(0), l_old, l_adv, a.clip_vf, sq, gv_);
          s_vl += sq;
          dv = a.vf_coef * gv_ * a.inv_bg;
        }
        lds[db] = dv;  // (q == 1 lanes write column 1 = 0: dv is zero there)
        if (kAccRegs) {
          acc_gm[0] += dv;
        } else {
          const float t = wave_sum(dv);
          if (lane == 0) lds[gb] += t;
        }
      }
      lsrc = lsrc_next;
      PSTAMP(15)  // gradient loop + row sums
    }
    // backward weight fragments (L2) and the next tile's observation rows: in flight during the backward phases
    Frags<8> b2;
    if (!kB3Resident) b3 = load_frags<4>(W.W3b + (size_t)wave * 4 * 64, lane, nkh);
    if (!kSmall) b2 = load_frags<8>(W.W2b + (size_t)wave * 8 * 64, lane);
#pragma unroll
    for (int u = 0; u < NG; ++u) {
      const int c = (tid0 + u * 128) % per;
      xr[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (nsrc[u] >= 0 && PAIR_ON(16)) xr[u] = ldg16(a.obs, (unsigned)nsrc[u] * (unsigned)(DP * 4) + (unsigned)(c * 16));
    }
    PSTAMP(6)  // loss (wave 0) + backward fragment / next-row loads
    __syncthreads();
    PSTAMP(7)

    {  // dW3 (own h2 block), then dh2 -> dz2 over the own block of h2 (nobody else reads that block)
      const int ao = opaque(lb + L::DO + h * FLDO + r);
      const int bo = opaque(lb + L::H2 + h * GLDH + 32 * wave + r);
      if constexpr (kH16) {  // lane group g = lane >> 4 takes batch rows kk + 4 g: all 32 rows in eight k-steps of four
        const int i16 = lane & 15, g4 = 4 * (lane >> 4);
        const int a16 = opaque(lb + L::DO + g4 * FLDO + i16);               // A[i = a][k = row] = dout[row][a]
        const int b16 = opaque(lb + L::H2 + g4 * GLDH + 32 * wave + i16);   // B[k = row][j] = h2[row][32 wave + 16 b + j]
#pragma unroll 4
        for (int t = 0; t < (PAIR_ON(32) ? 8 : 0); ++t) {
          const int kk = (t >> 2) * 16 + (t & 3);
          const float x = lds[a16 + kk * FLDO];
          gW3h0 = MFMA16(x, lds[b16 + kk * GLDH], gW3h0);
          gW3h1 = MFMA16(x, lds[b16 + kk * GLDH + 16], gW3h1);
        }
      } else {
#pragma unroll 4
        for (int k = 0; k < (PAIR_ON(32) ? GR : 0); k += 2) gW3 = MFMA32(lds[ao + k * FLDO], lds[bo + k * GLDH], gW3);
      }
      f32x16 c = zero16();
      gemm_one<FLDO, 4>(lb + L::DO, b3, c, lane, nkh);
      const int o = opaque(lb + L::H2 + 4 * h * GLDH + 32 * wave + r);
      // bias gradient of layer 2 (column 32*wave + r): the lane holds 16 of the column's 32 rows of dz2 in registers
      // right here -- summed in four chains as they are produced, the two row halves added across lanes; no LDS re-read
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float hv = PAIR_ON(128) ? lds[o + crc(i) * GLDH] : 0.5f;
        const float dz = c[i] * (1.0f - hv * hv);
        lds[o + crc(i) * GLDH] = dz;
        if (PAIR_ON(64)) { if ((i & 3) == 0) s0 += dz; else if ((i & 3) == 1) s1 += dz; else if ((i & 3) == 2) s2 += dz; else s3 += dz; }
      }
      float s = (s0 + s1) + (s2 + s3);
      s += __shfl_xor(s, 32, 64);
      gb2 += s;
    }
    PSTAMP(8)
    __syncthreads();
    PSTAMP(9)
    {  // dW2: (dz2 block 0, dz2 block 1) x own h1 block; then dh1 -> dz1 over the own block of h1; dW1 from it
      const int ao = opaque(lb + L::H2 + h * GLDH + r);
      const int bo = opaque(lb + L::H1 + h * GLDH + 32 * wave + r);
#pragma unroll 4
      for (int k = 0; k < (PAIR_ON(32) ? GR : 0); k += 2) {
        const float y = lds[bo + k * GLDH];
        gW2a = MFMA32(lds[ao + k * GLDH], y, gW2a);
        gW2b = MFMA32(lds[ao + k * GLDH + 32], y, gW2b);
      }
      f32x16 c = zero16();
      if (kSmall) {  // B operand from the LDS pack
        const int ab = 4 * opaque((lb + L::H2 + r * GLDH + 4 * h) >> 2);
        const int bb = 4 * opaque(((W2B_LDS + wave * 8 * 256) >> 2) + lane);
#pragma unroll
        for (int kg = 0; kg < 8; ++kg) {
          const f32x4 u = *reinterpret_cast<const f32x4*>(&lds[ab + 8 * kg]);
          const f32x4 w = *reinterpret_cast<const f32x4*>(&lds[bb + kg * 256]);
#pragma unroll
          for (int s_ = 0; s_ < 4; ++s_) c = MFMA32(u[s_], w[s_], c);
        }
      } else {
        gemm_one<GLDH, 8>(lb + L::H2, b2, c, lane);
      }
      const int o = opaque(lb + L::H1 + 4 * h * GLDH + 32 * wave + r);
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;  // bias gradient of layer 1, from registers like layer 2's
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float hv = lds[o + crc(i) * GLDH];
        const float dz = c[i] * (1.0f - hv * hv);
        lds[o + crc(i) * GLDH] = dz;
        if (PAIR_ON(64)) { if ((i & 3) == 0) s0 += dz; else if ((i & 3) == 1) s1 += dz; else if ((i & 3) == 2) s2 += dz; else s3 += dz; }
      }
      const int ob = opaque(lb + L::H1 + h * GLDH + 32 * wave + r);
      float s = (s0 + s1) + (s2 + s3);
      s += __shfl_xor(s, 32, 64);
      gb1 += s;
      // dW1: own dz1 block x (X columns 0..31 | 32..)
      const int c0 = (r < DP) ? r : 0;
      const int c1 = (32 + r < DP) ? 32 + r : c0;
      const int x0 = opaque(lb + L::X + h * ldx + c0), x1 = opaque(lb + L::X + h * ldx + c1);
#pragma unroll 4
      for (int k = 0; k < (PAIR_ON(32) ? GR : 0); k += 2) {
        const float xv = lds[ob + k * GLDH];
        gW1a = MFMA32(xv, lds[x0 + k * ldx], gW1a);
        if (two) gW1c = MFMA32(xv, lds[x1 + k * ldx], gW1c);
      }
    }
    PSTAMP(10)
    __syncthreads();  // X / h1 / h2 / head tiles are rewritten by the next tile
    PSTAMP(11)
  }

#ifdef MOBROB_PAIR_STAMPS
  if (wave == 1) {  // role 1 has no loss stamps: slots 13 / 14 carry shader-clock cycles and 100 MHz ticks of the tile loop
    st_acc[13] = __builtin_readcyclecounter() - st_c0;
    st_acc[14] = __builtin_amdgcn_s_memrealtime() - st_r0;
  }
  if (lane0 == 0 && pair_on)
    for (int k = 0; k < 16; ++k) atomicAdd(&a.stamps[16 * wave + k], st_acc[k]);
#endif
  // ---- ONE slab per workgroup (fragment order of k_fused64_train; tiles t = ib*2 + jb): the second pair lays its
  //      contribution out as a slab in LDS (the tile regions are dead), the first pair adds its own on top and stores.
  //      A pair without tiles contributes exact zeros. ----
  const int lane = lane0;
  float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f, out_m = 0.f, out_l = 0.f;
  if (wave == 0) {
    t0 = wave_sum(s_pl); t1 = wave_sum(s_vl); t2 = wave_sum(s_kl); t3 = wave_sum(s_cf);
    // head-bias / log_std gradients -> lane (r, q) holds action k = 2r + q (r < 16)
    const int r = lane & 31, q = lane >> 5;
    if (kAccRegs) {  // the lane's running sums over its rows, reduced over the 32 lanes of equal parity
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        float gm = acc_gm[j], gl = acc_gl[j];
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) {
          gm += xor_lane(gm, o);
          gl += xor_lane(gl, o);
        }
        if (r == j) { out_m = gm; out_l = gl; }
      }
    } else if (r < 16) {
      out_m = lds[lb + L::GACC + 2 * r + q];
      out_l = lds[lb + L::GACC + 32 + 2 * r + q];
    }
  }
  __syncthreads();  // every wave has read what it needs from its tile region
  if constexpr (kH16) {  // dW3 from its two 16x16 tiles into the 32x32 tile layout the slab uses (rows 16..31 are zero)
    float* sc = &lds[lb + L::H1 + wave * (16 * 33)];  // [16 head rows][32 own columns], private to the wave
    const int i16 = lane & 15, g = lane >> 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      sc[(4 * g + e) * 33 + i16] = gW3h0[e];
      sc[(4 * g + e) * 33 + 16 + i16] = gW3h1[e];
    }
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = crc(i) + 4 * h;        // compile-time crc(i): rows 0..3 | 8..11 | 16.. | 24.. (+4 for the upper lanes)
      gW3[i] = crc(i) < 16 ? sc[row * 33 + r] : 0.f;
    }
    __syncthreads();  // the staging below overwrites these scratch rows
  }
  float* stage = &lds[0];  // [s64_size()] floats: spans the first pair's region and the head of the second one's
  float* slab = a.slabs + (size_t)(2 * bseq + net) * s64_size();
  // Each pair FINALISES half of the slab: pair 0 the dW2 tiles of dz2 block 0, the dW1 tiles of input block 0 and the
  // small vectors; pair 1 the dW2 tiles of dz2 block 1, the second dW1 tiles and dW3.  Pass 0: every pair lays the half it
  // does NOT finalise out in LDS (slab layout); pass 1: it adds what the other pair staged to its own registers and stores.
  // Both pairs work in both passes (round 2: the second pair staged everything while the first waited, then the first
  // added and stored everything: 5.5 us of a 12 us fixed cost per launch).  own + staged is the same sum as before.
  auto put = [&](bool stage_it, int region, int t, const f32x16& acc) {
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      const int o = region + ((t * 4 + qd) * 64 + lane) * 4;
      f32x4 v = {acc[4 * qd], acc[4 * qd + 1], acc[4 * qd + 2], acc[4 * qd + 3]};
      if (stage_it) {
        *reinterpret_cast<f32x4*>(&stage[o]) = v;
      } else {
        const f32x4 w = *reinterpret_cast<const f32x4*>(&stage[o]);
        stg16(slab, (unsigned)o * 4u, f32x4{v[0] + w[0], v[1] + w[1], v[2] + w[2], v[3] + w[3]});
      }
    }
  };
  auto put1 = [&](bool stage_it, int o, float v) {
    if (stage_it) stage[o] = v;
    else slab[o] = v + stage[o];
  };
#pragma unroll
  for (int pass = 0; pass < (PAIR_ON(512) ? 2 : 0); ++pass) {
    if (pass == 1) __syncthreads();
    // half A (finalised by pair 0) is staged by pair 1 in pass 0; half B the other way round
    const bool doA = (pass == 0) == (pr == 1), st = pass == 0;
    if (doA) {
      put(st, s64_w2(), 0 * 2 + wave, gW2a);   // dW2[n][j]: neuron block ib = which dz2 block (a: 0, b: 1), input block jb = wave
      put(st, s64_w1(), wave * 2 + 0, gW1a);   // dW1[n][j]: neuron block ib = wave, input block jb (a: 0, c: 1)
      if (lane < 32) {
        put1(st, s64_b2() + 32 * wave + lane, gb2);
        put1(st, s64_b1() + 32 * wave + lane, gb1);
      }
      if (wave == 0) {
        if (lane < 4) put1(st, s64_st() + lane, lane == 0 ? t0 : (lane == 1 ? t1 : (lane == 2 ? t2 : t3)));
        if ((lane & 31) < 16) {
          put1(st, s64_b3() + 2 * (lane & 31) + (lane >> 5), out_m);
          put1(st, s64_ls() + 2 * (lane & 31) + (lane >> 5), out_l);
        }
      }
    } else {
      put(st, s64_w2(), 1 * 2 + wave, gW2b);
      if (two) put(st, s64_w1(), wave * 2 + 1, gW1c);
      put(st, s64_w3(), wave, gW3);
    }
  }
}

}  // namespace mobrob
