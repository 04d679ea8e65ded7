// EXPERIMENT (not in the product build unless -DMOBROB_VALUE8): the forward pass of the 2x256 value network with
// EIGHT waves per workgroup -- two per SIMD -- instead of four.  Wave w owns 32 output columns of every hidden
// layer (one 32-column block, two 32-row blocks: 2 accumulators instead of 4), so that one wave's tanh epilogue,
// LDS traffic and operand waits run under the other wave's MFMAs on the same SIMD.  Each output element sees the
// same k order as in the four-wave kernel: results are bit-identical (scratch/value8.py checks that).
// This is the forward third of the two-waves-per-SIMD training kernel planned in DESIGN.md section 7.
#pragma once
#include "../mobrob_amd/csrc/kernels_fused.h"

namespace mobrob {

constexpr int FTHREADS8 = 512;

#define MFMA_KG_C1(u, v, p)                          \
  _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) { \
    c0 = MFMA32(u[s_], p[s_], c0);                   \
    c1 = MFMA32(v[s_], p[s_], c1);                   \
  }

// acc[rb] += A[rb*32 + 0..31][0..8*nkg) . Bpacked for this wave's ONE 32-column block (nkg even)
template <int LDA>
__device__ __forceinline__ void gemm_lds_packed_c1(int a_off, const f32x4* __restrict__ Bp, int nkg, f32x16& c0,
                                                   f32x16& c1, int lane, const f32x4& first) {
  const int r = lane & 31, h = lane >> 5;
  const int ab = 4 * opaque((a_off + r * LDA + 4 * h) >> 2);
  unsigned bo = opaque_u((unsigned)lane * 16u);
  f32x4 pA = first, pB;
  f32x4 uA = *reinterpret_cast<const f32x4*>(&lds[ab]);
  f32x4 vA = *reinterpret_cast<const f32x4*>(&lds[ab + 32 * LDA]);
  f32x4 uB, vB;
  int ao = ab;
#pragma unroll 1
  for (int kg = 0; kg < nkg - 2; kg += 2) {
    pB = ldg16(Bp, bo + 1024u);
    uB = *reinterpret_cast<const f32x4*>(&lds[ao + 8]);
    vB = *reinterpret_cast<const f32x4*>(&lds[ao + 32 * LDA + 8]);
    MFMA_KG_C1(uA, vA, pA)
    pA = ldg16(Bp, bo + 2048u);
    uA = *reinterpret_cast<const f32x4*>(&lds[ao + 16]);
    vA = *reinterpret_cast<const f32x4*>(&lds[ao + 32 * LDA + 16]);
    MFMA_KG_C1(uB, vB, pB)
    bo += 2048u;
    ao += 16;
  }
  pB = ldg16(Bp, bo + 1024u);
  uB = *reinterpret_cast<const f32x4*>(&lds[ao + 8]);
  vB = *reinterpret_cast<const f32x4*>(&lds[ao + 32 * LDA + 8]);
  MFMA_KG_C1(uA, vA, pA)
  MFMA_KG_C1(uB, vB, pB)
}

// K = 256 contraction, weight fragments three k-groups ahead (nkg % 4 == 0, nkg >= 8)
template <int LDA>
__device__ __forceinline__ void gemm_lds_packed_deep_c1(int a_off, const f32x4* __restrict__ Bp, int nkg, f32x16& c0,
                                                        f32x16& c1, int lane, const f32x4& first) {
  const int r = lane & 31, h = lane >> 5;
  const int ab = 4 * opaque((a_off + r * LDA + 4 * h) >> 2);
  unsigned bo = opaque_u((unsigned)lane * 16u);
  f32x4 p0 = first, p1 = ldg16(Bp, bo + 1024u), p2 = ldg16(Bp, bo + 2048u), p3;
  f32x4 uA = *reinterpret_cast<const f32x4*>(&lds[ab]);
  f32x4 vA = *reinterpret_cast<const f32x4*>(&lds[ab + 32 * LDA]);
  f32x4 uB, vB;
  int ao = ab;
#define LDA_NEXT(U, V, off)                              \
  U = *reinterpret_cast<const f32x4*>(&lds[ao + (off)]); \
  V = *reinterpret_cast<const f32x4*>(&lds[ao + 32 * LDA + (off)]);
#pragma unroll 1
  for (int kg = 0; kg < nkg - 4; kg += 4) {
    p3 = ldg16(Bp, bo + 3072u);
    LDA_NEXT(uB, vB, 8)
    MFMA_KG_C1(uA, vA, p0)
    p0 = ldg16(Bp, bo + 4096u);
    LDA_NEXT(uA, vA, 16)
    MFMA_KG_C1(uB, vB, p1)
    p1 = ldg16(Bp, bo + 5120u);
    LDA_NEXT(uB, vB, 24)
    MFMA_KG_C1(uA, vA, p2)
    p2 = ldg16(Bp, bo + 6144u);
    LDA_NEXT(uA, vA, 32)
    MFMA_KG_C1(uB, vB, p3)
    bo += 4096u;
    ao += 32;
  }
  p3 = ldg16(Bp, bo + 3072u);
  LDA_NEXT(uB, vB, 8)
  MFMA_KG_C1(uA, vA, p0)
  LDA_NEXT(uA, vA, 16)
  MFMA_KG_C1(uB, vB, p1)
  LDA_NEXT(uB, vB, 24)
  MFMA_KG_C1(uA, vA, p2)
  MFMA_KG_C1(uB, vB, p3)
#undef LDA_NEXT
}

// lds[dst][row][32*wave + r] = tanh(acc / kTanhScale) for the wave's 64 x 32 block
__device__ __forceinline__ void store_tanh_c1(int dst_off, int wave, int lane, const f32x16& c0, const f32x16& c1) {
  const int r = lane & 31, h = lane >> 5;
  const int o = opaque(dst_off + 4 * h * FLDH + 32 * wave + r);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    lds[o + crc(i) * FLDH] = fast_tanh_scaled(c0[i]);
    lds[o + (32 + crc(i)) * FLDH] = fast_tanh_scaled(c1[i]);
  }
}

// forward of one 64-row tile through the hidden layers with 8 waves; leaves h1, h2 in LDS; ends with a barrier.
// Returns the first two k-groups of the 16x16x4 head pack (consumed by waves 0..3 in tile_head16).
template <int DP>
__device__ __forceinline__ Frag2 tile_layers8(const FusedNet& W, int wave, int lane, const f32x4& f1) {
  using L = Lay<DP>;
  constexpr int nkg1 = DP / 8, nkg2 = FH / 8;
  const int r_ = lane & 31;
  const f32x4* w2 = W.W2f + (size_t)wave * nkg2 * 64;
  f32x4 f2;
  {
    const float bz = W.b1s[32 * wave + r_];
    f32x16 c0 = splat16(bz), c1 = splat16(bz);
    gemm_lds_packed_c1<L::LDX>(L::X, W.W1f + (size_t)wave * nkg1 * 64, nkg1, c0, c1, lane, f1);
    f2 = ldg16(w2, opaque_u((unsigned)lane * 16u));
    store_tanh_c1(L::H1, wave, lane, c0, c1);
  }
  __syncthreads();
  Frag2 f3;
  {
    const float bz = W.b2s[32 * wave + r_];
    f32x16 c0 = splat16(bz), c1 = splat16(bz);
    gemm_lds_packed_deep_c1<FLDH>(L::H1, w2, nkg2, c0, c1, lane, f2);
    f3 = prefetch_frag(W.W3h, W.W3h + 64, lane);
    store_tanh_c1(L::H2, wave, lane, c0, c1);
  }
  __syncthreads();
  return f3;
}

template <int DP>
__global__ __launch_bounds__(FTHREADS8, 1) void k_value_batch8(FusedNet W, const float* __restrict__ X, int rows,
                                                               float* __restrict__ v) {
  using L = Lay<DP>;
  constexpr int ldx = L::LDX, per = DP / 4;
  constexpr int NG = (FR * per + FTHREADS8 - 1) / FTHREADS8;
  const int tid0 = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int ntiles = (rows + FR - 1) / FR;
  f32x4 xr[NG];
#pragma unroll
  for (int u = 0; u < NG; ++u) {
    const int i = tid0 + u * FTHREADS8, rr = i / per, c = i - rr * per;
    xr[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (i < FR * per && (int)blockIdx.x < ntiles && blockIdx.x * FR + rr < rows)
      xr[u] = ldg16(X, (unsigned)(blockIdx.x * FR + rr) * (unsigned)(DP * 4) + (unsigned)(c * 16));
  }
  const float bv = W.b3[0];
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int tid = opaque(tid0), lane = tid & 63;
    const f32x4 f1 = ldg16(W.W1f + (size_t)wave * (DP / 8) * 64, opaque_u((unsigned)lane * 16u));
#pragma unroll
    for (int u = 0; u < NG; ++u) {
      const int i = tid + u * FTHREADS8, rr = i / per, c = i - rr * per;
      if (i < FR * per) *reinterpret_cast<f32x4*>(&lds[L::X + rr * ldx + 4 * c]) = xr[u];
    }
    const int nt = tile + gridDim.x;
#pragma unroll
    for (int u = 0; u < NG; ++u) {
      const int i = tid + u * FTHREADS8, rr = i / per, c = i - rr * per;
      xr[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (i < FR * per && nt < ntiles && nt * FR + rr < rows)
        xr[u] = ldg16(X, (unsigned)(nt * FR + rr) * (unsigned)(DP * 4) + (unsigned)(c * 16));
    }
    __syncthreads();
    const Frag2 f3 = tile_layers8<DP>(W, wave, lane, f1);
    if (wave < 4) {  // 16x16x4 head: wave w owns rows 16w..16w+15 over the full K
      tile_head16<DP>(W, wave, lane, f3);
      if (lane < 16) {
        const int rr = 16 * wave + lane, row = tile * FR + rr;
        if (row < rows) v[row] = lds[L::DO + rr * FLDO] + bv;
      }
    }
    __syncthreads();  // X / h1 / h2 / head tile are rewritten by the next tile
  }
}

}  // namespace mobrob
