// Device-side helpers shared by all kernels: MFMA types, wave reductions, Philox4x32-10, Feistel.
// gfx950 only: wavefront = 64 lanes, hard-coded.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mobrob {

constexpr int kMaxHidden = 8;   // hidden layers per network the generic chain accepts (the fused kernel families are two-layer)


typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kWave = 64;

// Value of lane (l ^ o) for o = 1, 2, 4, 8 as DPP moves (one or two VALU instructions) instead of ds_bpermute_b32 (an LDS
// round trip each: hipcc lowers every __shfl_xor to it, and a butterfly is a chain of dependent ones).  Exactly the
// xor exchange -- same partner, same bits -- so every reduction built on it keeps its summation order.
//   quad_perm [1,0,3,2] / [2,3,0,1];  xor 4: row_shl:4 into banks 0,2 + row_shr:4 into banks 1,3;  xor 8: row_ror:8.
__device__ __forceinline__ int xor_lane_i(int v, int o) {
  switch (o) {
    case 1: return __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false);
    case 2: return __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false);
    case 4: {
      const int t = __builtin_amdgcn_update_dpp(v, v, 0x104, 0xF, 0x5, false);
      return __builtin_amdgcn_update_dpp(t, v, 0x114, 0xF, 0xA, false);
    }
    case 8: return __builtin_amdgcn_update_dpp(v, v, 0x128, 0xF, 0xF, false);
    default: return __shfl_xor(v, o, 64);
  }
}
__device__ __forceinline__ float xor_lane(float v, int o) { return __int_as_float(xor_lane_i(__float_as_int(v), o)); }
__device__ __forceinline__ double xor_lane(double v, int o) {
  const long long b = __double_as_longlong(v);
  const int lo = xor_lane_i((int)(b & 0xffffffffll), o), hi = xor_lane_i((int)(b >> 32), o);
  return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned)lo);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += xor_lane(v, o);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += xor_lane(v, o);
  return v;
}

// Block-wide sum of one float per thread; result valid in thread 0.  `scratch` >= 16 floats of LDS.
__device__ __forceinline__ float block_sum(float v, float* scratch) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) scratch[w] = v;
  __syncthreads();
  float r = 0.f;
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    for (int i = 0; i < nw; ++i) r += scratch[i];
  }
  return r;
}
__device__ __forceinline__ double block_sum_d(double v, double* scratch) {
  v = wave_sum_d(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) scratch[w] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    for (int i = 0; i < nw; ++i) r += scratch[i];
  }
  return r;
}

// value-loss term of one row [SB3 PPO.train]: without value clipping (clip_vf < 0, SB3's clip_range_vf = None)
// sq = (ret - v)^2; with it the prediction is old_v + clamp(v - old_v, -clip_vf, clip_vf) and the gradient passes
// where the clamp does (torch: inclusive at both ends).  g = d sq / d v.
__device__ __forceinline__ void value_loss_terms(float v, float rt, float old_v, float clip_vf, float& sq, float& g) {
  float vp = v, pass = 1.f;
  if (clip_vf >= 0.f) {
    const float d = v - old_v;
    vp = old_v + fminf(fmaxf(d, -clip_vf), clip_vf);
    pass = (d >= -clip_vf && d <= clip_vf) ? 1.f : 0.f;
  }
  sq = (rt - vp) * (rt - vp);
  g = 2.0f * (vp - rt) * pass;
}

// ---- Philox4x32-10 (Salmon et al. 2011); counter-based, restated bit-exactly in tests ----------
struct Philox4 {
  uint32_t x, y, z, w;
};
__host__ __device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                          uint32_t k0, uint32_t k1) {
  constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  return Philox4{c0, c1, c2, c3};
}
__device__ __forceinline__ float u32_to_unit_open(uint32_t x) {  // (0,1)
  // explicit roundings: every kernel must map a draw to the same float (no context-dependent fma contraction)
  return __fmul_rn(__fadd_rn((float)(x >> 8), 0.5f), 1.0f / 16777216.0f);
}
// 4 uniforms -> 4 standard normals (Box-Muller) on the transcendental hardware: v_log_f32 (log2), v_sqrt_f32 and
// v_sin_f32 / v_cos_f32, which take their argument in REVOLUTIONS -- cos(2 pi u) is v_cos_f32(u) with no range
// reduction.  ~20 instructions instead of ~300 for the libm path; the rollout kernels draw 3-5 of these per
// thread and step.  Absolute error ~1e-6: irrelevant for a noise source (tests check the moments).
__device__ __forceinline__ void box_muller4(const Philox4& r, float out[4]) {
  const float u0 = u32_to_unit_open(r.x), u1 = u32_to_unit_open(r.y);
  const float u2 = u32_to_unit_open(r.z), u3 = u32_to_unit_open(r.w);
  constexpr float kM2Ln2 = -1.38629436111989061883f;  // -2 ln 2: -2 ln u = kM2Ln2 * log2 u
  const float ra = __builtin_amdgcn_sqrtf(kM2Ln2 * __builtin_amdgcn_logf(u0));
  const float rb = __builtin_amdgcn_sqrtf(kM2Ln2 * __builtin_amdgcn_logf(u2));
  const float c0 = __builtin_amdgcn_cosf(u1), s0 = __builtin_amdgcn_sinf(u1);
  const float c1 = __builtin_amdgcn_cosf(u3), s1 = __builtin_amdgcn_sinf(u3);
  out[0] = ra * c0; out[1] = ra * s0; out[2] = rb * c1; out[3] = rb * s1;
}

// ---- Feistel permutation (oracle/ppo_oracle.py:feistel_permutation, bit-exact) ------------------
__host__ __device__ __forceinline__ uint32_t feistel_rf(uint32_t r, int rnd, uint32_t k0, uint32_t k1,
                                                        uint32_t mask) {
  uint32_t x = r + 0x9E3779B9u * (uint32_t)(rnd + 1) + k0;
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= k1; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  return x & mask;
}
__host__ __device__ __forceinline__ uint64_t feistel_perm(uint64_t i, uint64_t n, int half_bits, uint32_t k0,
                                                          uint32_t k1) {
  const uint32_t mask = (uint32_t)((1ull << half_bits) - 1ull);
  uint64_t v = i;
  do {
    uint32_t l = (uint32_t)(v >> half_bits) & mask, r = (uint32_t)v & mask;
#pragma unroll
    for (int rnd = 0; rnd < 6; ++rnd) {
      const uint32_t t = l ^ feistel_rf(r, rnd, k0, k1, mask);
      l = r; r = t;
    }
    v = ((uint64_t)l << half_bits) | r;
  } while (v >= n);
  return v;
}
// inverse of feistel_perm: the position i with feistel_perm(i) == v (rounds run backwards; the cycle walk of a value < n
// through the inverse round function visits the same cycle in the opposite direction)
__host__ __device__ __forceinline__ uint64_t feistel_perm_inv(uint64_t v, uint64_t n, int half_bits, uint32_t k0,
                                                              uint32_t k1) {
  const uint32_t mask = (uint32_t)((1ull << half_bits) - 1ull);
  do {
    uint32_t l = (uint32_t)(v >> half_bits) & mask, r = (uint32_t)v & mask;
#pragma unroll
    for (int rnd = 5; rnd >= 0; --rnd) {
      const uint32_t pl = r ^ feistel_rf(l, rnd, k0, k1, mask);
      r = l; l = pl;
    }
    v = ((uint64_t)l << half_bits) | r;
  } while (v >= n);
  return v;
}
// the same in 32-bit arithmetic for domains of at most 2^30 positions (half_bits <= 15): what the streaming
// advantage-statistics pass runs per element
__host__ __device__ __forceinline__ uint32_t feistel_perm_inv32(uint32_t v, uint32_t n, int half_bits, uint32_t k0,
                                                                uint32_t k1) {
  const uint32_t mask = (1u << half_bits) - 1u;
  do {
    uint32_t l = (v >> half_bits) & mask, r = v & mask;
#pragma unroll
    for (int rnd = 5; rnd >= 0; --rnd) {
      const uint32_t pl = r ^ feistel_rf(l, rnd, k0, k1, mask);
      r = l; l = pl;
    }
    v = (l << half_bits) | r;
  } while (v >= n);
  return v;
}
__host__ __device__ inline int feistel_half_bits(uint64_t n) {
  int bits = 0;
  uint64_t m = n > 1 ? n - 1 : 1;
  while (m) { ++bits; m >>= 1; }
  if (bits < 2) bits = 2;
  if (bits & 1) ++bits;
  return bits / 2;
}

}  // namespace mobrob
