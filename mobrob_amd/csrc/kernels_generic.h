// Generic-shape kernels of the PPO hot path (any hidden width that is a multiple of 8).
// The MLP contractions run on the f32-input matrix cores (v_mfma_f32_32x32x2_f32: exact f32, bitwise
// a k-ordered fmaf chain) with fragments loaded straight from L2/HBM; the fused fast path for the
// benchmark shape lives in kernels_fused.h.  All kernels are gfx950-only (wave64).
#pragma once
#include "device_utils.h"

namespace mobrob {

// ------------------------------------------------------------------------------------------------
// Hidden activations of the generic chain (`policy_kwargs.activation_fn`, SB3 MlpExtractor; the codes are MOBROB_ACT_* of
// include/mobrob_ppo.h).  Every one of them has a derivative that is a function of its OUTPUT h = f(z), so the backward
// epilogue needs the stored activations only (no pre-activations are kept):
//   tanh       1 - h^2                        relu        [h > 0]   (torch threshold_backward: 0 at z = 0)
//   elu        h > 0 ? 1 : h + 1              leaky_relu  h > 0 ? 1 : 0.01   (torch: slope where z <= 0)
//   sigmoid    h (1 - h)                      softplus    1 - exp(-h)        (= sigmoid(z); torch beta 1, linear above 20)
//   softsign   (1 - |h|)^2                    hardtanh    [-1 < h < 1]
//   relu6      [0 < h < 6]
// SiLU / GELU / Mish are not monotonic: their derivative needs the PRE-activation z.  The training forward leaves z in the layer's
// dz buffer (GemmArgs.Z; the backward epilogue of that layer reads the element it is about to overwrite):
//   silu  s (1 + z (1 - s)), s = sigmoid(z)      gelu  Phi(z) + z phi(z)  (torch's default, exact erf form)
//   mish  t + z s (1 - t^2), t = tanh(softplus(z))
// ------------------------------------------------------------------------------------------------
enum { ACT_TANH = 0, ACT_RELU = 1, ACT_ELU = 2, ACT_LEAKY_RELU = 3, ACT_SIGMOID = 4, ACT_SOFTPLUS = 5, ACT_SOFTSIGN = 6,
       ACT_HARDTANH = 7, ACT_RELU6 = 8, ACT_SILU = 9, ACT_GELU = 10, ACT_MISH = 11, ACT_COUNT = 12 };
__host__ __device__ __forceinline__ bool act_needs_z(int act) { return act >= ACT_SILU; }
__device__ __forceinline__ float act_fwd(int act, float z) {
  switch (act) {
    case ACT_TANH: return tanhf(z);
    case ACT_RELU: return fmaxf(z, 0.f);
    case ACT_ELU: return z > 0.f ? z : expm1f(z);
    case ACT_LEAKY_RELU: return z > 0.f ? z : 0.01f * z;
    case ACT_SIGMOID: return 1.0f / (1.0f + expf(-z));
    case ACT_SOFTPLUS: return z > 20.f ? z : log1pf(expf(z));
    case ACT_SOFTSIGN: return z / (1.0f + fabsf(z));
    case ACT_HARDTANH: return fminf(fmaxf(z, -1.f), 1.f);
    case ACT_RELU6: return fminf(fmaxf(z, 0.f), 6.f);
    case ACT_SILU: return z / (1.0f + expf(-z));
    case ACT_GELU: return 0.5f * z * (1.0f + erff(z * 0.70710678118654752440f));
    default: return z * tanhf(z > 20.f ? z : log1pf(expf(z)));   // ACT_MISH
  }
}
// g * f'(z) for the activations whose derivative is not a function of their output
__device__ __forceinline__ float act_bwd_z(int act, float z, float g) {
  switch (act) {
    case ACT_SILU: { const float s_ = 1.0f / (1.0f + expf(-z)); return g * (s_ * (1.0f + z * (1.0f - s_))); }
    case ACT_GELU: return g * (0.5f * (1.0f + erff(z * 0.70710678118654752440f)) + z * expf(-0.5f * z * z) * 0.39894228040143267794f);
    default: {   // ACT_MISH
      const float t = tanhf(z > 20.f ? z : log1pf(expf(z))), s_ = 1.0f / (1.0f + expf(-z));
      return g * (t + z * s_ * (1.0f - t * t));
    }
  }
}
// g * f'(z) from h = f(z)
__device__ __forceinline__ float act_bwd(int act, float h, float g) {
  switch (act) {
    case ACT_TANH: return g * (1.0f - h * h);
    case ACT_RELU: return h > 0.f ? g : 0.f;
    case ACT_ELU: return h > 0.f ? g : g * (h + 1.0f);
    case ACT_LEAKY_RELU: return h > 0.f ? g : 0.01f * g;
    case ACT_SIGMOID: return g * (h * (1.0f - h));
    case ACT_SOFTPLUS: return g * (1.0f - expf(-h));
    case ACT_SOFTSIGN: { const float u = 1.0f - fabsf(h); return g * (u * u); }
    case ACT_HARDTANH: return (h > -1.f && h < 1.f) ? g : 0.f;
    default: return (h > 0.f && h < 6.f) ? g : 0.f;   // ACT_RELU6
  }
}

// ------------------------------------------------------------------------------------------------
// MFMA GEMM, TM x TN output tiles of 32x32 per wave, 4 waves per block stacked along M.
//   MODE_NT: C[m][n] = sum_k A[m][k] * B[n][k]      (forward:  X . W^T)        K % 8 == 0
//   MODE_NN: C[m][n] = sum_k A[m][k] * B[k][n]      (backward: dY . W)         K % 8 == 0
//   MODE_TN: C[i][j] = sum_k A[k][i] * B[k][j]      (weight grad: dY^T . X), K = batch rows, split over
//                                                    blockIdx.z, accumulated with float atomics
// Fragment maps (cdna_hip_programming.md §3): A operand lane l holds A[i=l&31][k=l>>5], B operand lane l
// holds B[k=l>>5][j=l&31]; C/D: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5).
// The k index inside a group of 8 is permuted (lane half h handles k = kk+4h+s at MFMA s) so that each
// lane fetches its four k values with one 16-byte load; both operands use the same permutation.
// Rows/cols beyond M/N are clamped for loads (they only pollute outputs that are never stored).
// ------------------------------------------------------------------------------------------------
enum { MODE_NT = 0, MODE_NN = 1, MODE_TN = 2 };
enum { EPI_BIAS = 0, EPI_BIAS_TANH = 1, EPI_DTANH_COLSUM = 2, EPI_ATOMIC = 3 };

struct GemmArgs {
  const float* A; const float* B; float* C;
  int M, N, K;
  int lda, ldb, ldc;
  const float* bias;    // EPI_BIAS / EPI_BIAS_TANH: [N] or nullptr
  const float* Hact;    // EPI_DTANH_COLSUM: activation h (same shape as C), ld = ldh
  int ldh;
  float* colsum;        // EPI_DTANH_COLSUM: [N] += column sums of the stored tile (bias gradient)
  int act;              // hidden activation (ACT_*): tanh is SB3's default for MlpPolicy; the others come from policy_kwargs activation_fn
  float* Z; int ldz;    // EPI_BIAS_TANH, act_needs_z: where the pre-activations go (null: not kept).  EPI_DTANH_COLSUM reads them from C
  int kchunk;           // MODE_TN: batch rows per blockIdx.z
  int wn;               // 1: the block's four waves sit side by side along N (they share the A rows: one fetch per block through L1 / the
                        // XCD's L2 instead of one per column block somewhere on the chip); 0: stacked along M (they share B)
};

// one of two argument structs, field by field: wave-uniform selects, the result stays in scalar registers
__device__ __forceinline__ GemmArgs gemm_pick(bool second, const GemmArgs& ga, const GemmArgs& gb) {
  GemmArgs g;
  g.A = second ? gb.A : ga.A; g.B = second ? gb.B : ga.B; g.C = second ? gb.C : ga.C;
  g.M = second ? gb.M : ga.M; g.N = second ? gb.N : ga.N; g.K = second ? gb.K : ga.K;
  g.lda = second ? gb.lda : ga.lda; g.ldb = second ? gb.ldb : ga.ldb; g.ldc = second ? gb.ldc : ga.ldc;
  g.bias = second ? gb.bias : ga.bias; g.Hact = second ? gb.Hact : ga.Hact; g.ldh = second ? gb.ldh : ga.ldh;
  g.colsum = second ? gb.colsum : ga.colsum; g.act = second ? gb.act : ga.act; g.Z = second ? gb.Z : ga.Z; g.ldz = second ? gb.ldz : ga.ldz;
  g.kchunk = second ? gb.kchunk : ga.kchunk; g.wn = second ? gb.wn : ga.wn;
  return g;
}
// One wave's share of problem g: tiles (blockIdx.x, blockIdx.y), batch split zz (MODE_TN).
// TM x TN: 32x32 tiles per wave (1x1, 2x1 or 2x2).  A wave that owns 2x2 tiles feeds sixteen MFMAs from four operand fetches where
// four one-tile waves need eight for the same sixteen: the one-tile form ran at 22 - 34 % of the f32 matrix rate on L1 / L2
// operand traffic (12.8 flop per byte fetched by a block; 25.6 with 2x2), profiles/r6/generic_chain.txt.
template <int MODE, int EPI, int TM, int TN>
__device__ __forceinline__ void gemm_body(const GemmArgs& g, int zz) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int m0 = (g.wn ? (int)blockIdx.x : (int)blockIdx.x * 4 + wv) * (32 * TM);
  const int n0 = (g.wn ? (int)blockIdx.y * 4 + wv : (int)blockIdx.y) * (32 * TN);
  // whole-wave exits (no block-level sync in this kernel): the grid covers the larger of two problems, and the longer batch split
  if (m0 >= g.M || n0 >= g.N || (MODE == MODE_TN ? zz * g.kchunk >= g.K : zz > 0)) return;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // KB k-steps of 8 are fetched before their MFMAs start.  One-tile waves exist for SMALL launches (a 100-row minibatch of the
  // reference's YAMLs, a rollout step): there a wave's whole K loop is a chain of dependent L2 round trips, and four steps in
  // flight make it two round trips for K = 64 instead of eight (profiles/r6/generic_chain.txt).  Fat tiles
  // keep one step in flight (their registers hold accumulators; the launches are large enough to hide latency across waves).
  constexpr int KB = TM * TN == 1 ? 4 : (TM * TN == 2 ? 2 : 1);
  if (MODE == MODE_NT) {
    const float* ap[TM];
    const float* bp[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) ap[i] = g.A + (size_t)min(m0 + 32 * i + r, g.M - 1) * g.lda + 4 * h;
#pragma unroll
    for (int j = 0; j < TN; ++j) bp[j] = g.B + (size_t)min(n0 + 32 * j + r, g.N - 1) * g.ldb + 4 * h;
#pragma unroll 2
    for (int kk = 0; kk < g.K; kk += 8 * KB) {
      f32x4 a[KB][TM], b[KB][TN];
#pragma unroll
      for (int u = 0; u < KB; ++u) {
        const int ku = min(kk + 8 * u, g.K - 8);   // (a step beyond K re-reads the last one; its MFMAs are skipped below)
#pragma unroll
        for (int i = 0; i < TM; ++i) a[u][i] = *reinterpret_cast<const f32x4*>(ap[i] + ku);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[u][j] = *reinterpret_cast<const f32x4*>(bp[j] + ku);
      }
#pragma unroll
      for (int u = 0; u < KB; ++u) {
        if (u > 0 && kk + 8 * u >= g.K) break;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][i][s], b[u][j][s], acc[i][j], 0, 0, 0);
      }
    }
  } else if (MODE == MODE_NN) {
    const float* ap[TM];
    const float* bp[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) ap[i] = g.A + (size_t)min(m0 + 32 * i + r, g.M - 1) * g.lda + 4 * h;
#pragma unroll
    for (int j = 0; j < TN; ++j) bp[j] = g.B + (size_t)(4 * h) * g.ldb + min(n0 + 32 * j + r, g.N - 1);
#pragma unroll 2
    for (int kk = 0; kk < g.K; kk += 8 * KB) {
      f32x4 a[KB][TM];
      float b[KB][TN][4];
#pragma unroll
      for (int u = 0; u < KB; ++u) {
        const int ku = min(kk + 8 * u, g.K - 8);
#pragma unroll
        for (int i = 0; i < TM; ++i) a[u][i] = *reinterpret_cast<const f32x4*>(ap[i] + ku);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const float* bk = bp[j] + (size_t)ku * g.ldb;
          b[u][j][0] = bk[0]; b[u][j][1] = bk[g.ldb]; b[u][j][2] = bk[2 * (size_t)g.ldb]; b[u][j][3] = bk[3 * (size_t)g.ldb];
        }
      }
#pragma unroll
      for (int u = 0; u < KB; ++u) {
        if (u > 0 && kk + 8 * u >= g.K) break;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][i][s], b[u][j][s], acc[i][j], 0, 0, 0);
      }
    }
  } else {  // MODE_TN
    int ai[TM], bj[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) ai[i] = min(m0 + 32 * i + r, g.M - 1);
#pragma unroll
    for (int j = 0; j < TN; ++j) bj[j] = min(n0 + 32 * j + r, g.N - 1);
    const int k0 = zz * g.kchunk, k1 = min(k0 + g.kchunk, g.K);
    for (int kk = k0; kk < k1; kk += 8 * KB) {
      float a[KB][TM][4], b[KB][TN][4];
#pragma unroll
      for (int u = 0; u < KB; ++u)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int k = kk + 8 * u + 2 * s + h;
          const bool ok = k < k1;
          const int kc = ok ? k : k0;
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const float av = g.A[(size_t)kc * g.lda + ai[i]];
            a[u][i][s] = ok ? av : 0.f;   // (rows beyond the chunk contribute zero: their MFMAs need no skipping)
          }
#pragma unroll
          for (int j = 0; j < TN; ++j) b[u][j][s] = g.B[(size_t)kc * g.ldb + bj[j]];
        }
#pragma unroll
      for (int u = 0; u < KB; ++u) {
        if (u > 0 && kk + 8 * u >= k1) break;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][i][s], b[u][j][s], acc[i][j], 0, 0, 0);
      }
    }
  }

#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + 32 * j + r;
    const bool col_ok = col < g.N;
    float csum = 0.f;
    float bias = 0.f;
    if ((EPI == EPI_BIAS || EPI == EPI_BIAS_TANH) && g.bias != nullptr && col_ok) bias = g.bias[col];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (row < g.M && col_ok) {
          float v = acc[i][j][e];
          if (EPI == EPI_BIAS) {
            g.C[(size_t)row * g.ldc + col] = v + bias;
          } else if (EPI == EPI_BIAS_TANH) {
            g.C[(size_t)row * g.ldc + col] = act_fwd(g.act, v + bias);
            if (g.Z != nullptr) g.Z[(size_t)row * g.ldz + col] = v + bias;
          } else if (EPI == EPI_DTANH_COLSUM) {
            if (act_needs_z(g.act)) {
              v = act_bwd_z(g.act, g.C[(size_t)row * g.ldc + col], v);   // C = the layer's dz buffer: it still holds z (training forward)
            } else {
              const float hv = g.Hact[(size_t)row * g.ldh + col];
              v = act_bwd(g.act, hv, v);
            }
            g.C[(size_t)row * g.ldc + col] = v;
            csum += v;
          } else {
            atomicAdd(&g.C[(size_t)row * g.ldc + col], v);
          }
        }
      }
    }
    if (EPI == EPI_DTANH_COLSUM && g.colsum != nullptr) {
      csum += __shfl_xor(csum, 32, 64);
      if (h == 0 && col_ok) atomicAdd(&g.colsum[col], csum);
    }
  }
}
// ONE or TWO problems of the same kind per launch (two != 0: blockIdx.z & 1 selects, the batch split of MODE_TN is blockIdx.z >> 1):
// the policy and the value network run the same sequence of GEMMs on independent data -- pairing them halves the launches of a
// step and doubles the blocks of a launch (headline shape through this chain +24 %).
template <int MODE, int EPI, int TM, int TN>
__global__ __launch_bounds__(256) void k_gemm(GemmArgs ga, GemmArgs gb, int two) {
  const bool second = two != 0 && (blockIdx.z & 1) != 0;
  const int zz = two != 0 ? (int)(blockIdx.z >> 1) : (int)blockIdx.z;
  const GemmArgs g = gemm_pick(second, ga, gb);
  gemm_body<MODE, EPI, TM, TN>(g, zz);
}

// Up to FOUR problems of ANY kind per launch, one tile per wave: a STAGE of the small-minibatch chain -- at a given depth the two
// networks' weight-gradient and input-gradient GEMMs (or their forward GEMMs, a head opposite a hidden layer included) depend on
// earlier stages only.  blockIdx.z runs over the problems' batch splits back to back (zbeg); the kind is a wave-uniform switch over
// the four (mode, epilogue) combinations the chain uses.  For launches whose cost is the launch (profiles/r6/generic_chain.txt).
struct GemmMulti {
  GemmArgs g[4];
  int mode[4], epi[4];
  int zbeg[5];   // problem p owns blockIdx.z in [zbeg[p], zbeg[p + 1])
};
__global__ __launch_bounds__(256) void k_gemm_multi(GemmMulti m) {
  const int z = blockIdx.z;
  const int p = (z >= m.zbeg[1] ? 1 : 0) + (z >= m.zbeg[2] ? 1 : 0) + (z >= m.zbeg[3] ? 1 : 0);
  const GemmArgs g = gemm_pick(p >= 2, gemm_pick(p == 1, m.g[0], m.g[1]), gemm_pick(p == 3, m.g[2], m.g[3]));
  const int mode = p == 0 ? m.mode[0] : (p == 1 ? m.mode[1] : (p == 2 ? m.mode[2] : m.mode[3]));
  const int epi = p == 0 ? m.epi[0] : (p == 1 ? m.epi[1] : (p == 2 ? m.epi[2] : m.epi[3]));
  const int zz = z - (p == 0 ? m.zbeg[0] : (p == 1 ? m.zbeg[1] : (p == 2 ? m.zbeg[2] : m.zbeg[3])));
  if (mode == MODE_NT && epi == EPI_BIAS_TANH) gemm_body<MODE_NT, EPI_BIAS_TANH, 1, 1>(g, zz);
  else if (mode == MODE_NT) gemm_body<MODE_NT, EPI_BIAS, 1, 1>(g, zz);
  else if (mode == MODE_NN) gemm_body<MODE_NN, EPI_DTANH_COLSUM, 1, 1>(g, zz);
  else gemm_body<MODE_TN, EPI_ATOMIC, 1, 1>(g, zz);
}

// ------------------------------------------------------------------------------------------------
// Pack the canonical SB3-ordered parameter vector into zero-padded compute copies.
//   W1 [H][D] -> [H][Dp];  head [A][H] -> [Ap][H] (extra rows zero)
// ------------------------------------------------------------------------------------------------
__global__ void k_pad_rows(const float* __restrict__ src, float* __restrict__ dst, int rows, int cols,
                           int rows_p, int cols_p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows_p * cols_p) return;
  const int r = i / cols_p, c = i - r * cols_p;
  dst[i] = (r < rows && c < cols) ? src[(size_t)r * cols + c] : 0.f;
}

// ------------------------------------------------------------------------------------------------
// Rollout-time sampling epilogue: a = mu + exp(log_std) * eps ; logp = sum_a Normal.log_prob
// [SB3 DiagGaussianDistribution.sample/log_prob; oracle act()].  One thread per env row.
// eps == nullptr -> Philox4x32-10 counter (row, a/4, draw_counter, STREAM_EPS), key = seed.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kStreamEps = 0x45505331u;   // 'EPS1'
constexpr uint32_t kStreamEnvObs = 0x4F425331u;
constexpr uint32_t kStreamEnvTerm = 0x54524D31u;
constexpr uint32_t kStreamEnvMisc = 0x4D495331u;
constexpr float kLogSqrt2Pi = 0.91893853320467274178f;

__global__ void k_sample(const float* __restrict__ mu, int ldmu, const float* __restrict__ log_std,
                         const float* __restrict__ eps, int n, int A, float lo, float hi, uint64_t seed,
                         uint32_t draw_rel, const uint32_t* __restrict__ draw_base, float* __restrict__ act_raw,
                         float* __restrict__ act_clip, float* __restrict__ logp_out, int row0 = 0) {
  // row0: env index of row 0 (the noise of env n is a function of n, not of the launch that samples it)
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n) return;
  const uint32_t draw = draw_rel + (draw_base ? *draw_base : 0u);
  float lp = 0.f;
  float z0 = 0.f, z1 = 0.f, z2 = 0.f, z3 = 0.f;  // (selects, not a dynamically indexed array: that lives in scratch memory)
  for (int a = 0; a < A; ++a) {
    float e;
    if (eps != nullptr) {
      e = eps[(size_t)row * A + a];
    } else {
      if ((a & 3) == 0) {
        const Philox4 rr = philox4x32_10((uint32_t)(row + row0), (uint32_t)(a >> 2), draw, kStreamEps, (uint32_t)seed,
                                         (uint32_t)(seed >> 32));
        float z[4];
        box_muller4(rr, z);
        z0 = z[0]; z1 = z[1]; z2 = z[2]; z3 = z[3];
      }
      const int q = a & 3;
      e = q == 0 ? z0 : (q == 1 ? z1 : (q == 2 ? z2 : z3));
    }
    const float ls = log_std[a];
    const float sd = expf(ls);
    const float m = mu[(size_t)row * ldmu + a];
    const float act = m + e * sd;
    const float d = act - m;
    lp += -(d * d) / (2.0f * (sd * sd)) - logf(sd) - kLogSqrt2Pi;
    if (act_raw) act_raw[(size_t)row * A + a] = act;
    if (act_clip) act_clip[(size_t)row * A + a] = fminf(fmaxf(act, lo), hi);
  }
  if (logp_out) logp_out[row] = lp;
}

// ------------------------------------------------------------------------------------------------
// gSDE -- generalised state-dependent exploration (`use_sde=True`; SB3 StateDependentNoiseDistribution with its defaults
// full_std, no expln, no squashing, learn_features=False).  log_std is a [HL][A] matrix (HL = width of the policy's last hidden
// layer, the "latent_sde"); per environment an exploration matrix theta_n ~ N(0, exp(log_std)^2) is drawn at reset_noise and kept
// for sde_sample_freq steps:
//   action = mu + latent . theta_n                     [get_noise: bmm(latent_sde, exploration_matrices)]
//   variance = latent^2 . exp(log_std)^2 ; sigma = sqrt(variance + 1e-6)      [proba_distribution]
//   log_prob = sum_a Normal(mu, sigma).log_prob(action) ; entropy = sum_a 0.5 + log sqrt(2 pi) + log sigma
// The latent is DETACHED in the variance (learn_features=False): log_std is the only parameter the variance reaches.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kStreamSde = 0x53444531u;   // 'SDE1'
constexpr float kSdeEpsilon = 1e-6f;
// The two shape / parametrisation options of the distribution (policy_kwargs full_std, use_expln):
//   full (SB3's default): log_std is [HL][A]; else [HL][1], one standard deviation per latent unit shared by the actions
//   expln: std = exp(ls) for ls <= 0, log1p(ls + 1e-6) + 1 above (get_std: keeps the standard deviation from growing too fast)
struct SdeMode { int full, expln; };
__device__ __forceinline__ float sde_std(float ls, int expln) {
  return (expln && ls > 0.f) ? log1pf(ls + kSdeEpsilon) + 1.0f : expf(ls);
}
// d std / d log_std
__device__ __forceinline__ float sde_dstd(float ls, int expln) {
  return (expln && ls > 0.f) ? 1.0f / (1.0f + (ls + kSdeEpsilon)) : expf(ls);
}
// log_std entry of (latent unit k, action a)
__device__ __forceinline__ float sde_ls(const float* __restrict__ log_std, int k, int a, int A, int full) { return log_std[full ? k * A + a : k]; }
// theta rows [row0, row0 + nrows) of E [N][HL * A] (+ the single matrix E1 by block row `nrows`): one thread per four elements.
// The matrix of env n at draw index d is a function of (n, d), not of the launch that draws it.
__global__ void k_sde_resample(const float* __restrict__ log_std, int HLA, int A, SdeMode md, int row0, int nrows, uint64_t seed, uint32_t draw_rel,
                               const uint32_t* __restrict__ draw_base, float* __restrict__ E, float* __restrict__ E1) {
  const int per = (HLA + 3) / 4;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (nrows + (E1 != nullptr ? 1 : 0)) * per) return;
  const int r = i / per, c = i - r * per;
  const bool single = r >= nrows;
  const uint32_t draw = draw_rel + (draw_base ? *draw_base : 0u);
  float z[4];
  box_muller4(philox4x32_10(single ? 0xFFFFFFFFu : (uint32_t)(row0 + r), (uint32_t)c, draw, kStreamSde, (uint32_t)seed, (uint32_t)(seed >> 32)), z);
  float* out = single ? E1 : E + (size_t)(row0 + r) * HLA;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (4 * c + j < HLA) out[4 * c + j] = z[j] * sde_std(sde_ls(log_std, (4 * c + j) / A, (4 * c + j) % A, A, md.full), md.expln);
}
// the same from caller-supplied standard normals z [rows][HL * A] (tests / the oracle's noise as an INPUT, like `eps`)
__global__ void k_sde_from_z(const float* __restrict__ z, const float* __restrict__ log_std, int HLA, int A, SdeMode md, int n, float* __restrict__ E) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) E[i] = z[i] * sde_std(sde_ls(log_std, (i % HLA) / A, i % A, A, md.full), md.expln);
}
// std^2 per (latent unit, action) of the current log_std: [HL][A] (full_std or not): what the variance contracts the squared latent with
__global__ void k_sde_std2(const float* __restrict__ log_std, int HL, int A, SdeMode md, float* __restrict__ S2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= HL * A) return;
  const float sd = sde_std(sde_ls(log_std, i / A, i % A, A, md.full), md.expln);
  S2[i] = sd * sd;
}
// sum over a wave of FOUR values at once (the four actions a wave-per-row kernel carries per pass)
__device__ __forceinline__ void wave_sum4(float (&v)[4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = wave_sum(v[j]);
}
// variance[row][a] = sum_k latent[row][k]^2 S2[k][a] (+ epsilon is added by the consumers' sqrt): ONE WAVE per row, lanes over the
// latent units, four actions per pass (first version: one thread per row with exp() per (k, a) -- 12.6 M exps and 50 MB of strided
// reads per rollout step at 4096 envs, 256 units, 12 actions: 1.2 ms; profiles/r6/generic_chain.txt)
// lat2 (training): the squared latent itself, [n][HL], for the log_std-gradient GEMM -- written here, coalesced, as a by-product.
__global__ __launch_bounds__(256) void k_sde_var(const float* __restrict__ lat, int ldl, const float* __restrict__ S2, int n, int HL, int A,
                                                 float* __restrict__ var, int ldv, float* __restrict__ lat2) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const float* l = lat + (size_t)row * ldl;
  for (int a0 = 0; a0 < A; a0 += 4) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = lane; k < HL; k += 64) {
      const float l2 = l[k] * l[k];
      if (lat2 != nullptr && a0 == 0) lat2[(size_t)row * HL + k] = l2;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (a0 + j < A) acc[j] = fmaf(l2, S2[k * A + a0 + j], acc[j]);
    }
    wave_sum4(acc);
    if (lane == 0)
      for (int j = 0; j < 4 && a0 + j < A; ++j) var[(size_t)row * ldv + a0 + j] = acc[j];
  }
}
// rollout-time sampling epilogue under gSDE: ONE WAVE per env row.  noise = latent . theta_row with the row's matrix read once,
// contiguously ([k][a]: a lane owns latent unit k and its A consecutive entries); the variance comes from k_sde_var.
// E: the rows' matrices ([row][HL][A], erow0 = matrix row of row 0) or, with single != 0, ONE matrix for every row (SB3 get_noise when the
// batch is not the exploration batch).
__global__ __launch_bounds__(256) void k_sample_sde(const float* __restrict__ mu, int ldmu, const float* __restrict__ lat, int ldl,
                                                    const float* __restrict__ var, int ldv, const float* __restrict__ E, int erow0, int single,
                                                    int n, int HL, int A, float lo, float hi, float* __restrict__ act_raw,
                                                    float* __restrict__ act_clip, float* __restrict__ logp_out) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const float* l = lat + (size_t)row * ldl;
  const float* Er = single ? E : E + (size_t)(erow0 + row) * HL * A;
  float lp = 0.f;
  for (int a0 = 0; a0 < A; a0 += 4) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = lane; k < HL; k += 64) {
      const float lk = l[k];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (a0 + j < A) acc[j] = fmaf(lk, Er[k * A + a0 + j], acc[j]);
    }
    wave_sum4(acc);
    if (lane == 0)
      for (int j = 0; j < 4 && a0 + j < A; ++j) {
        const int a = a0 + j;
        const float sigma = sqrtf(var[(size_t)row * ldv + a] + kSdeEpsilon);
        const float m = mu[(size_t)row * ldmu + a];
        const float act = m + acc[j];
        const float d = act - m;
        lp += -(d * d) / (2.0f * (sigma * sigma)) - logf(sigma) - kLogSqrt2Pi;
        if (act_raw) act_raw[(size_t)row * A + a] = act;
        if (act_clip) act_clip[(size_t)row * A + a] = fminf(fmaxf(act, lo), hi);
      }
  }
  if (lane == 0 && logp_out) logp_out[row] = lp;
}

// deterministic predict: clip(mu)
__global__ void k_clip_mean(const float* __restrict__ mu, int ldmu, int n, int A, float lo, float hi,
                            float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * A) return;
  const int row = i / A, a = i - row * A;
  out[i] = fminf(fmaxf(mu[(size_t)row * ldmu + a], lo), hi);
}

// ------------------------------------------------------------------------------------------------
// rollout_buffer.add scalars: rewards (after time-limit bootstrap), episode_starts (= previous dones)
// bootstrap: r = f32( f64(r) + f64( f32(gamma) * V(terminal_obs) ) )   [oracle bootstrap_reward]
// ------------------------------------------------------------------------------------------------
__global__ void k_store_step(const float* __restrict__ rew_in, const float* __restrict__ prev_dones,
                             const uint8_t* __restrict__ trunc, const float* __restrict__ term_values,
                             float gamma, int n, float* __restrict__ rew_out, float* __restrict__ es_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float r = rew_in[i];
  if (trunc != nullptr && trunc[i]) {
    const float gv = __fmul_rn(gamma, term_values[i]);
    r = (float)((double)r + (double)gv);
  }
  rew_out[i] = r;
  es_out[i] = prev_dones[i];
}

// Zero-copy staging for pinned host buffers (hipHostMalloc memory is device-visible under the same address): the
// kernel itself pulls the rows over PCIe with coalesced loads and pads them to the device row stride -- no DMA
// descriptor per row as hipMemcpy2DAsync needs, no second stream, no events.
__global__ void k_pull_rows(const float* __restrict__ src, float* __restrict__ dst, int rows, int D, int Dp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * Dp) return;
  const int r = i / Dp, c = i - r * Dp;
  dst[i] = c < D ? src[(size_t)r * D + c] : 0.f;
}
__global__ void k_copy_f32(const float* __restrict__ src, float* __restrict__ dst, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

__global__ void k_u8_to_f32(const uint8_t* __restrict__ in, float* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i] ? 1.f : 0.f;
}

// ------------------------------------------------------------------------------------------------
// Value of flagged rows only (time-limit bootstrap is rare: one block per env, exits unless flagged).
// V(x) = Wv . tanh(W2 tanh(W1 x + b1) + b2) + bv with the canonical (unpadded) parameter vector.
// ------------------------------------------------------------------------------------------------
// x[D] is already in LDS; h1[G1], h2[G2], red[16] are LDS scratch.  Every thread of the block returns V(x).
// Two tanh layers: the form the fused rollout kernels call (their networks are 2 x 64 / 2 x 256 tanh by construction).
__device__ __forceinline__ float value_row_lds(const float* x, float* h1, float* h2, float* red,
                                               const float* __restrict__ W1, const float* __restrict__ b1,
                                               const float* __restrict__ W2, const float* __restrict__ b2,
                                               const float* __restrict__ Wv, const float* __restrict__ bv, int D, int G1,
                                               int G2) {
  for (int j = threadIdx.x; j < G1; j += blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < D; ++k) s = fmaf(x[k], W1[(size_t)j * D + k], s);
    h1[j] = tanhf(s + b1[j]);
  }
  __syncthreads();
  for (int j = threadIdx.x; j < G2; j += blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < G1; ++k) s = fmaf(h1[k], W2[(size_t)j * G1 + k], s);
    h2[j] = tanhf(s + b2[j]);
  }
  __syncthreads();
  float p = 0.f;
  for (int k = threadIdx.x; k < G2; k += blockDim.x) p += h2[k] * Wv[k];
  return block_sum(p, red) + bv[0];
}

// The value network as the per-row evaluators of the generic paths take it: canonical (unpadded) parameters, `L` hidden layers
// (1 .. kMaxHidden) of widths G[l], activation `act` (ACT_*).
struct ValueNetArgs {
  const float* W[kMaxHidden]; const float* b[kMaxHidden]; const float* Wv; const float* bv;
  int G[kMaxHidden];
  int L, act;
  int width_sum;   // G[0] + ... + G[L - 1]: floats of LDS scratch value_net_row needs for the activations
};
// x[D] in LDS; hbuf[width_sum], red[16] LDS scratch.  Every thread of the block returns V(x).  Same per-unit fma chains as the
// two-layer form above.  (The layer loop is unrolled with static indices: a runtime-indexed by-value argument struct would be
// copied to scratch memory.)
__device__ __forceinline__ float value_net_row(const float* x, float* hbuf, float* red, const ValueNetArgs& vn, int D) {
  const float* in = x;
  int K = D;
  float* out = hbuf;
#pragma unroll
  for (int l = 0; l < kMaxHidden; ++l) {
    if (l < vn.L) {
      const float* __restrict__ W = vn.W[l];
      const float* __restrict__ bb = vn.b[l];
      const int G = vn.G[l];
      for (int j = threadIdx.x; j < G; j += blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < K; ++k) s = fmaf(in[k], W[(size_t)j * K + k], s);
        out[j] = act_fwd(vn.act, s + bb[j]);
      }
      __syncthreads();
      in = out; K = G; out += G;
    }
  }
  float p = 0.f;
  for (int k = threadIdx.x; k < K; k += blockDim.x) p += in[k] * vn.Wv[k];
  return block_sum(p, red) + vn.bv[0];
}

__global__ __launch_bounds__(256) void k_value_flagged(const float* __restrict__ obs, int ldo,
                                                       const uint8_t* __restrict__ flags, ValueNetArgs vn, int D,
                                                       float* __restrict__ out, float* __restrict__ rew_inout, float gamma) {
  const int row = blockIdx.x;
  if (!flags[row]) return;
  extern __shared__ float sm[];  // x[D] | activations[width_sum] | red[16]
  float* x = sm;
  float* hbuf = x + D;
  float* red = hbuf + vn.width_sum;
  for (int i = threadIdx.x; i < D; i += blockDim.x) x[i] = obs[(size_t)row * ldo + i];
  __syncthreads();
  const float v = value_net_row(x, hbuf, red, vn, D);
  if (threadIdx.x == 0) {
    out[row] = v;
    if (rew_inout != nullptr)  // rewards[idx] += gamma * V(terminal_obs)  [oracle bootstrap_reward]
      rew_inout[row] = (float)((double)rew_inout[row] + (double)__fmul_rn(gamma, v));
  }
}

// ------------------------------------------------------------------------------------------------
// Pipelined host-env rollout: everything that follows one env step of a row range, in ONE launch (every launch is
// ~5 us of host time on this path).  Block b owns rows [16 b, 16 b + 16) of the range:
//   * rows flagged TimeLimit.truncated: V(terminal_obs) with the row read from the caller's pinned buffer in place,
//     then the bootstrap r = f32(f64(r) + f64(f32(gamma) * V))            [same arithmetic as k_value_flagged + k_store_step]
//   * rollout_buffer.add scalars: rewards, episode_starts (= previous dones), previous dones <- dones
//   * the NEXT observations of the rows are pulled into rollout slot t+1 (padded to Dp), so that the following
//     act_part launches only the policy kernel.
// ------------------------------------------------------------------------------------------------
constexpr int kPartRows = 16;
struct StorePullArgs {
  const float* rew_in; const uint8_t* dones; const uint8_t* trunc; const float* term_obs;  // caller's pinned rows (range base)
  ValueNetArgs vn;                                                                         // value network (canonical)
  int D, Dp, n;
  float gamma;
  float *prev_dones, *rew_out, *es_out, *term_val;
  const float* next_obs; float* obs_slot;  // null: no pull
};
__global__ __launch_bounds__(256) void k_store_pull_part(StorePullArgs a) {
  extern __shared__ float sm[];  // x[D] | activations[width_sum] | red[16] | tv[16]
  float* x = sm;
  float* hbuf = x + a.D;
  float* red = hbuf + a.vn.width_sum;
  float* tv = red + 16;
  const int i0 = blockIdx.x * kPartRows, tid = threadIdx.x;
  if (a.trunc != nullptr) {
    for (int r = 0; r < kPartRows; ++r) {
      const int i = i0 + r;
      if (i >= a.n || !a.trunc[i]) continue;  // block-uniform
      for (int k = tid; k < a.D; k += blockDim.x) x[k] = a.term_obs[(size_t)i * a.D + k];
      __syncthreads();
      const float v = value_net_row(x, hbuf, red, a.vn, a.D);
      if (tid == 0) tv[r] = v;
      __syncthreads();
    }
  }
  __syncthreads();
  if (tid < kPartRows && i0 + tid < a.n) {
    const int i = i0 + tid;
    float r = a.rew_in[i];
    if (a.trunc != nullptr && a.trunc[i]) {
      const float gv = __fmul_rn(a.gamma, tv[tid]);
      r = (float)((double)r + (double)gv);
      a.term_val[i] = tv[tid];
    }
    a.rew_out[i] = r;
    a.es_out[i] = a.prev_dones[i];
    a.prev_dones[i] = a.dones[i] ? 1.f : 0.f;
  }
  if (a.next_obs != nullptr) {
    for (int idx = tid; idx < kPartRows * a.Dp; idx += blockDim.x) {
      const int r = idx / a.Dp, c = idx - r * a.Dp, i = i0 + r;
      if (i < a.n) a.obs_slot[(size_t)i * a.Dp + c] = c < a.D ? a.next_obs[(size_t)i * a.D + c] : 0.f;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// GAE(lambda) reverse scan [SB3 RolloutBuffer.compute_returns_and_advantage; oracle gae()].
// The recurrence g[t] = delta[t] + coef[t] * g[t+1] is serial in t (and must stay in the oracle's order to be
// bit-exact: float64 carry, one rounding per operation), but everything around it is not.  One block owns
// kGaeEnvs = 16 envs (64-byte row segments of the [T][N] arrays -> cdiv(N,16) blocks: 256 at N = 4096) and walks
// time in tiles of kGaeChunk = 64 steps with its waves specialised:
//   wave 0 (16 lanes)   the serial steps of tile k out of LDS: two dependent f64 operations per step;
//   waves 1-4           meanwhile write advantages / returns of tile k-1, turn the rows of tile k+1 (already in
//                       registers) into delta and coef in the other LDS buffer -- no carry involved -- and issue
//                       the global loads of tile k+2.
// One barrier per tile.  The former one-lane-per-env kernel had 64 waves chip-wide and was bound by HBM latency
// (205 us at T = 1000, N = 4096); this one is bound by the 1000-step dependent f64 chain.
// ------------------------------------------------------------------------------------------------
#ifndef GAE_SKIP
#define GAE_SKIP 0  // timing-only ablation switch (scratch/time_gae.py)
#endif
constexpr int kGaeEnvs = 16;
constexpr int kGaeChunk = 64;
constexpr int kGaeThreads = 64 + 256;
__device__ __forceinline__ double pinned(double x) {  // keeps a product from being contracted into the next add
  asm volatile("" : "+v"(x));
  return x;
}
__global__ __launch_bounds__(kGaeThreads) void k_gae(const float* __restrict__ rewards,
                                                     const float* __restrict__ values,
                                                     const float* __restrict__ episode_starts,
                                                     const float* __restrict__ last_values,
                                                     const float* __restrict__ last_dones, float gamma, double gl_d,
                                                     int T, int N, float* __restrict__ adv, float* __restrict__ ret) {
  constexpr int E = kGaeEnvs, C = kGaeChunk, PER = C * E / 256;
  // row j of tile k <-> t = tlo(k) + j.  delta / coef / carry cross LDS as float64 so that the conversions (as slow
  // as the f64 arithmetic itself) are done by the loader waves, not on the serial chain.
  __shared__ double s_d[2][C * E], s_c[2][C * E], s_a[2][C * E];
  __shared__ float s_v[2][C * E];
  const int tid = threadIdx.x, n0 = blockIdx.x * E;
  const float gl_f = (float)gl_d;  // python-float gamma*lambda meets a float32 array -> rounded to f32
  const int K = (T - 1 + C - 1) / C;  // tiles over t = T-2 .. 0; tile k ends at t_hi = T-2 - k*C
  const bool scan_wave = tid < 64;
  const int lt = tid - 64;  // loader thread id

  float pr[PER], pv[PER], pn[PER], pe[PER];
  auto fetch = [&](int k) {  // rows of tile k -> registers (rows with t < 0 are padding)
    const int tlo = T - 2 - k * C - C + 1;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int idx = lt + i * 256, t = tlo + (idx / E), n = n0 + (idx % E);
      const bool ok = t >= 0 && n < N;
      const size_t o = ok ? (size_t)t * N + n : 0;
      pr[i] = ok ? rewards[o] : 0.f;
      pv[i] = ok ? values[o] : 0.f;
      pn[i] = ok ? values[o + N] : 0.f;
      pe[i] = ok ? episode_starts[o + N] : 0.f;
    }
  };
  auto stash = [&](int b) {  // registers -> delta / coef / values of the tile in LDS buffer b
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int idx = lt + i * 256;
      const float nnt = __fsub_rn(1.0f, pe[i]);
      s_d[b][idx] = (double)__fsub_rn(__fadd_rn(pr[i], __fmul_rn(__fmul_rn(gamma, pn[i]), nnt)), pv[i]);
      s_c[b][idx] = (double)__fmul_rn(gl_f, nnt);
      s_v[b][idx] = pv[i];
    }
  };
  auto write_out = [&](int k) {
    const int tlo = T - 2 - k * C - C + 1, b = k & 1;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int idx = lt + i * 256, t = tlo + (idx / E), n = n0 + (idx % E);
      if (t >= 0 && n < N) {
        const size_t o = (size_t)t * N + n;
        const float a = (float)s_a[b][idx];
        adv[o] = a;
        ret[o] = __fadd_rn(a, s_v[b][idx]);
      }
    }
  };

  double g = 0.0;
  if (scan_wave) {
    if (tid < E && n0 + tid < N) {  // t = T-1: float64 arithmetic because `1.0 - dones` (bool) is float64 in SB3
      const int n = n0 + tid;
      const size_t o = (size_t)(T - 1) * N + n;
      const double nnt = 1.0 - (double)last_dones[n];
      const float gv = __fmul_rn(gamma, last_values[n]);
      const double tmp = pinned(__dmul_rn((double)gv, nnt));
      const float v = values[o];
      g = __dsub_rn(__dadd_rn((double)rewards[o], tmp), (double)v);  // + (gl * nnt) * 0
      const float a = (float)g;
      adv[o] = a;
      ret[o] = __fadd_rn(a, v);
    }
  } else if (K > 0) {
    fetch(0);
    stash(0);
    if (K > 1) fetch(1);
  }
  __syncthreads();
  for (int k = 0; k < K; ++k) {
    if (scan_wave) {
      if (tid < E && !(GAE_SKIP & 1)) {  // padding rows (t < 0, last tile only) are scanned too: their results are never written out
        const double* sd = s_d[k & 1] + tid;
        const double* sc = s_c[k & 1] + tid;
        double* sa = s_a[k & 1] + tid;
        double d[8], c[8], dn[8], cn[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          d[u] = sd[(C - 1 - u) * E];
          c[u] = sc[(C - 1 - u) * E];
        }
#pragma unroll
        for (int j = C - 1; j >= 0; j -= 8) {
          if (j - 8 >= 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              dn[u] = sd[(j - 8 - u) * E];
              cn[u] = sc[(j - 8 - u) * E];
            }
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {  // the chain first, its eight LDS stores after it
            g = __dadd_rn(d[u], pinned(__dmul_rn(c[u], g)));
            d[u] = g;
          }
          asm volatile("" ::: "memory");
#pragma unroll
          for (int u = 0; u < 8; ++u) sa[(j - u) * E] = d[u];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            d[u] = dn[u];
            c[u] = cn[u];
          }
        }
      }
    } else {
      if (k >= 1 && !(GAE_SKIP & 2)) write_out(k - 1);
      if (k + 1 < K && !(GAE_SKIP & 8)) stash((k + 1) & 1);
      if (k + 2 < K && !(GAE_SKIP & 4)) fetch(k + 2);
    }
    __syncthreads();
  }
  if (!scan_wave && K > 0) write_out(K - 1);
}

// ------------------------------------------------------------------------------------------------
// Minibatch index construction.  SB3 flat index is env-major (flat = n*T + t); device rows are t*N + n.
// ------------------------------------------------------------------------------------------------
// `perm` holds int64 flat indices < 2^30: only the LOW words are read, and the HIGH word of entry `row` receives the
// minibatch the row belongs to (k_adv_stats_stream reads it back in storage order) -- no second [T*N] array.
__global__ void k_perm_from_host(int64_t* __restrict__ perm, int total, int T, int N, int bl,
                                 int* __restrict__ rows) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int f = reinterpret_cast<const int*>(perm)[2 * (size_t)i];
  const int n = f / T, t = f - n * T;
  rows[i] = t * N + n;
  reinterpret_cast<int*>(perm)[2 * (size_t)(t * N + n) + 1] = i / bl;
}
__global__ void k_perm_feistel(int total, int T, int N, int half_bits, uint32_t k0, uint32_t k1,
                               int* __restrict__ rows, int64_t* __restrict__ flat_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const uint64_t f = feistel_perm((uint64_t)i, (uint64_t)total, half_bits, k0, k1);
  if (flat_out) flat_out[i] = (int64_t)f;
  if (rows) {
    const int n = (int)(f / (uint64_t)T), t = (int)(f - (uint64_t)n * T);
    rows[i] = t * N + n;
  }
}

// ------------------------------------------------------------------------------------------------
// Per-minibatch (sum, sum of squares, count) of the advantages (normalize_advantage).
//
// Round 1-2 gathered adv[rows[i]] minibatch by minibatch: 4.1 M four-byte gathers pull 419 MB of lines for 16 MB of
// data.  Now the advantages are STREAMED once in storage order (coalesced 16-byte loads, 16 MB) and every element finds
// its minibatch itself: position = inverse permutation of its env-major index (the Feistel permutation is invertible in
// registers: feistel_perm_inv; a host-supplied permutation leaves the minibatch id beside each row, k_perm_from_host),
// minibatch = position / batch.  Elements of one minibatch therefore arrive in arbitrary order on arbitrary workgroups,
// so the sums are taken in INTEGERS -- fixed point, scaled by a power of two chosen from max |adv| so that neither sum
// can overflow 63 bits -- whose addition is associative: the result does not depend on the order, the launch geometry
// or the atomics' timing, and is reproducible bit for bit.  Resolution: 2^-45 of max |adv| per element (float32
// carries 2^-24); the sums equal the exact ones to ~1e-12 relative.
//   k_adv_absmax   max |adv| (bits of a non-negative float order like unsigned integers) + zeroes the bins
//   k_adv_stats_stream   the pass: LDS bins per workgroup (ds_add_u64), one global 64-bit atomic per bin and workgroup
//   k_adv_fold     integers -> (sum, sumsq, count) in float64
// ------------------------------------------------------------------------------------------------
constexpr int kAdvLdsMinibatches = 2048;  // LDS bins (32 KB); beyond that the adds go straight to the global bins
// LDS atomics on ONE address serialise, and 64 lanes that fall into 32-63 minibatches collide on almost every instruction:
// the bins are replicated, lane l adds to replica l & (R - 1) (integer sums: any split adds up to the same bits)
__host__ __device__ inline int adv_bin_replicas(int nmb) { return nmb <= 256 ? 8 : 1; }
struct AdvStatArgs {
  const float* adv; int total, T, N, bl, nmb;
  int half_bits; uint32_t k0, k1;       // Feistel permutation of this epoch ...
  const int* mb_of_row;                 // ... or (host permutation) the minibatch of storage row r at [2r + 1]
  unsigned* absmax_bits;                // max |adv| of THIS epoch's pass (zero on entry) ...
  unsigned* absmax_next;                // ... and the word the next epoch will use, zeroed by this one
  unsigned long long* bins;             // [nmb][2] two's-complement sums
};
// (one atomic per WORKGROUP on the max word and per non-empty bin: atomics on one address retire at ~12 ns each, so the
//  grids are at most one workgroup per CU -- 4 096 wave-level atomicMax calls were 49 us of a 50 us kernel)
__global__ __launch_bounds__(1024) void k_adv_absmax(AdvStatArgs a) {
  __shared__ float wmax[16];
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
  for (int i = tid; i < 2 * a.nmb; i += nth) a.bins[i] = 0ull;
  if (tid == 0) *a.absmax_next = 0u;
  float m = 0.f;
  const int n4 = a.total >> 2;
  for (int i = tid; i < n4; i += nth) {
    const f32x4 v = reinterpret_cast<const f32x4*>(a.adv)[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
  }
  for (int i = 4 * n4 + tid; i < a.total; i += nth) m = fmaxf(m, fabsf(a.adv[i]));
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) m = fmaxf(m, wmax[w]);
    if (m > 0.f) atomicMax(a.absmax_bits, __float_as_uint(m));
  }
}
// power-of-two scales: |adv| < 2^e, so |adv * 2^(s1)| < 2^(61 - log2ceil(bl)) and a minibatch of bl elements sums below 2^61
__device__ __forceinline__ void adv_scales(unsigned absmax_bits, int bl, int& s1, int& s2) {
  const int e = (int)((absmax_bits >> 23) & 0xff) - 127 + 1;  // |adv| < 2^e (0 for an all-zero rollout: any scale works)
  int lb = 0;
  while ((1ll << lb) < (long long)bl) ++lb;
  s1 = 61 - lb - e;
  s2 = 61 - lb - 2 * e;
}
__global__ __launch_bounds__(1024) void k_adv_stats_stream(AdvStatArgs a) {
  extern __shared__ unsigned long long adv_bins_lds[];
  const bool use_lds = a.nmb <= kAdvLdsMinibatches;
  const int R = adv_bin_replicas(a.nmb);
  if (use_lds)
    for (int i = threadIdx.x; i < 2 * a.nmb * R; i += blockDim.x) adv_bins_lds[i] = 0ull;
  int s1, s2;
  adv_scales(*a.absmax_bits, a.bl, s1, s2);
  const double f1 = ldexp(1.0, s1), f2 = ldexp(1.0, s2);
  __syncthreads();
  unsigned long long* bins = use_lds ? adv_bins_lds + (size_t)(threadIdx.x & (R - 1)) * 2 * a.nmb : a.bins;
  // (the pass is VALU bound, not memory bound: everything per element is 32-bit -- total <= 2^30 -- and the storage row
  //  is split into (step, env) once per quad of consecutive rows)
  auto add = [&](int row, int t, int n, float x) {
    int mb;
    if (a.mb_of_row) {
      mb = a.mb_of_row[2 * (size_t)row + 1];
    } else {
      const uint32_t pos = feistel_perm_inv32((uint32_t)n * (uint32_t)a.T + (uint32_t)t, (uint32_t)a.total, a.half_bits, a.k0, a.k1);
      mb = (int)(pos / (uint32_t)a.bl);
    }
    const double xd = (double)x;
    const long long q1 = __double2ll_rn(xd * f1), q2 = __double2ll_rn(xd * xd * f2);
    atomicAdd(&bins[2 * mb], (unsigned long long)q1);
    atomicAdd(&bins[2 * mb + 1], (unsigned long long)q2);
  };
  // contiguous quads of storage rows per workgroup (coalesced 16-byte loads), grid-strided
  const int n4 = a.total >> 2;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
    const f32x4 v = reinterpret_cast<const f32x4*>(a.adv)[i];
    int t = (int)((uint32_t)(4 * i) / (uint32_t)a.N), n = 4 * i - t * a.N;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      add(4 * i + j, t, n, v[j]);
      if (++n == a.N) { n = 0; ++t; }
    }
  }
  if (blockIdx.x == 0)
    for (int i = 4 * n4 + threadIdx.x; i < a.total; i += blockDim.x) add(i, i / a.N, i % a.N, a.adv[i]);
  if (use_lds) {
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * a.nmb; i += blockDim.x) {
      unsigned long long v = 0ull;
      for (int q = 0; q < R; ++q) v += adv_bins_lds[(size_t)q * 2 * a.nmb + i];
      if (v != 0ull) atomicAdd(&a.bins[i], v);
    }
  }
}
__global__ void k_adv_fold(AdvStatArgs a, double* __restrict__ out) {
  const int mb = blockIdx.x * blockDim.x + threadIdx.x;
  int s1, s2;
  adv_scales(*a.absmax_bits, a.bl, s1, s2);
  if (mb < a.nmb) {
    out[4 * mb + 0] = ldexp((double)(long long)a.bins[2 * mb], -s1);
    out[4 * mb + 1] = ldexp((double)(long long)a.bins[2 * mb + 1], -s2);
    out[4 * mb + 2] = (double)(min(mb * a.bl + a.bl, a.total) - mb * a.bl);
    out[4 * mb + 3] = 0.0;
  }
}
// gather the minibatch rows into contiguous work arrays (generic path).  First kernel of an optimizer step: it also zeroes what the
// step accumulates into with atomics (zero0 / zero1: the gradient vector with its loss sums, the gSDE log_std GEMM's output) -- one
// launch instead of a memset each.
__global__ void k_gather(const int* __restrict__ rows, int count, const float* __restrict__ obs, int Dp,
                         const float* __restrict__ actions, int A, const float* __restrict__ logp,
                         const float* __restrict__ adv, const float* __restrict__ ret, float* __restrict__ Xg,
                         float* __restrict__ actg, float* __restrict__ lpg, float* __restrict__ advg,
                         float* __restrict__ retg, const float* __restrict__ values, float* __restrict__ oldvg,
                         float* __restrict__ zero0, int nzero0, float* __restrict__ zero1, int nzero1) {
  const int per = Dp / 4;  // float4 chunks per row
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  for (int z = i; z < nzero0; z += gridDim.x * blockDim.x) zero0[z] = 0.f;
  for (int z = i; z < nzero1; z += gridDim.x * blockDim.x) zero1[z] = 0.f;
  if (i >= count * per) return;
  const int b = i / per, c = i - b * per;
  const int row = rows[b];
  reinterpret_cast<f32x4*>(Xg)[(size_t)b * per + c] = reinterpret_cast<const f32x4*>(obs)[(size_t)row * per + c];
  if (c == 0) {
    lpg[b] = logp[row];
    advg[b] = adv[row];
    retg[b] = ret[row];
    if (oldvg != nullptr) oldvg[b] = values[row];
  }
  for (int a = c; a < A; a += per) actg[(size_t)b * A + a] = actions[(size_t)row * A + a];
}

// ------------------------------------------------------------------------------------------------
// PPO loss forward + the gradient w.r.t. the network outputs [SB3 PPO.train; oracle loss_and_grads].
// One thread per minibatch row.  inv_bg = 1 / GLOBAL batch size.  advstat = global (sum,sumsq,count).
// Accumulates (atomics): sums[0..5] = sum min-surrogate, sum (ret-v)^2, sum kl-term, clip count,
// rows; g_log_std[A]; g_bias_action[A]; g_bias_value.
// ------------------------------------------------------------------------------------------------
struct LossArgs {
  const float* mu; int ldmu;
  const float* v;             // [B]
  const float* actions;       // [B][A]
  const float* old_logp; const float* adv; const float* ret;
  const float* log_std;
  const double* advstat;      // 4 doubles of this minibatch
  int B, A;
  int normalize;
  float clip, vf_coef, ent_coef, inv_bg;
  float clip_vf; const float* old_v;  // clip_range_vf (< 0: none), gathered old value predictions [B]
  float* dmu; int lddmu;      // [B][Ap]
  float* dv; int lddv;        // [B][8]
  float* sums;                // [8]
  float* g_log_std; float* g_b_action; float* g_b_value;
  // gSDE (lat != nullptr): the policy's last hidden activations of the minibatch [B][HL] (ld = HL), log_std is [HL][A];
  // gsig [B][ldg] <- dLoss / d sigma^2 per (row, action): one operand of the log_std gradient GEMM (the other, latent^2: k_sde_var)
  const float* lat; int HL;
  const float* var; int ldvar;   // [B][ldvar] <- k_sde_var: sigma^2 - epsilon per (row, action)
  float* gsig; int ldg;
};

__device__ __forceinline__ void adv_mean_std(const double* st, float* mean, float* sd, bool* on) {
  const double n = st[2];
  *on = n > 1.0;
  const double m = st[0] / (n > 0 ? n : 1.0);
  double var = (n > 1.0) ? (st[1] - n * m * m) / (n - 1.0) : 0.0;
  if (var < 0.0) var = 0.0;
  *mean = (float)m;
  *sd = (float)sqrt(var);
}

// Dynamic LDS: [4 waves][2 A + 8] floats.  Every per-action and per-minibatch sum is reduced inside its wave by shuffles and
// crosses the waves through LDS behind ONE barrier (one block_sum -- two barriers -- per quantity made this kernel 23 us for a
// 100-row minibatch with twelve actions); the zero padding of dmu / dv (K padding of the backward GEMMs) and the entropy term of
// the state-independent log_std are written here too (no memsets, no k_entropy_grad launch).
inline size_t loss_lds_bytes(int A) { return (size_t)4 * (2 * A + 8) * sizeof(float); }
__global__ __launch_bounds__(256) void k_loss(LossArgs L) {
  extern __shared__ float part[];
  const int nq = 2 * L.A + 8;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float* mine = part + wv * nq;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i < L.B;
  float s_pl = 0.f, s_vl = 0.f, s_kl = 0.f, s_cf = 0.f, g_logp = 0.f, dvv = 0.f, s_ent = 0.f;
  if (live) {
    float a = L.adv[i];
    float mean, sd; bool on;
    adv_mean_std(L.advstat, &mean, &sd, &on);
    if (L.normalize && on) a = (a - mean) / (sd + 1e-8f);
    float lp = 0.f;
    for (int k = 0; k < L.A; ++k) {
      const float sdv = L.lat != nullptr ? sqrtf(L.var[(size_t)i * L.ldvar + k] + kSdeEpsilon) : expf(L.log_std[k]);
      const float d = L.actions[(size_t)i * L.A + k] - L.mu[(size_t)i * L.ldmu + k];
      lp += -(d * d) / (2.0f * (sdv * sdv)) - logf(sdv) - kLogSqrt2Pi;
      s_ent += (0.5f + kLogSqrt2Pi) + logf(sdv);
    }
    const float log_ratio = lp - L.old_logp[i];
    const float ratio = expf(log_ratio);
    const float lo = 1.0f - L.clip, hi = 1.0f + L.clip;
    const float s1 = a * ratio, s2 = a * fminf(fmaxf(ratio, lo), hi);
    s_pl = fminf(s1, s2);
    s_cf = (fabsf(ratio - 1.0f) > L.clip) ? 1.f : 0.f;
    s_kl = (ratio - 1.0f) - log_ratio;
    const float in_range = (ratio >= lo && ratio <= hi) ? 1.f : 0.f;
    const float w1 = (s1 < s2) ? 1.f : ((s1 > s2) ? 0.f : 0.5f);
    const float d_ratio = -(w1 * a + (1.0f - w1) * a * in_range) * L.inv_bg;
    g_logp = d_ratio * ratio;
    float gv_;
    value_loss_terms(L.v[i], L.ret[i], L.clip_vf >= 0.f ? L.old_v[i] : 0.f, L.clip_vf, s_vl, gv_);
    dvv = L.vf_coef * gv_ * L.inv_bg;
    L.dv[(size_t)i * L.lddv] = dvv;
    for (int c = 1; c < L.lddv; ++c) L.dv[(size_t)i * L.lddv + c] = 0.f;                 // K padding of the value head's backward GEMM
    for (int c = L.A; c < L.lddmu; ++c) L.dmu[(size_t)i * L.lddmu + c] = 0.f;            // ... and of the action head's
    if (L.lat != nullptr)
      for (int c = L.A; c < L.ldg; ++c) L.gsig[(size_t)i * L.ldg + c] = 0.f;
  }
  // per-action pieces, reduced inside the wave
  for (int k = 0; k < L.A; ++k) {
    float gm = 0.f, gls = 0.f;
    if (live) {
      const float sdv = L.lat != nullptr ? sqrtf(L.var[(size_t)i * L.ldvar + k] + kSdeEpsilon) : expf(L.log_std[k]);
      const float var = sdv * sdv;
      const float d = L.actions[(size_t)i * L.A + k] - L.mu[(size_t)i * L.ldmu + k];
      gm = g_logp * d / var;
      gls = g_logp * (d * d / var - 1.0f);
      L.dmu[(size_t)i * L.lddmu + k] = gm;
      if (L.lat != nullptr)   // d logp / d sigma^2 = (d^2 / sigma^2 - 1) / (2 sigma^2); d(-mean entropy) / d sigma^2 = -1 / (2 sigma^2 B)
        L.gsig[(size_t)i * L.ldg + k] = (gls - L.ent_coef * L.inv_bg) / (2.0f * var);
    }
    const float t1 = wave_sum(gm), t2 = wave_sum(gls);
    if (lane == 0) { mine[2 * k] = t1; mine[2 * k + 1] = t2; }
  }
  {
    const float r0 = wave_sum(s_pl), r1 = wave_sum(s_vl), r2 = wave_sum(s_kl), r3 = wave_sum(s_cf), r4 = wave_sum(dvv), r5 = wave_sum(s_ent);
    if (lane == 0) {
      float* q = mine + 2 * L.A;
      q[0] = r0; q[1] = r1; q[2] = r2; q[3] = r3; q[4] = r4; q[5] = r5;
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < 2 * L.A + 6; t += blockDim.x) {
    const float tot = ((part[t] + part[nq + t]) + part[2 * nq + t]) + part[3 * nq + t];   // waves in order
    if (t < 2 * L.A) {
      const int k = t >> 1;
      if ((t & 1) == 0) atomicAdd(&L.g_b_action[k], tot);
      else if (L.lat == nullptr) {   // (gSDE: log_std's gradient is a GEMM over gsig, see k_sde_scale_grad)
        // + the entropy term of the loss, once: d(-mean(entropy)) / d log_std_a = -(B_local / B_global) * ent_coef
        atomicAdd(&L.g_log_std[k], blockIdx.x == 0 ? tot + L.ent_coef * (-(float)L.B) * L.inv_bg : tot);
      }
    } else {
      const int j = t - 2 * L.A;
      if (j < 4) atomicAdd(&L.sums[j], tot);
      else if (j == 4) atomicAdd(L.g_b_value, tot);
      else if (L.lat != nullptr) atomicAdd(&L.sums[5], tot);   // state-dependent entropy: summed per row (stats_row_from_sums, sde)
    }
  }
  if (threadIdx.x == 0) atomicAdd(&L.sums[4], (float)min(L.B - (int)(blockIdx.x * blockDim.x), (int)blockDim.x));
}

// gSDE: g_log_std[k][a] = (latent^2)^T . gsig, as the GEMM left it, times d sigma^2 / d log_std[k][a] / latent_k^2 = 2 exp(log_std[k][a])^2
// raw [HL][A]: the GEMM's output (== g_log_std itself with full_std); one latent unit's [HL][1] entry sums its row of raw in action order.
// In general d sigma^2 / d log_std = latent^2 * 2 std (d std / d log_std): 2 exp(2 ls) without expln.
__global__ void k_sde_scale_grad(float* __restrict__ g_log_std, const float* __restrict__ raw, const float* __restrict__ log_std, int HL, int A,
                                 SdeMode md) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (md.full ? HL * A : HL)) return;
  const float ls = log_std[i];
  const float f = 2.0f * sde_std(ls, md.expln) * sde_dstd(ls, md.expln);
  if (md.full) {
    g_log_std[i] = raw[i] * f;
  } else {
    float t = 0.f;
    for (int a = 0; a < A; ++a) t += raw[i * A + a];
    g_log_std[i] = t * f;
  }
}

// ------------------------------------------------------------------------------------------------
// Device-resident synthetic env source.  One thread per (env, 4 obs features).
// state: ep_len[N].  Writes next obs into rollout slot t+1, reward/done/trunc flags, terminal obs rows.
// ------------------------------------------------------------------------------------------------
// Per-step kernel of the device-resident rollout (the persistent kernels of kernels_rollout.h inline the same
// rules): env draw + rollout_buffer.add scalars.
// ep_len is double buffered (every chunk thread of an env must see the OLD value); the time-limit
// bootstrap of the (rare) truncated rows is applied in the same launch (see below).
__global__ void k_add_counters(uint32_t* ctr, uint32_t d0, uint32_t d1) {
  if (threadIdx.x == 0 && blockIdx.x == 0) { ctr[0] += d0; ctr[1] += d1; }
}
struct BootArgs {  // value network (canonical parameters, two tanh layers) for the in-kernel time-limit bootstrap of the fused rollout kernels
  const float *W1, *b1, *W2, *b2, *Wv, *bv;
  int G1, G2;
  float gamma;
  float* term_val;  // [N] V(terminal_obs) of truncated rows (diagnostics / tests)
};
struct BootNetArgs {  // the same for the per-step kernels of the generic path: any depth, any activation
  ValueNetArgs vn;
  float gamma;
  float* term_val;
};
constexpr int kBootMaxEnvs = 72;  // envs whose chunk-0 thread can live in one 256-thread block (Dp >= 16: <= 65)
inline size_t env_step_lds_bytes(int Dp, int width_sum) { return (size_t)(Dp + width_sum + 16 + 4 + 2 * kBootMaxEnvs) * 4; }

// Time-limit truncation is rare (one row in `time_limit`), so the bootstrap  r += gamma * V(terminal_obs)  [oracle
// bootstrap_reward] runs in the same launch: the block that owns chunk 0 of a truncated env re-draws its terminal
// observation into LDS and evaluates the value MLP with all 256 threads (same arithmetic as k_value_flagged).
__global__ __launch_bounds__(256) void k_env_step_store(uint64_t seed, uint32_t step_rel, const uint32_t* __restrict__ step_base, int N, int D,
                                 int Dp, float p_term, int time_limit,
                                 const int* __restrict__ ep_len_in, int* __restrict__ ep_len_out,
                                 float* __restrict__ obs_next, float* __restrict__ term_obs,
                                 const float* __restrict__ prev_dones, float* __restrict__ next_dones,
                                 uint8_t* __restrict__ trunc, float* __restrict__ rew_out, float* __restrict__ es_out,
                                 BootNetArgs bt) {
  extern __shared__ float sm[];  // x[Dp] | activations[width_sum] | red[16] | cnt[4] | env[kBootMaxEnvs] | rew[kBootMaxEnvs]
  float* x = sm;
  float* hbuf = x + Dp;
  float* red = hbuf + bt.vn.width_sum;
  int* cnt = reinterpret_cast<int*>(red + 16);
  int* lenv = cnt + 4;
  float* lrew = reinterpret_cast<float*>(lenv + kBootMaxEnvs);
  if (threadIdx.x == 0) *cnt = 0;
  __syncthreads();
  const int per = Dp / 4;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  const uint32_t step = step_rel + (step_base ? *step_base : 0u);  // device-resident base: graph replays advance it
  if (i < N * per) {
    const int n = i / per, c = i - n * per;
    const Philox4 mr = philox4x32_10((uint32_t)n, 0u, step, kStreamEnvMisc, k0, k1);
    const bool term = u32_to_unit_open(mr.x) < p_term;
    const int len = ep_len_in[n] + 1;
    const bool tr = (len >= time_limit) && !term;
    const bool done = term || tr;
    float z[4];
    box_muller4(philox4x32_10((uint32_t)n, (uint32_t)c, step, kStreamEnvObs, k0, k1), z);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (4 * c + j < D) ? z[j] : 0.f;
    if (tr) {
      reinterpret_cast<f32x4*>(term_obs)[(size_t)n * per + c] = o;
      box_muller4(philox4x32_10((uint32_t)n, (uint32_t)c, step, kStreamEnvTerm, k0, k1), z);
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (4 * c + j < D) ? z[j] : 0.f;
    }
    reinterpret_cast<f32x4*>(obs_next)[(size_t)n * per + c] = o;
    if (c == 0) {
      float zz[4];
      box_muller4(Philox4{mr.y, mr.z, mr.w, mr.x ^ 0x9E3779B9u}, zz);
      const float rew = 0.03f + 0.1f * zz[0] + (term ? 5.0f : 0.f);
      if (tr) {  // reward is written after the bootstrap below
        const int q = atomicAdd(cnt, 1);
        lenv[q] = n;
        lrew[q] = rew;
      } else {
        rew_out[n] = rew;
      }
      es_out[n] = prev_dones[n];
      next_dones[n] = done ? 1.f : 0.f;
      trunc[n] = tr ? 1 : 0;
      ep_len_out[n] = done ? 0 : len;
    }
  }
  __syncthreads();
  const int m = *cnt;
  for (int q = 0; q < m; ++q) {  // block-uniform; order within the list does not matter (rows are independent)
    const int n = lenv[q];
    if ((int)threadIdx.x < per) {  // the terminal observation = this step's kStreamEnvObs draw of env n
      const int c = threadIdx.x;
      float z[4];
      box_muller4(philox4x32_10((uint32_t)n, (uint32_t)c, step, kStreamEnvObs, k0, k1), z);
#pragma unroll
      for (int j = 0; j < 4; ++j) x[4 * c + j] = (4 * c + j < D) ? z[j] : 0.f;
    }
    __syncthreads();
    const float v = value_net_row(x, hbuf, red, bt.vn, D);
    if (threadIdx.x == 0) {
      bt.term_val[n] = v;
      rew_out[n] = (float)((double)lrew[q] + (double)__fmul_rn(bt.gamma, v));
    }
    __syncthreads();
  }
}
__global__ void k_env_reset(uint64_t seed, int N, int D, int Dp, float* __restrict__ obs0, int* __restrict__ ep_len) {
  const int per = Dp / 4;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * per) return;
  const int n = i / per, c = i - n * per;
  float z[4];
  box_muller4(philox4x32_10((uint32_t)n, (uint32_t)c, 0xFFFFFFFFu, kStreamEnvObs, (uint32_t)seed,
                            (uint32_t)(seed >> 32)), z);
  f32x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = (4 * c + j < D) ? z[j] : 0.f;
  reinterpret_cast<f32x4*>(obs0)[(size_t)n * per + c] = o;
  if (c == 0) ep_len[n] = 0;
}

// ------------------------------------------------------------------------------------------------
// train/explained_variance of SB3's PPO.train (explained_variance(values, returns) = 1 - Var[returns - values] / Var[returns]
// over the whole rollout buffer): per-block float64 partial sums (n is implied) of y, y^2, d = y - v, d^2 in a fixed order;
// the host adds the blocks in block order.  A logging quantity, called once per logged iteration.
// ------------------------------------------------------------------------------------------------
constexpr int kEvBlocks = 256;
__global__ __launch_bounds__(256) void k_explained_variance_partials(const float* __restrict__ values, const float* __restrict__ returns,
                                                                     int n, double* __restrict__ out) {
  __shared__ double red[4][256];
  double sy = 0.0, syy = 0.0, sd = 0.0, sdd = 0.0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const double y = returns[i], d = (double)returns[i] - (double)values[i];
    sy += y; syy += y * y; sd += d; sdd += d * d;
  }
  red[0][threadIdx.x] = sy; red[1][threadIdx.x] = syy; red[2][threadIdx.x] = sd; red[3][threadIdx.x] = sdd;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s)
      for (int k = 0; k < 4; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x < 4) out[blockIdx.x * 4 + threadIdx.x] = red[threadIdx.x][0];
}

}  // namespace mobrob
