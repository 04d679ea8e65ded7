// Probe (not product code): one GEMM phase of k_fused_train -- C[64][256] = A[64][256] . W^T, A in LDS as float32, W streamed from
// L2 in MFMA fragment order -- on the f32 matrix pipe and on the bf16 pipe with every float32 operand split into three bf16
// pieces (x = x1 + x2 + x3, each piece the bf16 rounding of what is left) and six of the nine piece products kept
// (x1y1, x1y2, x2y1, x1y3, x3y1, x2y2; the dropped ones are below 2^-24 of the product).  DESIGN.md 7.0e: "measure, do not guess".
//
//   variant 0   v_mfma_f32_32x32x2_f32, 128 per accumulator and phase (what the product kernels issue)
//   variant 1   v_mfma_f32_32x32x16_bf16 x 6 per 16 k; A split in the k loop (v_cvt_pk_bf16_f32 + exact residuals), W pre-split
//   variant 2   the same with A pre-split in LDS (three bf16 planes): the bound if the split is paid once per activation
// Workgroup = 4 waves, wave w owns output columns 64w..64w+63 (two 32-column blocks) x 64 rows (two row blocks): four 32x32
// accumulators, exactly the shape of the layer-2 forward phase.  Every workgroup repeats the phase `reps` times on its own tile.
//   hipcc --offload-arch=gfx950 -O3 -o scratch/bf16x3_probe scratch/bf16x3_probe.hip && scratch/bf16x3_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int R = 64, K = 256, N = 256, LDA = K + 4;

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
// two floats -> their three bf16 pieces, packed (lo element in the low half)
__device__ __forceinline__ void split2(float a, float b, unsigned& p1, unsigned& p2, unsigned& p3) {
  p1 = cvt_pk_bf16(a, b);
  const float ra = a - __uint_as_float(p1 << 16), rb = b - __uint_as_float(p1 & 0xffff0000u);   // exact
  p2 = cvt_pk_bf16(ra, rb);
  const float sa = ra - __uint_as_float(p2 << 16), sb = rb - __uint_as_float(p2 & 0xffff0000u);  // exact
  p3 = cvt_pk_bf16(sa, sb);
}
__device__ __forceinline__ bf16x8 as_bf16x8(const u32x4& v) { return __builtin_bit_cast(bf16x8, v); }

#define MFMA32F(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)
#define MFMA32B(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

// W packs.  f32: [cb 8][kg 32][lane 64] f32x4: W[cb*32 + r][kg*8 + 4h + s].  bf16: [cb 8][ks 16][piece 3][lane 64] 8 x bf16:
// piece of W[cb*32 + r][ks*16 + 8h + j].
template <int VARIANT>
__global__ __launch_bounds__(256, 1) void k_probe(const float* __restrict__ a_in, const f32x4* __restrict__ wf,
                                                  const u32x4* __restrict__ wb, int reps, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  unsigned* planes = reinterpret_cast<unsigned*>(lds);  // variant 2 (instead of the float32 tile): [piece 3][row 64][K/2 + 4] packed bf16 pairs
  constexpr int LDP = K / 2 + 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  if (VARIANT != 2) {
    for (int i = tid; i < R * K; i += 256) lds[(i / K) * LDA + (i % K)] = a_in[(size_t)blockIdx.x * R * K + i];
    __syncthreads();
  } else {
    for (int i = tid; i < R * K / 2; i += 256) {
      const int row = i / (K / 2), c2 = i % (K / 2);
      unsigned p1, p2, p3;
      const float* src = a_in + (size_t)blockIdx.x * R * K + (size_t)row * K + 2 * c2;
      split2(src[0], src[1], p1, p2, p3);
      planes[(0 * R + row) * LDP + c2] = p1;
      planes[(1 * R + row) * LDP + c2] = p2;
      planes[(2 * R + row) * LDP + c2] = p3;
    }
    __syncthreads();
  }
  f32x16 c00, c01, c10, c11;  // [col block][row block]
#pragma unroll
  for (int i = 0; i < 16; ++i) c00[i] = c01[i] = c10[i] = c11[i] = 0.f;
  for (int rep = 0; rep < reps; ++rep) {
    asm volatile("" ::: "memory");  // nothing of a phase (LDS reads, splits) may be hoisted out of the repetition loop
    if (VARIANT == 0) {
      const f32x4* w0 = wf + (size_t)(2 * wave) * 32 * 64 + lane;
      const f32x4* w1 = wf + (size_t)(2 * wave + 1) * 32 * 64 + lane;
#pragma unroll 4
      for (int kg = 0; kg < 32; ++kg) {
        const f32x4 p = w0[kg * 64], q = w1[kg * 64];
        const f32x4 u = *reinterpret_cast<const f32x4*>(&lds[r * LDA + 8 * kg + 4 * h]);
        const f32x4 v = *reinterpret_cast<const f32x4*>(&lds[(32 + r) * LDA + 8 * kg + 4 * h]);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          c00 = MFMA32F(u[s], p[s], c00);
          c01 = MFMA32F(v[s], p[s], c01);
          c10 = MFMA32F(u[s], q[s], c10);
          c11 = MFMA32F(v[s], q[s], c11);
        }
      }
    } else {
      const u32x4* w0 = wb + (size_t)(2 * wave) * 16 * 3 * 64 + lane;
      const u32x4* w1 = wb + (size_t)(2 * wave + 1) * 16 * 3 * 64 + lane;
#pragma unroll 2
      for (int ks = 0; ks < 16; ++ks) {
        u32x4 P[3], Q[3], U[3], V[3];
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
          P[pc] = w0[(ks * 3 + pc) * 64];
          Q[pc] = w1[(ks * 3 + pc) * 64];
        }
        if (VARIANT == 1) {
          const f32x4 ua = *reinterpret_cast<const f32x4*>(&lds[r * LDA + 16 * ks + 8 * h]);
          const f32x4 ub = *reinterpret_cast<const f32x4*>(&lds[r * LDA + 16 * ks + 8 * h + 4]);
          const f32x4 va = *reinterpret_cast<const f32x4*>(&lds[(32 + r) * LDA + 16 * ks + 8 * h]);
          const f32x4 vb = *reinterpret_cast<const f32x4*>(&lds[(32 + r) * LDA + 16 * ks + 8 * h + 4]);
          auto split8 = [](const f32x4& a, const f32x4& b, u32x4* dst) {
            unsigned p1, p2, p3;
            split2(a[0], a[1], p1, p2, p3); dst[0][0] = p1; dst[1][0] = p2; dst[2][0] = p3;
            split2(a[2], a[3], p1, p2, p3); dst[0][1] = p1; dst[1][1] = p2; dst[2][1] = p3;
            split2(b[0], b[1], p1, p2, p3); dst[0][2] = p1; dst[1][2] = p2; dst[2][2] = p3;
            split2(b[2], b[3], p1, p2, p3); dst[0][3] = p1; dst[1][3] = p2; dst[2][3] = p3;
          };
          split8(ua, ub, U);
          split8(va, vb, V);
        } else {
#pragma unroll
          for (int pc = 0; pc < 3; ++pc) {
            U[pc] = *reinterpret_cast<const u32x4*>(&planes[(pc * R + r) * LDP + 8 * ks + 4 * h]);
            V[pc] = *reinterpret_cast<const u32x4*>(&planes[(pc * R + 32 + r) * LDP + 8 * ks + 4 * h]);
          }
        }
        // small terms first
        constexpr int ia[6] = {1, 0, 2, 0, 1, 0}, ib[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t) {
          c00 = MFMA32B(as_bf16x8(U[ia[t]]), as_bf16x8(P[ib[t]]), c00);
          c01 = MFMA32B(as_bf16x8(V[ia[t]]), as_bf16x8(P[ib[t]]), c01);
          c10 = MFMA32B(as_bf16x8(U[ia[t]]), as_bf16x8(Q[ib[t]]), c10);
          c11 = MFMA32B(as_bf16x8(V[ia[t]]), as_bf16x8(Q[ib[t]]), c11);
        }
      }
    }
  }
  // C layout: element i of a 32x32 accumulator = row (i & 3) + 8 (i >> 2) + 4 h, column r
  float* o = out + (size_t)blockIdx.x * R * N;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
    o[row * N + 64 * wave + r] = c00[i];
    o[(32 + row) * N + 64 * wave + r] = c01[i];
    o[row * N + 64 * wave + 32 + r] = c10[i];
    o[(32 + row) * N + 64 * wave + 32 + r] = c11[i];
  }
}

static uint16_t bf16_rne(float x) {
  uint32_t u;
  memcpy(&u, &x, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static float bf16_f(uint16_t b) {
  uint32_t u = (uint32_t)b << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

int main() {
  const int wgs = 256, reps = 200;
  std::vector<float> A((size_t)wgs * R * K), W((size_t)N * K);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
  for (auto& v : A) v = 2.0f * rnd();            // activations in (-1, 1)
  for (auto& v : W) v = 0.25f * rnd();
  std::vector<float> wf((size_t)8 * 32 * 64 * 4);
  for (int cb = 0; cb < 8; ++cb)
    for (int kg = 0; kg < 32; ++kg)
      for (int l = 0; l < 64; ++l)
        for (int e = 0; e < 4; ++e) wf[(((size_t)cb * 32 + kg) * 64 + l) * 4 + e] = W[(size_t)(cb * 32 + (l & 31)) * K + kg * 8 + 4 * (l >> 5) + e];
  std::vector<uint16_t> wb((size_t)8 * 16 * 3 * 64 * 8);
  for (int cb = 0; cb < 8; ++cb)
    for (int ks = 0; ks < 16; ++ks)
      for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 8; ++j) {
          const float x = W[(size_t)(cb * 32 + (l & 31)) * K + ks * 16 + 8 * (l >> 5) + j];
          const uint16_t p1 = bf16_rne(x);
          const float r1 = x - bf16_f(p1);
          const uint16_t p2 = bf16_rne(r1);
          const float r2 = r1 - bf16_f(p2);
          const uint16_t p3 = bf16_rne(r2);
          const uint16_t pc[3] = {p1, p2, p3};
          for (int q = 0; q < 3; ++q) wb[((((size_t)cb * 16 + ks) * 3 + q) * 64 + l) * 8 + j] = pc[q];
        }
  float *dA, *dwf, *dout;
  uint16_t* dwb;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dwf, wf.size() * 4); hipMalloc(&dwb, wb.size() * 2); hipMalloc(&dout, (size_t)wgs * R * N * 4);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dwf, wf.data(), wf.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dwb, wb.data(), wb.size() * 2, hipMemcpyHostToDevice);
  // float64 reference of tile 0
  std::vector<double> ref((size_t)R * N);
  for (int i = 0; i < R; ++i)
    for (int n = 0; n < N; ++n) {
      double acc = 0.0;
      for (int k = 0; k < K; ++k) acc += (double)A[(size_t)i * K + k] * (double)W[(size_t)n * K + k];
      ref[(size_t)i * N + n] = acc;
    }
  const size_t lds_bytes = std::max((size_t)R * LDA * 4, (size_t)3 * R * (K / 2 + 4) * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<float> out((size_t)R * N);
  double t_ms[3] = {0, 0, 0};
  auto run = [&](int variant, int nreps) {
    switch (variant) {
      case 0: hipLaunchKernelGGL(k_probe<0>, dim3(wgs), dim3(256), lds_bytes, 0, dA, (const f32x4*)dwf, (const u32x4*)dwb, nreps, dout); break;
      case 1: hipLaunchKernelGGL(k_probe<1>, dim3(wgs), dim3(256), lds_bytes, 0, dA, (const f32x4*)dwf, (const u32x4*)dwb, nreps, dout); break;
      default: hipLaunchKernelGGL(k_probe<2>, dim3(wgs), dim3(256), lds_bytes, 0, dA, (const f32x4*)dwf, (const u32x4*)dwb, nreps, dout); break;
    }
  };
  hipFuncSetAttribute((const void*)k_probe<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  hipFuncSetAttribute((const void*)k_probe<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  hipFuncSetAttribute((const void*)k_probe<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  for (int variant = 0; variant < 3; ++variant) {
    run(variant, 1);  // accuracy: one phase
    hipDeviceSynchronize();
    hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
    double emax = 0.0, rmax = 0.0;
    for (size_t i = 0; i < out.size(); ++i) {
      emax = std::max(emax, std::fabs((double)out[i] - ref[i]));
      rmax = std::max(rmax, std::fabs(ref[i]));
    }
    run(variant, 20);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    run(variant, reps);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    t_ms[variant] = ms;
    const double flops = 2.0 * wgs * R * (double)K * N * reps;
    printf("variant %d: %.3f us per 64x256x256 phase per workgroup, %.1f TFLOP/s algorithmic on 256 workgroups, max |err| %.3e (max |C| %.3f, %.2e relative)  [%s]\n",
           variant, 1e3 * ms / reps, flops / (ms * 1e-3) / 1e12, emax, rmax, emax / rmax, hipGetErrorString(hipGetLastError()));
  }
  printf("speed-up over the f32 pipe: in-loop split %.2fx, pre-split activations %.2fx\n", t_ms[0] / t_ms[1], t_ms[0] / t_ms[2]);
  return 0;
}
