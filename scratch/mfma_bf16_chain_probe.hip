// Probe: cycles per v_mfma_f32_32x32x16_bf16 when 1, 2 or 4 accumulators take turns (dependent-issue latency of the bf16 matrix pipe)
// and for v_mfma_f32_32x32x2_f32 likewise.  One wave per SIMD, no memory traffic.
//   hipcc --offload-arch=gfx950 -O3 -o scratch/mfma_bf16_chain_probe scratch/mfma_bf16_chain_probe.hip && scratch/mfma_bf16_chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NACC, bool BF16>
__global__ __launch_bounds__(256, 1) void k(int iters, float* out, unsigned long long* cyc) {
  f32x16 c[4];
  for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) c[t][i] = 0.f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x - i)); }
  const float fa = 0.001f * threadIdx.x, fb = 0.002f * threadIdx.x;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      f32x16& acc = c[u % NACC];
      if (BF16) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
      else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc, 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int t = 0; t < NACC; ++t) for (int i = 0; i < 16; ++i) s += c[t][i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NACC, bool BF16>
void run(const char* name, float* out, unsigned long long* cyc) {
  const int iters = 2000;
  hipLaunchKernelGGL((k<NACC, BF16>), dim3(256), dim3(256), 0, 0, 10, out, cyc);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<NACC, BF16>), dim3(256), dim3(256), 0, 0, iters, out, cyc);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  const double n = 16.0 * iters;
  printf("%-28s %d accumulator(s): %.1f shader cycles per MFMA (s_memtime), %.2f ns per MFMA per SIMD -> clock %.2f GHz\n", name, NACC,
         (double)h / n, 1e6 * ms / n, (double)h / n / (1e6 * ms / n));
}
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
  run<1, true>("v_mfma_f32_32x32x16_bf16", out, cyc); run<2, true>("v_mfma_f32_32x32x16_bf16", out, cyc); run<4, true>("v_mfma_f32_32x32x16_bf16", out, cyc);
  run<1, false>("v_mfma_f32_32x32x2_f32", out, cyc); run<2, false>("v_mfma_f32_32x32x2_f32", out, cyc); run<4, false>("v_mfma_f32_32x32x2_f32", out, cyc);
  return 0;
}
