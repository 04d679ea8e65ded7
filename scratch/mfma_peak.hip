// What does the f32 MFMA pipe of this MI355X sustain?  Back-to-back v_mfma_f32_32x32x2_f32 on NACC independent
// accumulators, one or two waves per SIMD, nothing else in the loop.  Prints TFLOP/s against the 157.3 TFLOP/s
// data-sheet figure (256 CUs x 4 SIMDs x 64 FLOP/cycle x 2.4 GHz) that bench.py's roofline uses.
//   hipcc --offload-arch=gfx950 -O3 -o scratch/mfma_peak scratch/mfma_peak.hip && scratch/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void k(float* out, int iters, float a, float b) {
  f32x16 c[NACC];
  for (int j = 0; j < NACC; ++j)
    for (int i = 0; i < 16; ++i) c[j][i] = (float)(threadIdx.x + i + j);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int j = 0; j < NACC; ++j) c[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c[j], 0, 0, 0);
  }
  float s = 0.f;
  for (int j = 0; j < NACC; ++j)
    for (int i = 0; i < 16; ++i) s += c[j][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int threads, const char* what) {
  float* out;
  hipMalloc(&out, 256 * 1024 * 4);
  const int iters = 20000, blocks = 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, out, 100, 1.0f, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0f, 0.5f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * (threads / 64) * iters * 8.0 * NACC * (32.0 * 32 * 2 * 2);
  printf("%-44s %7.2f ms  %6.1f TFLOP/s  (%.1f %% of 157.3)\n", what, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100);
  hipFree(out);
}
int main() {
  run<4>(256, "4 accumulators, 1 wave/SIMD");
  run<2>(256, "2 accumulators, 1 wave/SIMD");
  run<1>(256, "1 accumulator (dependent chain), 1 wave/SIMD");
  run<4>(512, "4 accumulators, 2 waves/SIMD");
  run<2>(512, "2 accumulators, 2 waves/SIMD");
  run<4>(256, "4 accumulators, 1 wave/SIMD (again)");
  return 0;
}
