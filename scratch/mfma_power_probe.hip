// Probe (not a product path): what the matrix pipes SUSTAIN under the package power cap.  A stream of nothing but
// v_mfma_f32_16x16x32_bf16 (k_chain_train's instruction) on every SIMD of every CU for seconds, operands = random bf16 bit patterns
// (the power of a multiplier array depends on how its inputs toggle: the all-zero run is printed beside it), and the same for
// v_mfma_f32_16x16x4_f32.  Prints TFLOP/s from HIP events over the whole run and the shader clock the stream held
// (s_memtime cycles / wall time).  Run scratch/clock_probe_mfma.sh to sample rocm-smi beside it.
//   hipcc --offload-arch=gfx950 -O3 -o scratch/mfma_power_probe scratch/mfma_power_probe.hip && scratch/mfma_power_probe [seconds]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <ctime>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
// a pair of bf16 with random sign / mantissa and exponents in [2^-8, 2^0): finite, no denormals
__device__ __forceinline__ unsigned rnd_bf16x2(unsigned s) {
  const unsigned h = hash32(s);
  const unsigned lo = (h & 0x807fu) | ((119u + ((h >> 8) & 7u)) << 7);
  const unsigned hi = ((h >> 16) & 0x807fu) | ((119u + ((h >> 28) & 7u)) << 7);
  return lo | (hi << 16);
}

// MODE 0: bf16 16x16x32, random operands (eight different A / B register sets in turn)   1: the same, all operands zero
// MODE 5 / 6: 16x16x32 with the same A and B registers in every MFMA / A fixed and B changing (does switching operands cost power?)
// MODE 4: bf16 32x32x16, random operands (two accumulators in turn)
// MODE 2: f32 16x16x4, random operands                                                   3: bf16 at HALF issue rate (an s_sleep-free gap: 4 v_nop-class VALU between MFMAs)
template <int MODE, int NT>
__global__ __launch_bounds__(NT) void k(int iters, float* out, unsigned long long* cyc) {
  f32x4 acc[4];
  for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 a[4], b[4];
  for (int r = 0; r < 4; ++r)
    for (int i = 0; i < 4; ++i) {
      a[r][i] = MODE == 1 ? 0u : rnd_bf16x2(threadIdx.x * 131u + blockIdx.x * 7919u + 17u * r + i);
      b[r][i] = MODE == 1 ? 0u : rnd_bf16x2(threadIdx.x * 257u + blockIdx.x * 104729u + 29u * r + i + 1000u);
    }
  float fa[4], fb[4];
  for (int r = 0; r < 4; ++r) { fa[r] = __uint_as_float((a[r][0] & 0x807fffffu) | 0x3c000000u); fb[r] = __uint_as_float((b[r][0] & 0x807fffffu) | 0x3c000000u); }
  f32x16 big[2];
  for (int t = 0; t < 2; ++t) for (int i = 0; i < 16; ++i) big[t][i] = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      if (MODE == 4) {   // 16 of these per 32 of the others: the same flops per trip
        if (u & 1) continue;
        big[(u >> 1) & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[(u >> 2) & 3]), __builtin_bit_cast(bf16x8, b[(u >> 1) & 3]), big[(u >> 1) & 1], 0, 0, 0);
        continue;
      }
      f32x4& c = acc[u & 3];
      if (MODE == 2) c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[(u >> 2) & 3], fb[u & 3], c, 0, 0, 0);
      else if (MODE == 5) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[0]), __builtin_bit_cast(bf16x8, b[0]), c, 0, 0, 0);   // the SAME operand registers every time
      else if (MODE == 6) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[0]), __builtin_bit_cast(bf16x8, b[u & 3]), c, 0, 0, 0);   // A fixed, B changes every MFMA
      else c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[(u >> 2) & 3]), __builtin_bit_cast(bf16x8, b[u & 3]), c, 0, 0, 0);
      if (MODE == 3) asm volatile("s_nop 7\n\ts_nop 7");
    }
    // keep the sums finite without leaving the pipe: nothing (products of values < 1, 2^-8 average: sums grow ~ sqrt(n) x 2^-9)
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int t = 0; t < 4; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  for (int t = 0; t < 2; ++t) for (int i = 0; i < 16; ++i) s += big[t][i];
  out[blockIdx.x * NT + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE, int NT>
void run(const char* name, double seconds, float* out, unsigned long long* cyc) {
  const int iters = 20000;   // 640 k MFMAs per wave and launch: ~ 5 ms
  hipLaunchKernelGGL((k<MODE, NT>), dim3(256), dim3(NT), 0, 0, 100, out, cyc);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  // calibrate, then run for `seconds`
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<MODE, NT>), dim3(256), dim3(NT), 0, 0, iters, out, cyc);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms1; hipEventElapsedTime(&ms1, e0, e1);
  const int n = (int)(seconds * 1e3 / ms1) + 1;
  printf("[t=%ld] ", (long)time(nullptr)); fflush(stdout);
  hipEventRecord(e0, 0);
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL((k<MODE, NT>), dim3(256), dim3(NT), 0, 0, iters, out, cyc);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  const double mf = 32.0 * iters;                                  // MFMAs per wave and launch
  const double flop = (MODE == 2 ? 2.0 * 16 * 16 * 4 : 2.0 * 16 * 16 * 32) * mf * (NT / 64) * 256.0 * n;
  const double per_launch_ms = ms / n;
  printf("%-64s %7.1f TFLOP/s sustained over %.1f s   %.1f cycles per MFMA per wave, shader clock %.2f GHz in the last launch\n", name, flop / (ms * 1e-3) * 1e-12,
         ms * 1e-3, (double)h / mf, (double)h / (per_launch_ms * 1e6));
  fflush(stdout);
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 4.0;
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
  run<0, 256>("bf16 16x16x32, random operands, 1 wave / SIMD", seconds, out, cyc);
  run<0, 512>("bf16 16x16x32, random operands, 2 waves / SIMD", seconds, out, cyc);
  run<1, 256>("bf16 16x16x32, all operands zero, 1 wave / SIMD", seconds, out, cyc);
  run<3, 256>("bf16 16x16x32, random operands, ~half issue rate (s_nop gaps)", seconds, out, cyc);
  run<5, 512>("bf16 16x16x32, random operands, the SAME A and B registers every MFMA, 2 waves / SIMD", seconds, out, cyc);
  run<6, 512>("bf16 16x16x32, random operands, A fixed, B changes every MFMA, 2 waves / SIMD", seconds, out, cyc);
  run<4, 256>("bf16 32x32x16, random operands, 1 wave / SIMD (per 2 x 16x16x32 of work)", seconds, out, cyc);
  run<4, 512>("bf16 32x32x16, random operands, 2 waves / SIMD", seconds, out, cyc);
  run<2, 256>("f32 16x16x4, random operands, 1 wave / SIMD", seconds, out, cyc);
  return 0;
}
