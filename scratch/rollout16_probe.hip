// Probe (not product code): what would the GEMM part of a rollout step cost with 16-ROW tiles on all 256 CUs and the policy's
// weights STATIONARY in the register file?  ("measure, do not estimate".)
//
// One workgroup per 16 environments, four waves; wave w owns hidden columns 64w..64w+63 of both hidden layers:
//   W1 slice [64 K][64 cols]  = 16 k-steps x 4 column tiles of v_mfma_f32_16x16x4_f32 B operands =  64 registers per lane
//   W2 slice [256 K][64 cols] = 64 k-steps x 4 column tiles                                      = 256 registers per lane
//   head     [64 K of this wave][16 cols] = 16 k-steps                                           =  16 registers per lane
// Per step: layer 1 (64 MFMAs) -> tanh -> LDS -> barrier -> layer 2 (256 MFMAs) -> tanh -> LDS -> barrier -> head (16 MFMAs,
// K split over the waves) -> LDS -> barrier.  No sampling, no env, no stores: the matrix part only, to be compared with the
// 12.2 us per step the GEMM phases of k_rollout_persistent take on 128 CUs (scratch/time_rollout.py, ROLL_SKIP=153).
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/rollout16_probe scratch/rollout16_probe.hip && gpurun_out/rollout16_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)

constexpr int R = 16, DP = 64, H = 256, LDX = DP + 4, LDH = H + 4;

__device__ __forceinline__ float fast_tanh_scaled(float xs) {
  const float t = __builtin_amdgcn_exp2f(xs);
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(t + 1.0f);
}

template <int ROWS_PER_WG_TILE>
__global__ __launch_bounds__(256, 1) void k_probe(const float* __restrict__ w1, const float* __restrict__ w2,
                                                  const float* __restrict__ w3, const float* __restrict__ x0, int steps,
                                                  float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) float X[R * LDX];
  __shared__ __attribute__((aligned(16))) float H1[R * LDH];
  __shared__ __attribute__((aligned(16))) float H2[R * LDH];
  __shared__ float HD[4][R * 17];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  // stationary weights: fragment-packed [k-step][tile][lane]
  float W1[16][4], W2[64][4], W3[16];
#pragma unroll
  for (int s = 0; s < 16; ++s)
#pragma unroll
    for (int c = 0; c < 4; ++c) W1[s][c] = w1[((wave * 16 + s) * 4 + c) * 64 + lane];
#pragma unroll
  for (int s = 0; s < 64; ++s)
#pragma unroll
    for (int c = 0; c < 4; ++c) W2[s][c] = w2[((wave * 64 + s) * 4 + c) * 64 + lane];
#pragma unroll
  for (int s = 0; s < 16; ++s) W3[s] = w3[(wave * 16 + s) * 64 + lane];
  for (int i = tid; i < R * DP; i += 256) X[(i / DP) * LDX + (i % DP)] = x0[blockIdx.x * R * DP + i];
  __syncthreads();
  float keep = 0.f;
  for (int t = 0; t < steps; ++t) {
    {  // layer 1: A = X[16][64]; lane (i16, g) holds k = 16 kg + 4 g + s for s = 0..3 (one b128 read per 4 k-steps)
      f32x4 c[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
      for (int kg = 0; kg < 4; ++kg) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(&X[i16 * LDX + 16 * kg + 4 * g]);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int q = 0; q < 4; ++q) c[q] = MFMA16(a[s], W1[4 * kg + s][q], c[q]);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) H1[(4 * g + e) * LDH + 64 * wave + 16 * q + i16] = fast_tanh_scaled(c[q][e]);
    }
    __syncthreads();
    {  // layer 2: K = 256
      f32x4 c[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
      for (int kg = 0; kg < 16; ++kg) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(&H1[i16 * LDH + 16 * kg + 4 * g]);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int q = 0; q < 4; ++q) c[q] = MFMA16(a[s], W2[4 * kg + s][q], c[q]);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) H2[(4 * g + e) * LDH + 64 * wave + 16 * q + i16] = fast_tanh_scaled(c[q][e]);
    }
    __syncthreads();
    {  // head: this wave's K slice [64 wave, 64 wave + 64)
      f32x4 c = {0, 0, 0, 0};
#pragma unroll
      for (int kg = 0; kg < 4; ++kg) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(&H2[i16 * LDH + 64 * wave + 16 * kg + 4 * g]);
#pragma unroll
        for (int s = 0; s < 4; ++s) c = MFMA16(a[s], W3[4 * kg + s], c);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) HD[wave][(4 * g + e) * 17 + i16] = c[e];
    }
    __syncthreads();
    if (tid < R * 16) {  // "next observation": something that depends on the head so that nothing is optimised away
      const int rr = tid >> 4, k = tid & 15;
      const float m = (HD[0][rr * 17 + k] + HD[1][rr * 17 + k]) + (HD[2][rr * 17 + k] + HD[3][rr * 17 + k]);
      X[rr * LDX + k] = 0.5f * X[rr * LDX + k] + 0.01f * m;
      keep += m;
    }
    __syncthreads();
  }
  if (tid < R * 16) out[blockIdx.x * 256 + tid] = keep;
}

int main() {
  const int wgs = 256, steps = 1000;
  std::vector<float> h1(4 * 16 * 4 * 64), h2(4 * 64 * 4 * 64), h3(4 * 16 * 64), hx(wgs * R * DP);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
  for (auto& v : h1) v = 0.3f * rnd();
  for (auto& v : h2) v = 0.15f * rnd();
  for (auto& v : h3) v = 0.1f * rnd();
  for (auto& v : hx) v = rnd();
  float *d1, *d2, *d3, *dx, *dout;
  hipMalloc(&d1, h1.size() * 4); hipMalloc(&d2, h2.size() * 4); hipMalloc(&d3, h3.size() * 4); hipMalloc(&dx, hx.size() * 4);
  hipMalloc(&dout, wgs * 256 * 4);
  hipMemcpy(d1, h1.data(), h1.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(d2, h2.data(), h2.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(d3, h3.data(), h3.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int grid : {256, 128}) {
    hipLaunchKernelGGL(k_probe<16>, dim3(grid), dim3(256), 0, 0, d1, d2, d3, dx, 10, dout);
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(k_probe<16>, dim3(grid), dim3(256), 0, 0, d1, d2, d3, dx, steps, dout);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    const double flops = 2.0 * grid * R * (64.0 * 256 + 256.0 * 256 + 256.0 * 16) * steps;
    printf("grid %d (%d envs): %.2f us per step, %.1f TFLOP/s on the padded shapes (err %s)\n", grid, grid * R, 1e3 * ms / steps,
           flops / (ms * 1e-3) / 1e12, hipGetErrorString(hipGetLastError()));
  }
  return 0;
}
