// Probe (not a product path): what paces the unit loop of k_chain_train -- six v_mfma_f32_16x16x32_bf16 per unit into one of sixteen
// accumulator tiles, the unit's three weight pieces read from LDS by all four waves.  Variants:
//   0: MFMAs only, one accumulator chain of six per unit (operands in registers)      1: the same, two units interleaved (a b a b ..)
//   2: variant 0 + the three ds_read_b128 of the next unit                            3: variant 1 + reads
//   4: variant 2 with 32x32x16 MFMAs instead (three per unit: same flops per unit... reference for the issue budget)
//   hipcc --offload-arch=gfx950 -O3 -o scratch/chain_loop_probe scratch/chain_loop_probe.hip && scratch/chain_loop_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
extern __shared__ __attribute__((aligned(16))) float lds[];
#define M16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), (c), 0, 0, 0)
// NT = 256: one wave per SIMD (the kernel as built); NT = 512: TWO waves per SIMD running the same unit stream (round 5: what a
// second wave per SIMD would buy the chain phases -- per-SIMD cycles per unit = the printed per-wave figure / 2)
template <int V, int NT = 256>
__global__ __launch_bounds__(NT, NT / 256) void k(int iters, float* out, unsigned long long* cyc) {
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 24 * 768; i += NT) lds[i] = 1e-3f * (float)(i & 1023);
  __syncthreads();
  f32x4 acc[16];
  for (int t = 0; t < 16; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 x0, x1, x2, w0, w1, w2, n0, n1, n2;
  for (int i = 0; i < 4; ++i) { x0[i] = 0x3f803f80u + lane; x1[i] = 0x3c003c00u; x2[i] = 0x38003800u; w0[i] = 0x3f003f00u + i; w1[i] = 0x3b003b00u; w2[i] = 0x37003700u; }
  n0 = w0; n1 = w1; n2 = w2;
  float fa = 1.0f + lane, fb = 2.0f + lane; unsigned sink = 0;
  int base = 4 * lane;
  asm volatile("" : "+v"(base));
  base = 4 * (base >> 2);
#ifdef PROBE_STAGGER   // the four (eight) waves of the workgroup enter the loop PROBE_STAGGER x wave cycles apart (are they slower in lockstep?)
  for (int k = 0; k < (int)(threadIdx.x >> 6) * (PROBE_STAGGER / 4); ++k) asm volatile("s_nop 0");
#endif
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; u += (V & 1) ? 2 : 1) {
      __builtin_amdgcn_sched_barrier(0);
      if (V & 2) {
        const int slot = (u + 1 + 16 * (it & 1)) % 24;
        n0 = *reinterpret_cast<const u32x4*>(&lds[base + slot * 768]);
        n1 = *reinterpret_cast<const u32x4*>(&lds[base + slot * 768 + 256]);
        n2 = *reinterpret_cast<const u32x4*>(&lds[base + slot * 768 + 512]);
      }
      if (V & 1) {
        f32x4 &a = acc[u], &b = acc[u + 1];
        a = M16(w2, x0, a); b = M16(w2, x0, b); a = M16(w1, x1, a); b = M16(w1, x1, b); a = M16(w1, x0, a); b = M16(w1, x0, b);
        a = M16(w0, x2, a); b = M16(w0, x2, b); a = M16(w0, x1, a); b = M16(w0, x1, b); a = M16(w0, x0, a); b = M16(w0, x0, b);
      } else {
        f32x4& a = acc[u];
        a = M16(w2, x0, a); a = M16(w1, x1, a); a = M16(w1, x0, a); a = M16(w0, x2, a); a = M16(w0, x1, a); a = M16(w0, x0, a);
      }
      if (V & 2) { w0 = n0; w1 = n1; w2 = n2; }
      if (V & 12) {   // side work of the real loop: half a pair-split (5-6 VALU) per unit, or a whole one
        constexpr int reps = ((V & 12) == 12) ? 3 : (V & 8) ? 2 : 1;
#pragma unroll
        for (int r = 0; r < reps; ++r) {
          unsigned p = __builtin_amdgcn_perm(__float_as_uint(fa), __float_as_uint(fb), 0x07060302u);
          fa = fa - __uint_as_float(p << 16); fb = fb - __uint_as_float(p & 0xffff0000u);
          fa = fa * 1.0001f; fb = fb + fa; sink ^= p;
        }
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int t = 0; t < 16; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  out[blockIdx.x * NT + threadIdx.x] = s + fa + fb + (float)sink;
  // every wave of workgroup 0 records its own span: with two waves per SIMD the older wave wins the arbitration and runs at nearly
  // its solo pace -- the SIMD's time per unit is (last end - first start) / units of BOTH its waves, not the oldest wave's span
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { cyc[2 * (threadIdx.x >> 6)] = t0; cyc[2 * (threadIdx.x >> 6) + 1] = t1; }
}
template <int V, int NT = 256>
void run(const char* name, float* out, unsigned long long* cyc) {
  const int iters = 2000;
  hipFuncSetAttribute((const void*)k<V, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 24 * 768 * 4);
  hipLaunchKernelGGL((k<V, NT>), dim3(256), dim3(NT), 24 * 768 * 4, 0, 10, out, cyc);
  hipDeviceSynchronize();
  hipLaunchKernelGGL((k<V, NT>), dim3(256), dim3(NT), 24 * 768 * 4, 0, iters, out, cyc);
  hipDeviceSynchronize();
  unsigned long long hs[16]; hipMemcpy(hs, cyc, 16 * 8, hipMemcpyDeviceToHost);
  unsigned long long lo = hs[0], hi = hs[1], own = hs[1] - hs[0];
  for (int w = 0; w < NT / 64; ++w) { if (hs[2 * w] < lo) lo = hs[2 * w]; if (hs[2 * w + 1] > hi) hi = hs[2 * w + 1]; }
  const unsigned long long h = (NT == 256) ? own : (hi - lo);
  if (NT == 256) printf("%-72s %.1f cycles per unit of six MFMAs (96 = the matrix pipe's own time)\n", name, (double)h / (16.0 * iters));
  else printf("%-72s %.1f cycles per unit on the SIMD's matrix pipe (96 = its own time; wave 0's own span %.1f per unit)\n", name, (double)h / (32.0 * iters), (double)own / (16.0 * iters));
}
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 16 * 8);
  run<0>("six dependent 16x16x32 MFMAs per unit, operands in registers", out, cyc);
  run<1>("two units interleaved (independent neighbours), operands in registers", out, cyc);
  run<2>("six dependent MFMAs + the next unit's three ds_read_b128", out, cyc);
  run<3>("two units interleaved + reads (one weight fragment feeds both: not the kernel's data flow)", out, cyc);
  run<6>("six dependent MFMAs + reads + 6 VALU per unit", out, cyc);
  run<10>("six dependent MFMAs + reads + 12 VALU per unit", out, cyc);
  run<4>("six dependent MFMAs + 6 VALU per unit (no reads)", out, cyc);
  printf("-- two waves per SIMD (512-thread workgroups), the same streams --\n");
  run<0, 512>("2 waves/SIMD: six dependent MFMAs per unit, operands in registers", out, cyc);
  run<2, 512>("2 waves/SIMD: + the next unit's three ds_read_b128", out, cyc);
  run<6, 512>("2 waves/SIMD: + reads + 6 VALU per unit", out, cyc);
  run<10, 512>("2 waves/SIMD: + reads + 12 VALU per unit", out, cyc);
  run<14, 512>("2 waves/SIMD: + reads + 18 VALU per unit", out, cyc);
  run<14>("1 wave/SIMD:  + reads + 18 VALU per unit", out, cyc);
  return 0;
}
