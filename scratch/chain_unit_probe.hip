// Probe (not a product path): the chain unit of k_chain_train with its REAL instruction mix -- six dependent
// v_mfma_f32_16x16x32_bf16, the next unit's three ds_read_b128, and on six of every eight units one LDS-DMA of the weight ring
// (global_load_lds_dwordx4) with its scalar set-up -- to price the non-MFMA instructions one kind at a time (round 5).
//   V = 0  MFMAs + reads                                   (chain_loop_probe's variant 2)
//   V = 1  + the product's DMA group: s_add_u32 / s_addc_u32 / s_mov_b32 m0 / s_nop 0 / global_load_lds_dwordx4
//   V = 2  + the DMA alone, addressed by its instruction offset from ONE m0 / base pair per segment
//   V = 3  V = 1 without the DMA itself (the four scalar instructions only)
//   V = 4  V = 2 + nothing else, but the segment's six DMAs as a burst behind the barrier
// every variant: s_waitcnt vmcnt(6) + s_barrier every eight units (the ring's segment boundary).
// Also checks that a NEGATIVE instruction offset moves the LDS address back as it moves the global address back.
//   hipcc --offload-arch=gfx950 -O3 -o scratch/chain_unit_probe scratch/chain_unit_probe.hip && scratch/chain_unit_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
extern __shared__ __attribute__((aligned(16))) float lds[];
#define M16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), (c), 0, 0, 0)
constexpr int UNIT = 768, SLOTS = 24, SEG = 8;   // floats per ring unit (3 KB), ring slots, units per segment

template <int V>
__global__ __launch_bounds__(256) void k(int iters, const u32x4* __restrict__ pack, float* out, unsigned long long* cyc) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < SLOTS * UNIT; i += 256) lds[i] = 1e-3f * (float)(i & 1023);
  __syncthreads();
  f32x4 acc[16];
  for (int t = 0; t < 16; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 x0, x1, x2, w0, w1, w2, n0, n1, n2;
  for (int i = 0; i < 4; ++i) { x0[i] = 0x3f803f80u + lane; x1[i] = 0x3c003c00u; x2[i] = 0x38003800u; w0[i] = 0x3f003f00u + i; w1[i] = 0x3b003b00u; w2[i] = 0x37003700u; }
  n0 = w0; n1 = w1; n2 = w2;
  int base = 4 * lane;
  asm volatile("" : "+v"(base));
  base = 4 * (base >> 2);
  const unsigned lane16 = 16u * lane;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 48; ++u) {   // two trips round the ring
      __builtin_amdgcn_sched_barrier(0);
      const int i = (u + 1) % SEG, q = (u + 1) / SEG;
      if (i == 0) {
        asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
        if (V == 4) {
          const int u0 = (SEG * (q + 2) + 2 * wave) % SLOTS;
          const unsigned dst = (unsigned)(u0 * UNIT + UNIT) * 4u;
          asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
                       "global_load_lds_dwordx4 %1, %2 offset:-3072\n\tglobal_load_lds_dwordx4 %1, %2 offset:-2048\n\t"
                       "global_load_lds_dwordx4 %1, %2 offset:-1024\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                       "global_load_lds_dwordx4 %1, %2 offset:1024\n\tglobal_load_lds_dwordx4 %1, %2 offset:2048"
                       ::"s"(dst), "v"(lane16), "s"(pack + (size_t)(u0 + 1) * 192) : "memory", "m0");
        }
      }
      if (i < 6) {
        if (V == 1 || V == 3) {   // the product's form: unit u0 = first unit of segment q + 2 + 4 (i / 3) + wave, piece i % 3
          const int u0 = (SEG * (q + 2) + 4 * (i / 3)) % SLOTS;
          const unsigned dst = (unsigned)((u0 + wave) * UNIT + 256 * (i % 3)) * 4u;
          const u32x4* src = pack + (size_t)(u0 + wave) * 192 + 64 * (i % 3);
          if (V == 1) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(lane16), "s"(src) : "memory", "m0");
          else asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t; no dma %1 %2" ::"s"(dst), "v"(lane16), "s"(src) : "memory", "m0");
        }
        if (V == 2) {             // wave w owns units 2 w, 2 w + 1 of the segment: one m0 / base pair, six offsets
          const int u0 = (SEG * (q + 2) + 2 * wave) % SLOTS;
          if (i == 0) {
            const unsigned dst = (unsigned)(u0 * UNIT + UNIT) * 4u;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:-3072" ::"s"(dst), "v"(lane16), "s"(pack + (size_t)(u0 + 1) * 192) : "memory", "m0");
          } else {
            const u32x4* src = pack + (size_t)(u0 + 1) * 192;   // the same expression: the pair stays in scalar registers (or is re-derived: see the ISA)
            if (i == 1) asm volatile("global_load_lds_dwordx4 %0, %1 offset:-2048" ::"v"(lane16), "s"(src) : "memory");
            if (i == 2) asm volatile("global_load_lds_dwordx4 %0, %1 offset:-1024" ::"v"(lane16), "s"(src) : "memory");
            if (i == 3) asm volatile("global_load_lds_dwordx4 %0, %1" ::"v"(lane16), "s"(src) : "memory");
            if (i == 4) asm volatile("global_load_lds_dwordx4 %0, %1 offset:1024" ::"v"(lane16), "s"(src) : "memory");
            if (i == 5) asm volatile("global_load_lds_dwordx4 %0, %1 offset:2048" ::"v"(lane16), "s"(src) : "memory");
          }
        }
      }
      {
        const int slot = (u + 1) % SLOTS;
        n0 = *reinterpret_cast<const u32x4*>(&lds[base + slot * UNIT]);
        n1 = *reinterpret_cast<const u32x4*>(&lds[base + slot * UNIT + 256]);
        n2 = *reinterpret_cast<const u32x4*>(&lds[base + slot * UNIT + 512]);
      }
      f32x4& a = acc[u % 16];
      a = M16(w2, x0, a); a = M16(w1, x1, a); a = M16(w1, x0, a); a = M16(w0, x2, a); a = M16(w0, x1, a); a = M16(w0, x0, a);
      w0 = n0; w1 = n1; w2 = n2;
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0.f;
  for (int t = 0; t < 16; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t0; cyc[1] = t1; }
}

template <int V>
void run(const char* name, const u32x4* pack, float* out, unsigned long long* cyc) {
  const int iters = 700;
  hipFuncSetAttribute((const void*)k<V>, hipFuncAttributeMaxDynamicSharedMemorySize, SLOTS * UNIT * 4);
  hipLaunchKernelGGL((k<V>), dim3(256), dim3(256), SLOTS * UNIT * 4, 0, 10, pack, out, cyc);
  hipDeviceSynchronize();
  hipLaunchKernelGGL((k<V>), dim3(256), dim3(256), SLOTS * UNIT * 4, 0, iters, pack, out, cyc);
  hipDeviceSynchronize();
  unsigned long long hs[2]; hipMemcpy(hs, cyc, 16, hipMemcpyDeviceToHost);
  printf("%-100s %.1f cycles per unit (96 = the matrix pipe's own time)\n", name, (double)(hs[1] - hs[0]) / (48.0 * iters));
}

__global__ __launch_bounds__(64) void k_neg(const u32x4* __restrict__ src, unsigned* out) {
  unsigned* l = reinterpret_cast<unsigned*>(lds);
  const unsigned lane16 = threadIdx.x * 16u;
  for (int i = threadIdx.x; i < 4096; i += 64) l[i] = 0xdeadbeefu;
  __syncthreads();
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:-3072" ::"s"(8192u), "v"(lane16), "s"(src + 192) : "memory", "m0");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 4096; i += 64) out[i] = l[i];
}

int main() {
  std::vector<unsigned> h(SLOTS * UNIT * 2);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3c003c00u + (unsigned)(i & 63);
  unsigned* pack; float* out; unsigned long long* cyc;
  hipMalloc(&pack, h.size() * 4); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 16);
  hipMemcpy(pack, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  {
    std::vector<unsigned> g(2048);
    for (int i = 0; i < 2048; ++i) g[i] = 0x10000u + i;
    unsigned *d, *o;
    hipMalloc(&d, 8192); hipMalloc(&o, 16384);
    hipMemcpy(d, g.data(), 8192, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_neg, dim3(1), dim3(64), 16384, 0, (const u32x4*)d, o);
    std::vector<unsigned> r(4096);
    hipMemcpy(r.data(), o, 16384, hipMemcpyDeviceToHost);
    // base = source word 768 (src + 192 x 16 B), m0 = LDS word 2048; offset -3072 B = -768 words: expect LDS words [1280, 1536) <- source word 0
    for (int b = 0; b < 4096; b += 256)
      if (r[b] != 0xdeadbeefu) printf("negative offset: LDS words [%d, %d) <- source word %d (expected [1280, 1536) <- 0)\n", b, b + 256, (int)(r[b] - 0x10000u));
  }
  const u32x4* p = (const u32x4*)pack;
  run<0>("MFMAs + the next unit's three ds_read_b128", p, out, cyc);
  run<3>("+ s_add / s_addc / s_mov m0 / s_nop on six of eight units (no DMA)", p, out, cyc);
  run<1>("+ the product's DMA group on six of eight units (4 scalar + global_load_lds_dwordx4)", p, out, cyc);
  run<2>("+ the DMA alone on six of eight units (instruction offsets from one m0 / base per segment)", p, out, cyc);
  run<4>("+ the segment's six DMAs as a burst behind the barrier (one m0 / base)", p, out, cyc);
  return 0;
}
