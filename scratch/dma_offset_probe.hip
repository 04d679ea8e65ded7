// Probe: does the instruction offset of global_load_lds_dwordx4 move BOTH the global address and the LDS address?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
extern __shared__ __attribute__((aligned(16))) unsigned lds[];
__global__ __launch_bounds__(64) void k(const u32x4* __restrict__ src, unsigned* out) {
  const unsigned lane16 = threadIdx.x * 16u;
  for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = 0xdeadbeefu;
  __syncthreads();
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024" ::"s"(4096u), "v"(lane16), "s"(src) : "memory", "m0");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 2048; i += 64) out[i] = lds[i];
}
int main() {
  std::vector<unsigned> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = 0x10000u + i;   // words 0..255 = first KB, 256..511 = second KB (offset 1024)
  unsigned *d, *o;
  hipMalloc(&d, 4096); hipMalloc(&o, 8192);
  hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 8192, 0, (const u32x4*)d, o);
  std::vector<unsigned> r(2048);
  hipMemcpy(r.data(), o, 8192, hipMemcpyDeviceToHost);
  for (int base = 0; base < 2048; base += 256) {
    if (r[base] != 0xdeadbeefu) printf("LDS words [%d, %d) <- source word %d (m0 = byte 4096 = word 1024; offset:1024 = 256 words)\n", base, base + 256, (int)(r[base] - 0x10000u));
  }
  return 0;
}
