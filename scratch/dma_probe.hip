// Probe (not a product path): can `global_load_lds_dwordx4` (LDS-DMA) land anywhere in the 160 KB of LDS of a gfx950 CU, and is the
// image lane-linear (lane l's 16 bytes at base + 16 l)?  Prints OK / the first mismatch per target offset.
//   hipcc --offload-arch=gfx950 -O3 -o scratch/dma_probe scratch/dma_probe.hip && scratch/dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
extern __shared__ __attribute__((aligned(16))) unsigned lds[];
__global__ __launch_bounds__(256) void k(const u32x4* __restrict__ src, unsigned* out, int off_words) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 40960; i += 256) lds[i] = 0xdeadbeefu;
  __syncthreads();
  // wave w: 1 KB of src (256 words starting at 256 w) -> lds[off_words + 256 w ...]
  __builtin_amdgcn_global_load_lds((const void*)(src + 64 * wave + lane), (__attribute__((address_space(3))) void*)(lds + off_words + 256 * wave), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += 256) out[i] = lds[off_words + i];
}
int main() {
  std::vector<unsigned> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = 0x10000u + i;
  unsigned *d, *o;
  hipMalloc(&d, 4096); hipMalloc(&o, 4096);
  hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
  const int offs[] = {0, 15360, 16384 - 512, 16384, 20000 / 4 * 4, 32768, 39936};
  for (int off : offs) {
    hipMemset(o, 0, 4096);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 163840, 0, (const u32x4*)d, o, off);
    std::vector<unsigned> r(1024);
    hipError_t e = hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
    int bad = -1;
    for (int i = 0; i < 1024; ++i) if (r[i] != h[i]) { bad = i; break; }
    printf("LDS word offset %6d (byte %6d): %s", off, off * 4, e != hipSuccess ? hipGetErrorString(e) : (bad < 0 ? "OK\n" : "MISMATCH"));
    if (bad >= 0) printf(" at word %d: got %08x want %08x\n", bad, r[bad], h[bad]);
  }
  return 0;
}
