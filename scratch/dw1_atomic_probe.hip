// Probe (not a product path): the per-tile dW1 running sums of k_chain_train -- 64 KB per workgroup, 256 workgroups, 8 tiles per
// launch -- kept in L2 three ways:
//   0  as shipped: global_load_dwordx4 (sc1) of the sums, add, global_store_dwordx4 (16 + 16 instructions per lane and tile)
//   1  global_atomic_add_f32 without return (64 instructions per lane and tile; the read-modify-write stays inside the L2)
//   2  global_atomic_pk_add ... not available for f32 pairs: skipped
// between two tiles every workgroup streams 830 KB of "weights" through the L2 (what the ring DMA does), so that the sums see the cache
// pressure they see in the kernel.  Prints microseconds per launch of 8 tiles.
//   hipcc --offload-arch=gfx950 -O3 -o scratch/dw1_atomic_probe scratch/dw1_atomic_probe.hip && scratch/dw1_atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, bool STREAM>
__global__ __launch_bounds__(256) void k(float* slab, const f32x4* __restrict__ packs, float* out, int tiles) {
  float* mine = slab + (size_t)blockIdx.x * 16384;          // 64 KB per workgroup
  const int tid = threadIdx.x;
  f32x4 acc[16];
  float sink = 0.f;
  for (int t = 0; t < tiles; ++t) {
    if (STREAM) {   // 830 KB of packs through this CU: 256 threads x 16 B x 208 = 832 KB, read from one of two 0.83 MB pack sets (per "network")
      const f32x4* p = packs + (size_t)(blockIdx.x & 1) * 53248 + tid;
      f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
      for (int i = 0; i < 208; ++i) s += p[i * 256];
      sink += s[0] + s[1] + s[2] + s[3];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{1e-3f * (tid + i + t), 2e-3f, 3e-3f, 4e-3f};   // "this tile's products"
    if (MODE == 0) {
      if (t > 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          f32x4 v;
          asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(mine + 4 * (i * 256 + tid)) : "memory");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          acc[i] += v;
        }
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) *reinterpret_cast<f32x4*>(mine + 4 * (i * 256 + tid)) = acc[i];
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)   // lane-contiguous dwords: a wave instruction covers 256 contiguous bytes
          __hip_atomic_fetch_add(mine + (4 * i + j) * 256 + tid, acc[i][j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  out[blockIdx.x * 256 + tid] = sink;
}

template <int MODE, bool STREAM>
void run(const char* name, float* slab, const f32x4* packs, float* out) {
  hipMemset(slab, 0, (size_t)256 * 65536);
  hipLaunchKernelGGL((k<MODE, STREAM>), dim3(256), dim3(256), 0, 0, slab, packs, out, 8);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  for (int r = 0; r < 50; ++r) hipLaunchKernelGGL((k<MODE, STREAM>), dim3(256), dim3(256), 0, 0, slab, packs, out, 8);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-88s %.1f us per launch of 8 tiles\n", name, 1e3 * ms / 50);
}

int main() {
  float *slab, *out; f32x4* packs;
  hipMalloc(&slab, (size_t)256 * 65536); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&packs, (size_t)2 * 53248 * 16 + 4096 * 16);
  hipMemset(packs, 0, (size_t)2 * 53248 * 16 + 4096 * 16);
  run<0, false>("load (sc1) + add + store, 16 + 16 x 16 B per lane and tile", slab, packs, out);
  run<1, false>("global_atomic_add_f32 (no return), 64 per lane and tile", slab, packs, out);
  run<0, true>("load + add + store, with 830 KB of packs streamed per tile", slab, packs, out);
  run<1, true>("atomic adds, with 830 KB of packs streamed per tile", slab, packs, out);
  run<1, true>("(again) atomic adds, with the stream", slab, packs, out);
  run<0, true>("(again) load + add + store, with the stream", slab, packs, out);
  return 0;
}
