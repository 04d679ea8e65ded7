// What a device-wide dependency costs on this part in the geometries of the PPO update (VERDICT r5 "next" item 1): the same two-phase
// step -- every workgroup publishes a slab, then every workgroup reads its 1/G slice of ALL slabs (the shape of gradient -> slab reduce)
// -- run (a) as two launches per step, kernel boundaries as the synchronisation (what the engine does), and (b) inside ONE persistent
// launch with a grid barrier between the phases (what a co-operative one-launch-per-epoch kernel would do): flat monotonic counter, and
// the XCD-hierarchical form of MI355X_MICROARCH.md "barrier-xcd".  Slabs are published write-through (sc1 stores, drained, no release
// fence) and read with sc1 loads in (b) -- the cheapest valid hand-off of the guide -- and with plain stores / loads in (a).
//   hipcc --offload-arch=gfx950 -O3 -o scratch/grid_sync_probe scratch/grid_sync_probe.hip && scratch/grid_sync_probe
// Output: microseconds per step (2 synchronisations + both bodies) for every geometry; (b) - (a) over 2 = barrier minus boundary.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void st16_sc1(f4* p, f4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ f4 ld16_sc1_nowait(const f4* p) {   // the value is valid behind wait8() only
  f4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void wait8(f4 (&v)[8]) {   // ties the wait to the registers: no use is scheduled in front of it
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])::"memory");
}

// ---- bodies ----
template <bool COH>
__device__ __forceinline__ void publish(f4* slabs, int q4, int step) {   // q4 float4s per workgroup
  f4* s = slabs + (size_t)blockIdx.x * q4;
  const f4 v = {(float)step, (float)blockIdx.x, 1.f, 2.f};
  for (int i = threadIdx.x; i < q4; i += blockDim.x) {
    if (COH) st16_sc1(s + i, v); else s[i] = v;
  }
}
template <bool COH>
__device__ __forceinline__ float consume(const f4* slabs, int q4, int G) {   // this workgroup's 1/G slice of every slab, eight loads in flight
  const int per = (q4 + G - 1) / G, i0 = blockIdx.x * per;
  float acc = 0.f;
  for (int i = i0 + threadIdx.x; i < min(q4, i0 + per); i += blockDim.x)
    for (int g = 0; g < G; g += 8) {
      f4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const f4* p = slabs + (size_t)min(g + u, G - 1) * q4 + i;
        v[u] = COH ? ld16_sc1_nowait(p) : *p;
      }
      if (COH) wait8(v);
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += (g + u < G) ? v[u][0] + v[u][1] : 0.f;
    }
  return acc;
}
__global__ void k_publish(f4* slabs, int q4, int step) { publish<false>(slabs, q4, step); }
__global__ void k_consume(const f4* slabs, int q4, int G, float* out) {
  const float a = consume<false>(slabs, q4, G);
  if (a == -1.f) out[0] = a;
}

// ---- grid barriers ----
struct Bar { unsigned* flat; unsigned* xcc_cnt; unsigned* xcc_gen; unsigned* top; unsigned* census; };
__device__ __forceinline__ void barrier_flat(unsigned* ctr, unsigned target) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
}
// per-XCD counter (own 128-byte line each); the XCD's last arriver adds to the top counter, waits for all eight XCDs and bumps its
// XCD's generation word; everybody else polls its XCD's generation word
__device__ __forceinline__ void barrier_xcd(const Bar& b, int xcc, unsigned n_on_xcc, unsigned nxcc, unsigned round) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(b.xcc_cnt + 32 * xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1 == n_on_xcc * round) {
      __hip_atomic_fetch_add(b.top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(b.top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nxcc * round) __builtin_amdgcn_s_sleep(1);
      __hip_atomic_store(b.xcc_gen + 32 * xcc, round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      while (__hip_atomic_load(b.xcc_gen + 32 * xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < round) __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
}
template <int KIND>   // 0 flat counter, 1 XCD-hierarchical
__global__ void k_persistent(f4* slabs, int q4, int G, int steps, Bar b, float* out) {
  __shared__ unsigned sh[2];
  int xcc = 0;
  unsigned n_on = 0, nx = 0, flat_round = 0;
  if (KIND == 1) {   // census: workgroups per XCD (placement is not a contract: counted, then agreed through one flat barrier)
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    xcc = (int)(id & 7u);
    if (threadIdx.x == 0) __hip_atomic_fetch_add(b.census + xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    barrier_flat(b.flat, (unsigned)G * ++flat_round);
    if (threadIdx.x == 0) {
      unsigned n = 0;
      for (int x = 0; x < 8; ++x) n += __hip_atomic_load(b.census + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
      sh[0] = __hip_atomic_load(b.census + xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      sh[1] = n;
    }
    __syncthreads();
    n_on = sh[0]; nx = sh[1];
  }
  float acc = 0.f;
  unsigned round = 0;
  for (int s = 0; s < steps; ++s) {
    publish<true>(slabs, q4, s);
    if (KIND == 0) barrier_flat(b.flat, (unsigned)G * ++flat_round); else barrier_xcd(b, xcc, n_on, nx, ++round);
    acc += consume<true>(slabs, q4, G);
    if (KIND == 0) barrier_flat(b.flat, (unsigned)G * ++flat_round); else barrier_xcd(b, xcc, n_on, nx, ++round);
  }
  if (acc == -1.f) out[0] = acc;
}

int main() {
  HC(hipSetDevice(0));
  hipStream_t st;
  HC(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
  unsigned* words; float* out;
  HC(hipMalloc((void**)&words, 4096 * 4)); HC(hipMalloc((void**)&out, 256));
  struct Geo { const char* what; int G, threads; size_t slab_bytes; int lds; };   // lds: dynamic LDS per workgroup, what fixes the workgroups per CU
  const Geo geos[] = {
      {"ref YAML shape: 8 gradient workgroups, 42 KB slabs (k_split64_train)", 8, 256, 41792, 100 << 10},
      {"82 workgroups (k_slab64_reduce's grid), 42 KB slabs", 82, 256, 41792, 100 << 10},
      {"config 2: 512 workgroups, two per CU, 42 KB slabs (k_pair64_train)", 512, 256, 41792, 70 << 10},
      {"headline: 256 workgroups, one per CU, 354 KB slabs (k_chain_train)", 256, 256, 362752, 100 << 10},
      {"256 workgroups, nothing published", 256, 256, 0, 100 << 10},
      {"512 workgroups, nothing published", 512, 256, 0, 70 << 10},
  };
  for (const void* f : {(const void*)k_publish, (const void*)k_consume, (const void*)k_persistent<0>, (const void*)k_persistent<1>})
    HC(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 100 << 10));
  const int steps = 200;
  printf("%-78s %12s %14s %14s\n", "geometry (per step: publish -> sync -> read 1/G of every slab -> sync)", "two launches", "flat barrier", "XCD barrier");
  for (const Geo& g : geos) {
    const int q4 = (int)(g.slab_bytes / 16);
    f4* slabs;
    HC(hipMalloc((void**)&slabs, (size_t)g.G * (q4 ? q4 : 1) * 16));
    float us[3] = {0, 0, 0};
    for (int rep = 0; rep < 3; ++rep) {   // the last repetition counts
      HC(hipEventRecord(e0, st));
      for (int s = 0; s < steps; ++s) {
        hipLaunchKernelGGL(k_publish, dim3(g.G), dim3(g.threads), g.lds, st, slabs, q4, s);
        hipLaunchKernelGGL(k_consume, dim3(g.G), dim3(g.threads), g.lds, st, slabs, q4, g.G, out);
      }
      HC(hipEventRecord(e1, st));
      HC(hipEventSynchronize(e1));
      float ms;
      HC(hipEventElapsedTime(&ms, e0, e1));
      us[0] = 1e3f * ms / steps;
      for (int kind = 0; kind < 2; ++kind) {
        HC(hipMemsetAsync(words, 0, 4096 * 4, st));
        Bar b{words, words + 64, words + 64 + 8 * 32, words + 64 + 16 * 32, words + 64 + 17 * 32};
        HC(hipEventRecord(e0, st));
        if (kind == 0) hipLaunchKernelGGL(k_persistent<0>, dim3(g.G), dim3(g.threads), g.lds, st, slabs, q4, g.G, steps, b, out);
        else hipLaunchKernelGGL(k_persistent<1>, dim3(g.G), dim3(g.threads), g.lds, st, slabs, q4, g.G, steps, b, out);
        HC(hipEventRecord(e1, st));
        HC(hipEventSynchronize(e1));
        HC(hipEventElapsedTime(&ms, e0, e1));
        us[1 + kind] = 1e3f * ms / steps;
      }
    }
    printf("%-78s %9.2f us %11.2f us %11.2f us   -> per sync, barrier - boundary: flat %+.2f, XCD %+.2f us\n", g.what, us[0], us[1], us[2],
           0.5f * (us[1] - us[0]), 0.5f * (us[2] - us[0]));
    HC(hipFree(slabs));
  }
  return 0;
}
