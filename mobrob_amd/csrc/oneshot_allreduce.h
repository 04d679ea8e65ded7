// One-shot all-reduce of a small message over peer-mapped device memory (the 645 KB gradient of a
// data-parallel PPO step is latency-bound on the xGMI mesh, a direct exchange beats a ring).
//
// Every rank owns an exchange buffer (hipMalloc, exported with hipIpcGetMemHandle, opened by the other ranks with
// hipIpcOpenMemHandle: over xGMI on an 8-GPU node, the same physical memory for two test ranks that share one device):
//
//     [2 slots][payload bytes]   the rank's own contribution to message s sits in slot s & 1
//     [kMaxChunks] u64 flags     flags[c] = s once chunk c of message s is published (monotonic sequence numbers)
//
// One launch per message, one workgroup per 16 KB chunk c:
//   publish   copy chunk c of the local message into the own slot with SYSTEM-scope write-through stores (sc0 sc1: the bytes
//             leave the caches with the store); every storing wave drains its stores, workgroup barrier, lane 0: relaxed
//             system-scope store of flags[c] = s.  (Round 3 used plain stores + a system-scope RELEASE fence per workgroup: on
//             this part a release writes back the whole XCD's L2, 41 times per message; MOBROB_ONESHOT_FENCE=1 selects that form.)
//   wait      lane 0 polls flags[c] of every rank with relaxed system-scope loads until all are >= s (bounded: a dead peer
//             raises the error word, and the HOST fails the step at its next synchronisation), workgroup barrier
//   combine   sum chunk c of all ranks IN RANK ORDER into the local message, every remote byte read with system-scope
//             loads (sc0 sc1: past this device's caches): ((x0 + x1) + x2) + ...  -- the same bits on every rank and from
//             run to run (for two ranks also the bits any other all-reduce produces)
// The exchange is validated at set-up (mobrob_ppo_exchange_selfcheck: a known vector through it, compared with the rank-ordered
// sum); a run on real peers uses it only if that came out bit-equal.
// A chunk depends only on the same chunk of the peers, so no rank waits for a whole remote message.
//
// Reuse of a slot: message s + 2 overwrites slot s & 1.  A rank launches message s + 2 after its launch of s + 1 has
// finished, which waited for flags == s + 1 of every peer, which those publish after THEIR launch of message s (the last
// reader of the slot) has finished: stream order on every rank makes two slots enough.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace mobrob {

constexpr int kOneShotMaxRanks = 8;          // one node
constexpr int kOneShotChunkBytes = 16384;    // per workgroup: 256 threads x 4 x 16 B
constexpr int kOneShotMaxChunks = 256;

struct OneShotArgs {
  void* local;                                     // the message, in place (input and output)
  char* mine;                                      // own slot for this message
  const char* peer[kOneShotMaxRanks];              // slot of every rank for this message (peer[rank] == mine)
  unsigned long long* my_flags;                    // [kOneShotMaxChunks]
  const unsigned long long* peer_flags[kOneShotMaxRanks];
  int world, rank;
  size_t bytes;                                    // message size (multiple of sizeof(T))
  unsigned long long seq;
  int* error;                                      // device word: set to seq's low bits + 1 when a peer never arrived
  long long timeout_ticks;                         // of wall_clock64() (100 MHz)
  int fence_form;                                  // 1: round-3 protocol (plain stores + system-scope release / acquire fences)
};

template <typename T>
struct alignas(16) OneShotVec { T v[16 / sizeof(T)]; };
typedef unsigned oneshot_u4 __attribute__((ext_vector_type(4)));

// The 16 bytes at `off` of every rank's slot, system scope (sc0 sc1: past this device's caches).  Loads AND their wait sit in ONE asm
// statement with early-clobber outputs: the compiler sees the registers defined only behind the s_waitcnt, so it cannot copy, spill
// or re-allocate one while its load is in flight (a load in one statement and the wait in a later one leaves exactly that window).
// Worlds of 2, 4 and 8 ranks request everything first and wait once; other sizes wait per load (correct, one round trip each).
__device__ __forceinline__ void oneshot_load_all(oneshot_u4 (&xs)[kOneShotMaxRanks], const OneShotArgs& a, size_t off) {
#pragma unroll
  for (int r = 0; r < kOneShotMaxRanks; ++r) xs[r] = oneshot_u4{0u, 0u, 0u, 0u};   // ranks beyond the world: defined, unused
  if (a.world == 2) {
    asm volatile("global_load_dwordx4 %0, %2, off sc0 sc1\n\tglobal_load_dwordx4 %1, %3, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(xs[0]), "=&v"(xs[1]) : "v"(a.peer[0] + off), "v"(a.peer[1] + off) : "memory");
  } else if (a.world == 4) {
    asm volatile("global_load_dwordx4 %0, %4, off sc0 sc1\n\tglobal_load_dwordx4 %1, %5, off sc0 sc1\n\t"
                 "global_load_dwordx4 %2, %6, off sc0 sc1\n\tglobal_load_dwordx4 %3, %7, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(xs[0]), "=&v"(xs[1]), "=&v"(xs[2]), "=&v"(xs[3])
                 : "v"(a.peer[0] + off), "v"(a.peer[1] + off), "v"(a.peer[2] + off), "v"(a.peer[3] + off) : "memory");
  } else if (a.world == 8) {
    asm volatile("global_load_dwordx4 %0, %8, off sc0 sc1\n\tglobal_load_dwordx4 %1, %9, off sc0 sc1\n\t"
                 "global_load_dwordx4 %2, %10, off sc0 sc1\n\tglobal_load_dwordx4 %3, %11, off sc0 sc1\n\t"
                 "global_load_dwordx4 %4, %12, off sc0 sc1\n\tglobal_load_dwordx4 %5, %13, off sc0 sc1\n\t"
                 "global_load_dwordx4 %6, %14, off sc0 sc1\n\tglobal_load_dwordx4 %7, %15, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(xs[0]), "=&v"(xs[1]), "=&v"(xs[2]), "=&v"(xs[3]), "=&v"(xs[4]), "=&v"(xs[5]), "=&v"(xs[6]), "=&v"(xs[7])
                 : "v"(a.peer[0] + off), "v"(a.peer[1] + off), "v"(a.peer[2] + off), "v"(a.peer[3] + off),
                   "v"(a.peer[4] + off), "v"(a.peer[5] + off), "v"(a.peer[6] + off), "v"(a.peer[7] + off) : "memory");
  } else {
#pragma unroll
    for (int r = 0; r < kOneShotMaxRanks; ++r)
      if (r < a.world)
        asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(xs[r]) : "v"(a.peer[r] + off) : "memory");
  }
}

template <typename T>
__global__ __launch_bounds__(256) void k_oneshot_allreduce(OneShotArgs a) {
  using V = OneShotVec<T>;
  constexpr int kPer = 16 / sizeof(T);
  const int c = blockIdx.x;
  const size_t b0 = (size_t)c * kOneShotChunkBytes;
  const size_t b1 = b0 + kOneShotChunkBytes < a.bytes ? b0 + kOneShotChunkBytes : a.bytes;
  const size_t nvec = (b1 - b0) / 16;               // whole 16-byte units of this chunk ...
  const size_t tail0 = b0 + nvec * 16;              // ... and a tail of single elements (last chunk only)
  const size_t ntail = (b1 - tail0) / sizeof(T);
  char* loc = static_cast<char*>(a.local);
  // ---- publish ----
  if (a.fence_form) {
    for (size_t i = threadIdx.x; i < nvec; i += 256)
      *reinterpret_cast<V*>(a.mine + b0 + i * 16) = *reinterpret_cast<const V*>(loc + b0 + i * 16);
    if (threadIdx.x < ntail)
      *reinterpret_cast<T*>(a.mine + tail0 + threadIdx.x * sizeof(T)) = *reinterpret_cast<const T*>(loc + tail0 + threadIdx.x * sizeof(T));
  } else {
    for (size_t i = threadIdx.x; i < nvec; i += 256) {
      const oneshot_u4 x = *reinterpret_cast<const oneshot_u4*>(loc + b0 + i * 16);
      asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(a.mine + b0 + i * 16), "v"(x) : "memory");
    }
    if (threadIdx.x < ntail)
      __hip_atomic_store(reinterpret_cast<T*>(a.mine + tail0 + threadIdx.x * sizeof(T)), *reinterpret_cast<const T*>(loc + tail0 + threadIdx.x * sizeof(T)),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  __shared__ int ok;
  if (threadIdx.x == 0) {
    if (a.fence_form) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");   // system scope: the peers may sit on another device
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the flag must not overtake the write-back (MI355X_MICROARCH.md, compiler hazard)
    }
    __hip_atomic_store(a.my_flags + c, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // ---- wait ----
    int good = 1;
    const long long t0 = wall_clock64();
    for (int r = 0; r < a.world && good; ++r) {
      if (r == a.rank) continue;
      while (__hip_atomic_load(a.peer_flags[r] + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < a.seq) {
        if (wall_clock64() - t0 > a.timeout_ticks) { good = 0; break; }
        __builtin_amdgcn_s_sleep(8);
      }
    }
    if (!good) {  // pinned host word: a plain store (device atomics on host memory need PCIe atomics), pushed out by the fence below
      *reinterpret_cast<volatile int*>(a.error) = (int)(a.seq & 0x3fffffff) + 1;
      __threadfence_system();
    }
    if (a.fence_form) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    ok = good;
  }
  __syncthreads();
  if (!ok) return;  // the host finds the error word at its next synchronisation and FAILS the step (engine.hip check_async_error)
  // ---- combine, rank order ----
  for (size_t i = threadIdx.x; i < nvec; i += 256) {
    oneshot_u4 xs[kOneShotMaxRanks];
    if (a.fence_form) {
#pragma unroll
      for (int r = 0; r < kOneShotMaxRanks; ++r)
        if (r < a.world) xs[r] = *reinterpret_cast<const oneshot_u4*>(a.peer[r] + b0 + i * 16);
    } else {   // every rank's 16 bytes requested first (system scope: past this device's caches), one wait for all of them
      oneshot_load_all(xs, a, b0 + i * 16);
    }
    V acc = __builtin_bit_cast(V, xs[0]);
#pragma unroll
    for (int r = 1; r < kOneShotMaxRanks; ++r) {
      if (r < a.world) {
        const V x = __builtin_bit_cast(V, xs[r]);
#pragma unroll
        for (int k = 0; k < kPer; ++k) acc.v[k] = acc.v[k] + x.v[k];
      }
    }
    *reinterpret_cast<V*>(loc + b0 + i * 16) = acc;
  }
  if (threadIdx.x < ntail) {
    const size_t o = tail0 + threadIdx.x * sizeof(T);
    T acc = __hip_atomic_load(reinterpret_cast<const T*>(a.peer[0] + o), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    for (int r = 1; r < a.world; ++r) acc = acc + __hip_atomic_load(reinterpret_cast<const T*>(a.peer[r] + o), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    *reinterpret_cast<T*>(loc + o) = acc;
  }
}

}  // namespace mobrob
