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
//   publish   copy chunk c of the local message into the own slot; every storing wave drains its stores, workgroup barrier,
//             lane 0: SYSTEM-scope release fence, drain, relaxed system-scope store of flags[c] = s
//   wait      lane 0 polls flags[c] of every rank with relaxed system-scope loads until all are >= s (bounded: a dead peer
//             raises the error word instead of hanging the device), SYSTEM-scope acquire fence, drain, workgroup barrier
//   combine   sum chunk c of all ranks IN RANK ORDER into the local message: ((x0 + x1) + x2) + ...  -- the same bits on
//             every rank and from run to run (for two ranks also the bits any other all-reduce produces)
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
};

template <typename T>
struct alignas(16) OneShotVec { T v[16 / sizeof(T)]; };

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
  for (size_t i = threadIdx.x; i < nvec; i += 256)
    *reinterpret_cast<V*>(a.mine + b0 + i * 16) = *reinterpret_cast<const V*>(loc + b0 + i * 16);
  if (threadIdx.x < ntail)
    *reinterpret_cast<T*>(a.mine + tail0 + threadIdx.x * sizeof(T)) = *reinterpret_cast<const T*>(loc + tail0 + threadIdx.x * sizeof(T));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  __shared__ int ok;
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");   // system scope: the peers may sit on another device
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the flag must not overtake the write-back (MI355X_MICROARCH.md, compiler hazard)
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
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ok = good;
  }
  __syncthreads();
  if (!ok) return;  // the message stays local; the host finds the error word at its next synchronisation
  // ---- combine, rank order ----
  for (size_t i = threadIdx.x; i < nvec; i += 256) {
    V acc = *reinterpret_cast<const V*>(a.peer[0] + b0 + i * 16);
    for (int r = 1; r < a.world; ++r) {
      const V x = *reinterpret_cast<const V*>(a.peer[r] + b0 + i * 16);
#pragma unroll
      for (int k = 0; k < kPer; ++k) acc.v[k] = acc.v[k] + x.v[k];
    }
    *reinterpret_cast<V*>(loc + b0 + i * 16) = acc;
  }
  if (threadIdx.x < ntail) {
    const size_t o = tail0 + threadIdx.x * sizeof(T);
    T acc = *reinterpret_cast<const T*>(a.peer[0] + o);
    for (int r = 1; r < a.world; ++r) acc = acc + *reinterpret_cast<const T*>(a.peer[r] + o);
    *reinterpret_cast<T*>(loc + o) = acc;
  }
}

}  // namespace mobrob
