// k_epoch64: ONE co-operative launch per PPO epoch for small minibatches of the 64-wide networks -- the shape of every reference config
// (/root/reference/data/configs/*-ppo.yaml:12-23: batch_size 100, net_arch [64, 64]; SB3 PPO.train reached through
// /root/reference/src/mobrob/rl_control/ppo.py:73-74: 800 - 1600 dependent optimizer steps of four tiles per train()).
//
// Per optimizer step the engine launches k_split64_train -> k_slab64_reduce -> k_adam_pack: 11.4 + 4.4 + 5.9 us of kernels and three
// boundaries of 2.15 us (profiles/r6/ref16_kernel_stats.csv) -- half of the step is the launch / drain / dispatch of three tiny
// grids.  Here the SAME three bodies (split64_tile, slab64_reduce_block, adam_pack_block: the device functions the three kernels
// are wrappers of) run as phases of one persistent launch, a grid barrier behind each:
//
//   for every minibatch:   A  workgroups 0 .. 2 tiles - 1: gradient of one (tile, network), slab out
//                          |  barrier
//                          B  workgroups 0 .. 81: one 256-position block of the fixed-order slab reduction + its norm records
//                          |  barrier
//                          C  the same workgroups: clip coefficient from the records, Adam and packs of the element each thread reduced
//                          |  barrier
//
// Same arithmetic in the same order: parameters, moments and logged statistics are the three-launch path's BIT FOR BIT
// (tests/test_engine_gpu.py::test_epoch_kernel_is_bit_identical_to_the_three_launch_update).
//
// Hand-offs between workgroups inside the launch follow MI355X_MICROARCH.md ("inter-workgroup visibility", first row of the sc1 table):
// every handed-off byte is stored AND loaded at agent scope (relaxed atomics = sc1: write-through, past the vector L1 -- ldc / stc /
// ldg16c / stg16c, COH = true), every storing wave drains its stores (s_waitcnt vmcnt(0)), a workgroup barrier, then ONE lane adds to
// the arrival counter and polls it with sc1 loads; the other waves load behind the workgroup barrier that lane joins afterwards.
// One workgroup per CU (the launch asks for more than half of a CU's LDS), at most as many workgroups as CUs, launched with
// hipLaunchCooperativeKernel (grid checked against residency).  Every poll is bounded: a workgroup that waits longer than
// `timeout_ticks` raises the abort word, everybody leaves, the host fails the update at its next synchronisation.
// Why only this family: with large grids and real payloads a grid barrier costs what a kernel boundary costs or more
// (profiles/r6/grid_sync_probe.txt); it pays where the kernels are tiny.
#pragma once
#include "kernels_split64.h"

namespace mobrob {

constexpr int kEpochRedBlocks = (10 * 1024 + 64 + 64 + 32 + 32 + 8 + 255) / 256;   // ceil(s64_size() / 256) = 41 (checked on the host)

struct Epoch64Args {
  Fused64TrainArgs tr;     // gradient phase; rows / count / advstat / inv_bg are derived per minibatch in the kernel
  Slab64ReduceArgs rd;     // reduction phase; nblocks / b_local / inv_bg likewise
  AdamPackArgs ad;         // clip + Adam + packs; step_size / bc2_sqrt / statistics row likewise (ad.st.loss_sums != null: it logs the step's statistics)
  const int* rows;         // [total] permuted storage rows of the epoch
  const double* advstat;   // [nmb][4]
  int total, bl, nmb, world;
  const float* step_consts;  // [nmb][2] step_size, sqrt(bias correction 2) of every optimizer step (host float64 arithmetic, as k_adam_pack's launcher)
  const int* stats_idx;      // [nmb] row of the statistics ring every step logs into
  float* stats;              // [cap][8]
  unsigned* barrier;         // 1024 words, zero at launch: [0] flat counter, [1] abort word, per-XCD counters / generations (epoch_barrier_xcd)
  int* error_host;           // pinned: raised with the abort word (read by the host after its next synchronisation)
  long long timeout_ticks;   // wall_clock64() ticks (100 MHz) a workgroup waits at one barrier
};

// arrive + wait; true: the launch was aborted (uniform over the workgroup).  `under_wait()`: what every thread may do between its
// workgroup's arrival and the end of the wait -- loads that depend on nothing the barrier orders (their latency runs under the poll;
// requested in FRONT of the barrier they would hold the arrival back: the wait below drains loads and stores alike).
struct EpochNothing { __device__ __forceinline__ void operator()() const {} };
template <class TE, class F = EpochNothing>
__device__ __forceinline__ bool epoch_barrier(const TE& ea, unsigned target, F under_wait = F{}) {
  __shared__ int dead_s;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave: its (write-through) stores have left
  __syncthreads();
  if (threadIdx.x != 0) under_wait();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(ea.barrier, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const long long w0 = wall_clock64();
    int dead = 0;
    while (__hip_atomic_load(ea.barrier, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (__hip_atomic_load(ea.barrier + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { dead = 1; break; }
      if (wall_clock64() - w0 > ea.timeout_ticks) {
        __hip_atomic_store(ea.barrier + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(ea.error_host, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        dead = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    dead_s = dead;
  }
  if (threadIdx.x == 0) under_wait();   // (the polling lane: behind its wait -- one more round trip for one lane of the workgroup)
  __syncthreads();
  return dead_s != 0;
}

// The launch's arguments as they sit in the kernarg segment, re-materialised per phase (as the rollout kernels do, kernels_rollout.h):
// by value they are ~120 pointers and scalars that the compiler loads once and then keeps through the minibatch loop -- 155 of them parked
// in lanes of vector registers and read back (v_readlane) on every phase's critical path.
typedef const __attribute__((address_space(4))) Epoch64Args* EpochArgsK;
__device__ __forceinline__ EpochArgsK epoch_kernargs() {
  EpochArgsK p = (EpochArgsK)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return p;
}
#ifdef MOBROB_EPOCH_STAMPS   // diagnostic build: workgroup 0 sums wall_clock64() (100 MHz) intervals per phase over the launch -> tr.stamps[0..6]
#define ESTAMP(k) { const long long now_ = wall_clock64(); if (vb == 0 && threadIdx.x == 0) est[k] += now_ - eprev; if (vb == 81 && threadIdx.x == 0) est2[k] += now_ - eprev; eprev = now_; }
#else
#define ESTAMP(k)
#endif
#define ea (*eap)
// The same barrier, XCD-hierarchical (MI355X_MICROARCH.md "barrier-xcd"; scratch/grid_sync_probe.hip): a workgroup arrives at ITS XCD's
// counter (a 128-byte line of its own); the XCD's last arriver adds to the top counter, waits for all XCDs there and raises its XCD's
// generation word; everybody else polls that word.  With 82 workgroups arriving together the flat counter took ~4 us per barrier
// (82 adds and 82 pollers on one line), this form ~2.  Words (unsigned) of ea.barrier: [0] flat counter (the census barrier at launch),
// [1] abort, [32] top, [64 + x] census of XCD x, [128 + 32 x] counter of XCD x, [384 + 32 x] generation of XCD x.
template <class TE, class F = EpochNothing>
__device__ __forceinline__ bool epoch_barrier_xcd(const TE& q, int xcc, unsigned n_on_xcc, unsigned n_xcc, unsigned round, F under_wait = F{}) {
  __shared__ int dead_x;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x != 0) under_wait();
  if (threadIdx.x == 0) {
    unsigned* cnt = q.barrier + 128 + 32 * xcc;
    unsigned* gen = q.barrier + 384 + 32 * xcc;
    unsigned* top = q.barrier + 32;
    const long long w0 = wall_clock64();
    int dead = 0;
    auto wait_for = [&](unsigned* word, unsigned want) {
      while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        if (__hip_atomic_load(q.barrier + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { dead = 1; return; }
        if (wall_clock64() - w0 > q.timeout_ticks) {
          __hip_atomic_store(q.barrier + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(q.error_host, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          dead = 1;
          return;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    };
    const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1 == n_on_xcc * round) {   // this XCD's last arriver of the round
      __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      wait_for(top, n_xcc * round);
      __hip_atomic_store(gen, dead ? 0xFFFFFFFFu : round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (an aborted leader frees its pollers at once)
    } else {
      wait_for(gen, round);
      if (__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0xFFFFFFFFu) dead = 1;
    }
    dead_x = dead;
    under_wait();
  }
  __syncthreads();
  return dead_x != 0;
}

template <int DP, int NJ>
__global__ __launch_bounds__(256, 1) void k_epoch64(Epoch64Args ea_by_value) {
  const unsigned G = gridDim.x;
  const int vb = (int)blockIdx.x;
  unsigned round = 0;
  const int nmb = ea_by_value.nmb;
  // census: workgroups per XCD (placement is no contract: counted, and agreed through one flat barrier)
  int xcc;
  unsigned n_on_xcc, n_xcc;
  {
    __shared__ unsigned census_s[2];
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    xcc = (int)(id & 7u);
    if (threadIdx.x == 0) __hip_atomic_fetch_add(ea_by_value.barrier + 64 + xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (epoch_barrier(ea_by_value, G)) return;
    if (threadIdx.x == 0) {
      unsigned n = 0;
      for (int x = 0; x < 8; ++x) n += __hip_atomic_load(ea_by_value.barrier + 64 + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u ? 1u : 0u;
      census_s[0] = __hip_atomic_load(ea_by_value.barrier + 64 + xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      census_s[1] = n;
    }
    __syncthreads();
    n_on_xcc = census_s[0]; n_xcc = census_s[1];
  }
#ifdef MOBROB_EPOCH_STAMPS
  long long est[8] = {0, 0, 0, 0, 0, 0, 0, 0}, est2[8] = {0, 0, 0, 0, 0, 0, 0, 0}, eprev = wall_clock64();
#endif
  for (int mb = 0; mb < nmb; ++mb) {
    EpochArgsK eap = epoch_kernargs();
    const int start = mb * ea.bl;
    const int B = min(ea.bl, ea.total - start);
    const float inv_bg = 1.0f / (float)((long long)B * ea.world);   // (engine.hip mobrob_ppo_minibatch_grad)
    const int ntiles = (B + GR - 1) / GR;
    // The statistics duties -- the entropy term of the PRE-update log_std, the statistics row, zeroing the loss sums -- belong to the LAST
    // reduction workgroup (its block holds the slab's 200 tail positions and it idles through phase A; on workgroup 0, a gradient
    // workgroup, the twelve agent-scope loads of log_std sat on the step's critical path).  The read happens HERE: behind the previous
    // step's last barrier and two barriers in front of the phase in which some thread updates log_std.
    constexpr int kDutyWg = 2 * kEpochRedBlocks - 1;
    float ent = 0.f;
    if (vb == kDutyWg && threadIdx.x == 0 && ea.ad.st.loss_sums != nullptr) ent = entropy_of_log_std<true>(ea.ad.st.log_std, ea.ad.st.n_act);
    // ---- A: gradient of one (tile, network) per workgroup ----
    // (the per-minibatch fields travel BESIDE the argument structs: a modified local copy of one of them would live in scratch memory)
    if (vb < 2 * ntiles) split64_tile<DP, NJ, true>(ea.tr, vb, ea.rows + start, B, ea.advstat + 4 * (size_t)mb, inv_bg);
    ESTAMP(0)
    if (epoch_barrier_xcd(ea, xcc, n_on_xcc, n_xcc, ++round)) return;
    ESTAMP(1)
    eap = epoch_kernargs();
    // ---- B: fixed-order slab reduction (the per-tile slabs folded in k_fused64_train's wave grouping) + norm records.  What a thread
    //      reduced STAYS in its registers: after the barrier it applies clip + Adam to that very parameter (the three-launch path
    //      hands the gradient vector from the reduction's 82 blocks to k_adam_pack's ceil(P / 256) through memory; Adam is elementwise
    //      once the clip coefficient is known, so who updates which element changes no bit). ----
    int dst = -1, fidx = -1;
    float acc = 0.f, m_in = 0.f, v_in = 0.f, p_in = 0.f;
    if (vb < 2 * kEpochRedBlocks) {
      slab64_reduce_block<true>(ea.rd, vb % kEpochRedBlocks, vb / kEpochRedBlocks, kEpochRedBlocks, 2 * ntiles, (float)B, inv_bg, dst, acc);
      if (dst >= ea.ad.P) dst = -1;   // (the loss sums behind the gradient vector are no parameters)
    }
    ESTAMP(2)
    // (this thread's moments and parameter -- written by nobody but itself -- are requested UNDER the barrier's wait)
    if (epoch_barrier_xcd(ea, xcc, n_on_xcc, n_xcc, ++round, [&]() {
          if (dst >= 0) { m_in = ldc<true>(ea.ad.m + dst); v_in = ldc<true>(ea.ad.v + dst); p_in = ldc<true>(ea.ad.p + dst); }
          if ((int)threadIdx.x < ea.ad.fold_start[13] && ea.ad.fold_start[13] <= 128) fidx = ea.ad.fold_idx[threadIdx.x];   // (the host's constant table)
        })) return;
    ESTAMP(3)
    eap = epoch_kernargs();
    // ---- C: clip coefficient from the norm records, Adam, packs ----
    if (vb < 2 * kEpochRedBlocks)
      adam_pack_block_at<true>(ea.ad, (vb == kDutyWg ? 0 : 256 * (vb + 1)) + (int)threadIdx.x, dst, acc, m_in, v_in, p_in, ea.step_consts[2 * mb], ea.step_consts[2 * mb + 1],
                               ea.stats + 8 * (size_t)ea.stats_idx[mb], inv_bg, ent, fidx);
    ESTAMP(4)
    if (epoch_barrier_xcd(ea, xcc, n_on_xcc, n_xcc, ++round)) return;
    ESTAMP(5)
  }
#ifdef MOBROB_EPOCH_STAMPS
  if (vb == 0 && threadIdx.x == 0) {
    for (int k = 0; k < 6; ++k) atomicAdd(&ea_by_value.tr.stamps[k], (unsigned long long)est[k]);
    atomicAdd(&ea_by_value.tr.stamps[6], (unsigned long long)nmb);
  }
  if (vb == 81 && threadIdx.x == 0)
    for (int k = 0; k < 6; ++k) atomicAdd(&ea_by_value.tr.stamps[8 + k], (unsigned long long)est2[k]);
#endif
}
#undef ea

}  // namespace mobrob
