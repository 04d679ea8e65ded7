// Persistent update kernel for small minibatches of 64-wide networks: ONE launch per epoch.
//
// The reference's own configurations (data/configs/*.yaml: batch_size 100, net_arch 2x64, n_envs 2..16) make
// PPO.train() a chain of 800-1600 optimizer steps of 100 rows each.  One step is four launches of the per-step path
// (k_fused64_train, k_slab64_reduce, k_sqnorm_chunks, k_adam_pack): ~67 us, all of it launch latency and pipeline
// fill.  Here the whole epoch runs inside one launch of TWO workgroups, one per network:
//
//   for every minibatch of the epoch:
//     wave w: rows [32w, 32w+32) of the minibatch -> forward -> loss -> backward        (tile64_train, weights from L2)
//     waves summed through LDS in wave order; wave 0 scatters the gradient to the canonical vector
//     per-tensor sums of squares of the workgroup's own network (same chunk table and order as k_sqnorm_chunks)
//     HAND-OFF: the two workgroups swap their per-tensor norms (<= 7 floats each way) and the value loss sum --
//               the only cross-workgroup data of a step, because the global-norm clip couples the two networks
//     clip coefficient, Adam on the workgroup's own tensors, fragment packs / padded copies rewritten in place
//
// STATUS: correct and bit-identical, but SLOWER than the per-step path it was meant to replace (92 vs 60 us per
// optimizer step on the doggo YAML shape), so the engine uses it only on request (config.persistent_train).  Cycle
// stamps (-DMOBROB_SMALL_STAMPS, scratch/time_small_train.py), us per step: tile 23 (weights streamed from L2 by one
// wave per SIMD: latency bound), wave reduction 7, gradient scatter 9, norms 9, hand-off 2 (+12 waiting for the
// slower partner), clip + Adam + re-pack 37.  The elementwise tail is ~120 instructions for each of 8.3 k parameters
// per network = 1 M lane-instructions per step: >= 6.5 us on ONE CU even at perfect issue, where the per-step path
// spreads it over 65 + 82 + 13 workgroups.  A version that wins must keep the launch count of this kernel AND the
// width of the per-step kernels: the four kernel bodies as phases of one cooperative launch of ~32 workgroups with
// XCD-hierarchical barriers (3 x ~4 us + ~15 us of phases: DESIGN.md 7).
//
// Summation orders are those of the per-step path, so parameters, Adam moments and logged statistics are
// BIT-IDENTICAL to it (tests/test_engine_gpu.py::test_persistent_small_batch_update_is_bit_identical) as long as a
// minibatch is at most one tile per wave and one tile more than a block of the per-step kernel holds (<= 128 rows for
// 58-dim observations, <= 160 for 14-dim ones), which is also the eligibility rule in engine.hip (train_small_ok).
//
// Hand-off protocol (MI355X_MICROARCH.md, inter-workgroup visibility, first row of the sc1 table): the publishing
// lane stores its words with agent-scope relaxed atomics (sc1, write-through), drains vmcnt, then stores the flag
// (the global optimizer-step id, strictly increasing over the life of the engine) the same way; the consumer lane
// polls the flag with sc1 loads and only then loads the words with sc1 loads, and hands them to its workgroup
// through LDS behind a barrier.  Two slots (step parity): a workgroup can be at most one step ahead of its partner.
// The poll is bounded: on timeout the workgroup raises *error and both leave the loop (the host reports it).
#pragma once
#include "kernels_fused64.h"

namespace mobrob {

struct TrainSmallArgs {
  Fused64TrainArgs t;        // rollout storage, loss coefficients, network packs (rows / count / advstat / inv_bg per step below)
  const int* rows;           // [total] permuted device rows of this epoch
  int total, Bl, nmb, nw;    // transitions, rows per minibatch, minibatches in this launch, tile waves = ceil(Bl / 32)
  const double* advstat;     // [nmb][4]
  AdamPackArgs pk;           // canonical p / g / m / v, offsets, packs and padded copies (step scalars filled per step)
  const NormChunk* chunks; int nchunks;
  const float* sched;        // [nmb][2] (lr / bias_correction1, sqrt(bias_correction2)) of each optimizer step
  float* stats;              // [nmb][8] one row of logged statistics per optimizer step
  unsigned long long* mail;  // [2 networks][2 parities][16]: words 0..12 payload, word 15 flag
  unsigned long long step0;  // id of the first optimizer step of this launch
  int* error;
};

__device__ __forceinline__ void mail_store(unsigned long long* p, unsigned long long x) {
  __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long mail_load(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr int kSmallMaxWaves = 5;

template <int DP>
__global__ __launch_bounds__(kSmallMaxWaves * 64, 1) void k_train_small(TrainSmallArgs a) {
  using L = Lay64<DP>;
  constexpr int per = DP / 4, NGW = GR * per / 64;
  const int tid0 = threadIdx.x, lane = tid0 & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int net = blockIdx.x;  // 0 policy, 1 value
  const int NW = a.nw;
  const bool tile_wave = wave < NW;
  const int wb = wave * L::WAVE;
  const int cst = NW * L::WAVE;            // [3][32] per-action constants
  const int scr = cst + 96;                // 32 doubles (8 chunks x 4 waves) | 16 floats norms | 16 floats misc
  double* dsc = reinterpret_cast<double*>(&lds[scr]);
  float* nts = &lds[scr + 64];
  float* misc = &lds[scr + 80];
  const FusedNet W = a.t.net[net];
  Fused64TrainArgs ta = a.t;
  Slab64ReduceArgs sm{};                   // slab position -> canonical index (the mapping k_slab64_reduce uses)
  sm.P = a.pk.P; sm.D = a.pk.D; sm.A = a.t.A;
#pragma unroll
  for (int i = 0; i < 14; ++i) sm.offs[i] = a.pk.offs[i];
  // tensors of this workgroup's network, in canonical order
  const int my_tensors[7] = {net == 0 ? 0 : 5, net == 0 ? 1 : 6, net == 0 ? 2 : 7, net == 0 ? 3 : 8, net == 0 ? 4 : 11,
                             net == 0 ? 9 : 12, net == 0 ? 10 : -1};

  f32x4 xr[NGW];
  {
    const int cnt0 = min(a.Bl, a.total);
#pragma unroll
    for (int u = 0; u < NGW; ++u) {
      const int i = lane + u * 64, rr = i / per, c = i - rr * per;
      xr[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (tile_wave && wave * GR + rr < cnt0)
        xr[u] = ldg16(a.t.obs, (unsigned)a.rows[wave * GR + rr] * (unsigned)(DP * 4) + (unsigned)(c * 16));
    }
  }
#ifdef MOBROB_SMALL_STAMPS
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev = __builtin_readcyclecounter();
#define SSTAMP(k) { const unsigned long long now_ = __builtin_readcyclecounter(); st_acc[k] += now_ - st_prev; st_prev = now_; }
#else
#define SSTAMP(k)
#endif
  bool dead = false;  // hand-off timed out: leave the loop (uniform: decided through LDS)

  for (int mb = 0; mb < a.nmb && !dead; ++mb) {
    const int start = mb * a.Bl;
    const int cnt = min(a.Bl, a.total - start);
    const float inv_bg = 1.0f / (float)cnt;
    ta.inv_bg = inv_bg;
    // ---- per-action constants of the Gaussian head (log_std and the head bias change every step) ----
    if (tid0 < 32) {
      const int k = tid0;
      float iv = 0.f, lc = 0.f, bb = 0.f;
      if (net == 0 && k < a.t.A) {
        const float sd = expf(a.t.log_std[k]);
        iv = 1.0f / (sd * sd);
        lc = logf(sd) + 0.91893853320467274178f;
      }
      if (k < W.head) bb = W.b3[k];
      lds[cst + k] = iv;
      lds[cst + 32 + k] = lc;
      lds[cst + 64 + k] = bb;
    }
    if (tile_wave) lds[wb + L::GACC + lane] = 0.f;
    // entropy term of the logged loss uses the PRE-update log_std (k_sqnorm_chunks does the same)
    float ent_sum = 0.f;
    if (net == 0 && tid0 == 0)
      for (int k = 0; k < a.t.A; ++k) ent_sum += (0.5f + 0.91893853320467274178f) + logf(expf(a.t.log_std[k]));
    __syncthreads();

    float adv_mean = 0.f, adv_sd = 1.f;
    bool adv_on = false;
    {
      const double* st = a.advstat + 4 * (size_t)mb;
      const double n = st[2];
      adv_on = n > 1.0;
      const double m = st[0] / (n > 0 ? n : 1.0);
      double var = adv_on ? (st[1] - n * m * m) / (n - 1.0) : 0.0;
      if (var < 0.0) var = 0.0;
      adv_mean = (float)m;
      adv_sd = (float)sqrt(var);
    }

    SSTAMP(0)
    Grad64 g;
    g.zero();
    if (tile_wave && wave * GR < cnt) {
      const int ncnt = mb + 1 < a.nmb ? min(a.Bl, a.total - start - a.Bl) : 0;
      tile64_train<DP, false>(ta, W, net, wb, cst, tid0, a.rows + start, cnt, wave * GR,
                              ncnt > wave * GR ? a.rows + start + a.Bl : nullptr, ncnt, wave * GR, xr, adv_mean, adv_sd,
                              adv_on, g);
    }

    SSTAMP(1)
    // ---- sum the waves through LDS in wave order (k_fused64_train's block reduction with a runtime wave count) ----
    asm volatile("s_nop 15\n\ts_nop 3");  // last asm MFMA's D -> VALU read
    __syncthreads();
    // the head-bias / log_std sums and the loss sums sit in the tile regions the reduction is about to reuse
    float b3s = 0.f, lss = 0.f;
    if (wave == 0 && lane < 32) {
      for (int w = 0; w < NW; ++w) {
        b3s += lds[w * L::WAVE + L::GACC + lane];
        lss += lds[w * L::WAVE + L::GACC + 32 + lane];
      }
    }
    const float t0 = wave_sum(g.pl), t1 = wave_sum(g.vl), t2 = wave_sum(g.kl), t3 = wave_sum(g.cf);
    __syncthreads();
    // Two rounds of five accumulator tiles: waves 1.. stage theirs ([wave-1][slot][16][64] floats), wave 0 adds them
    // to its own in wave order.  The loop over the partner waves is unrolled to the maximum and masked: adding 0.f
    // is exact, and an idle slot re-reads wave 1's (in-bounds) data.
    auto stage = [&](int slot, const f32x16& acc) {
      if (wave > 0 && tile_wave) {
#pragma unroll
        for (int i = 0; i < 16; ++i) lds[((wave - 1) * 5 + slot) * 1024 + i * 64 + lane] = acc[i];
      }
    };
    auto fold = [&](int slot, f32x16& acc) {
      if (wave == 0 && NW > 1) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          float x[kSmallMaxWaves - 1];
#pragma unroll
          for (int w = 0; w < kSmallMaxWaves - 1; ++w) x[w] = lds[(((w < NW - 1) ? w : 0) * 5 + slot) * 1024 + i * 64 + lane];
          float v = acc[i];
#pragma unroll
          for (int w = 0; w < kSmallMaxWaves - 1; ++w) v += (w < NW - 1) ? x[w] : 0.f;
          acc[i] = v;
        }
      }
    };
    stage(0, g.W2a); stage(1, g.W2b); stage(2, g.W2c); stage(3, g.W2d); stage(4, g.W1a);
    __syncthreads();
    fold(0, g.W2a); fold(1, g.W2b); fold(2, g.W2c); fold(3, g.W2d); fold(4, g.W1a);
    __syncthreads();
    stage(0, g.W1b); stage(1, g.W3a); stage(2, g.W3b);
    if (DP > 32) { stage(3, g.W1c); stage(4, g.W1d); }
    __syncthreads();
    fold(0, g.W1b); fold(1, g.W3a); fold(2, g.W3b);
    if (DP > 32) { fold(3, g.W1c); fold(4, g.W1d); }
    __syncthreads();
    if (wave > 0 && tile_wave) {
      lds[(wave - 1) * 128 + lane] = g.b2;
      lds[(wave - 1) * 128 + 64 + lane] = g.b1;
    }
    if (tile_wave && lane == 0) {
      lds[1024 + wave * 4 + 0] = t0; lds[1024 + wave * 4 + 1] = t1; lds[1024 + wave * 4 + 2] = t2; lds[1024 + wave * 4 + 3] = t3;
    }
    __syncthreads();
    if (wave == 0) {
      for (int w = 0; w < NW - 1; ++w) {
        g.b2 += lds[w * 128 + lane];
        g.b1 += lds[w * 128 + 64 + lane];
      }
      if (lane < 4) {  // pl, vl, kl, clip count of this minibatch (this network's share)
        float loss_sum = 0.f;
        for (int w = 0; w < NW; ++w) loss_sum += lds[1024 + w * 4 + lane];
        misc[8 + lane] = loss_sum;
      }
    }
    __syncthreads();

    SSTAMP(2)
    // ---- fragment order -> canonical gradient vector (what k_slab64_reduce does for one slab): wave 0 lays its
    //      registers out as a slab in LDS, every thread of the workgroup then maps and stores its share ----
    if (wave == 0) {
      auto put = [&](int region, int t, const f32x16& acc) {
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[4 * qd + e];
          *reinterpret_cast<f32x4*>(&lds[region + ((t * 4 + qd) * 64 + lane) * 4]) = v;
        }
      };
      put(s64_w2(), 0, g.W2a); put(s64_w2(), 2, g.W2b); put(s64_w2(), 1, g.W2c); put(s64_w2(), 3, g.W2d);
      put(s64_w1(), 0, g.W1a); put(s64_w1(), 2, g.W1b);
      if (DP > 32) { put(s64_w1(), 1, g.W1c); put(s64_w1(), 3, g.W1d); }
      put(s64_w3(), 0, g.W3a); put(s64_w3(), 1, g.W3b);
      lds[s64_b2() + lane] = g.b2;
      lds[s64_b1() + lane] = g.b1;
      if (lane < 32) {
        lds[s64_b3() + lane] = b3s;
        lds[s64_ls() + lane] = lss;
      }
    }
    __syncthreads();
    {
      float* gv = a.pk.g_out;
      const float ent_g = a.t.ent_coef * (-(float)cnt) * inv_bg;  // entropy bonus gradient on log_std
      for (int pos = tid0; pos < s64_st(); pos += blockDim.x) {
        if (DP <= 32 && pos >= s64_w1() && pos < s64_w3() && (((pos - s64_w1()) >> 10) & 1)) continue;  // dW1 tiles 1, 3 are unused
        const int dst = slab64_to_canonical(sm, net, pos);
        if (dst >= 0) gv[dst] = lds[pos] + (dst < sm.offs[1] ? ent_g : 0.f);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    SSTAMP(3)
    // ---- per-tensor norms of this network (chunk table and summation order of k_sqnorm_chunks + k_adam_pack): one
    //      pass over the network's chunks (no barrier in between), then one thread per tensor folds its chunks ----
    {
      int slot = 0;
      for (int c = 0; c < a.nchunks; ++c) {
        const NormChunk ch = a.chunks[c];
        const bool mine = net == 0 ? (ch.tensor <= 4 || ch.tensor == 9 || ch.tensor == 10)
                                   : (ch.tensor >= 5 && ch.tensor != 9 && ch.tensor != 10);
        if (!mine) continue;
        double v = tid0 < 256 ? chunk_sumsq_thread(a.pk.g, ch, tid0) : 0.0;
        v = wave_sum_d(v);
        if (lane == 0 && wave < 4) dsc[slot * 4 + wave] = v;
        ++slot;
      }
      __syncthreads();
      if (tid0 < 7 && my_tensors[tid0] >= 0) {
        const int t = my_tensors[tid0];
        double ts = 0.0;
        int sl = 0;
        for (int c = 0; c < a.nchunks; ++c) {
          const int ct = a.chunks[c].tensor;
          const bool mine = net == 0 ? (ct <= 4 || ct == 9 || ct == 10) : (ct >= 5 && ct != 9 && ct != 10);
          if (!mine) continue;
          if (ct == t) {
            double r = 0.0;
            for (int i = 0; i < 4; ++i) r += dsc[sl * 4 + i];
            ts += r;
          }
          ++sl;
        }
        nts[t] = (float)sqrt(ts);
      }
      __syncthreads();
    }

    SSTAMP(4)
    // ---- hand-off: swap the per-tensor norms (and the value loss sum) with the partner workgroup ----
    if (tid0 == 0) {
      const unsigned long long id = a.step0 + (unsigned long long)mb;
      unsigned long long* mine = a.mail + ((size_t)net * 2 + (id & 1)) * 16;
      const unsigned long long* theirs = a.mail + ((size_t)(1 - net) * 2 + (id & 1)) * 16;
#pragma unroll
      for (int k = 0; k < 7; ++k)
        if (my_tensors[k] >= 0) mail_store(mine + k, (unsigned long long)__float_as_uint(nts[my_tensors[k]]));
      mail_store(mine + 8, (unsigned long long)__float_as_uint(misc[9]));  // sum of (ret - v)^2 (value workgroup)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // payload drained before the flag
      mail_store(mine + 15, id);
      bool ok = false;
      for (int spin = 0; spin < (1 << 22); ++spin) {
        if (mail_load(theirs + 15) == id) { ok = true; break; }
        if (spin > 64) __builtin_amdgcn_s_sleep(2);
        if ((spin & 1023) == 1023 && __hip_atomic_load(a.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
      }
      misc[4] = ok ? 0.f : 1.f;
      if (!ok) {
        __hip_atomic_store(a.error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        const int their_tensors[7] = {net == 1 ? 0 : 5, net == 1 ? 1 : 6, net == 1 ? 2 : 7, net == 1 ? 3 : 8,
                                      net == 1 ? 4 : 11, net == 1 ? 9 : 12, net == 1 ? 10 : -1};
#pragma unroll
        for (int k = 0; k < 7; ++k)
          if (their_tensors[k] >= 0) nts[their_tensors[k]] = __uint_as_float((unsigned)mail_load(theirs + k));
        if (net == 0) misc[9] = __uint_as_float((unsigned)mail_load(theirs + 8));
        float tot_sq = 0.f;
        for (int t = 0; t < 13; ++t) tot_sq = __fmaf_rn(nts[t], nts[t], tot_sq);  // one rounding per tensor, tensor order
        const float total = sqrtf(tot_sq);
        misc[0] = fminf(a.pk.max_norm / (total + 1e-6f), 1.0f);
        misc[1] = total;
      }
    }
    __syncthreads();
    dead = misc[4] != 0.f;
    if (dead) break;
    const float coef = misc[0];
    SSTAMP(5)
    // ---- logged statistics of this step (policy workgroup; formulas of k_sqnorm_chunks) ----
    if (net == 0 && tid0 == 0) {
      float* row = a.stats + (size_t)mb * 8;
      const float pl = -misc[8] * inv_bg;
      const float vl = misc[9] * inv_bg;
      const float el = -(ent_sum * (float)cnt) * inv_bg;
      row[0] = pl; row[1] = vl; row[2] = el;
      row[3] = pl + a.t.ent_coef * el + a.t.vf_coef * vl;
      row[4] = misc[10] * inv_bg;
      row[5] = misc[11] * inv_bg;
      row[6] = misc[1];
      row[7] = 0.f;
    }
    // ---- clip + Adam + re-pack of this network's tensors: two contiguous canonical ranges; the four operands of
    //      eight elements per thread are loaded before the first dependent store (one memory round trip per batch) ----
    {
      AdamPackArgs pa = a.pk;
      pa.step_size = a.sched[2 * mb];
      pa.bc2_sqrt = a.sched[2 * mb + 1];
      const int lo[2] = {pa.offs[net == 0 ? 0 : 5], pa.offs[net == 0 ? 9 : 11]};
      const int hi[2] = {pa.offs[net == 0 ? 5 : 9], pa.offs[net == 0 ? 11 : 13]};
      const int nt = blockDim.x;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        for (int base = lo[r] + tid0; base < hi[r]; base += 8 * nt) {
          float gg[8], mm[8], vv[8], pp[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int i = base + u * nt;
            const bool in = i < hi[r];
            gg[u] = in ? pa.g[i] : 0.f;
            mm[u] = in ? pa.m[i] : 0.f;
            vv[u] = in ? pa.v[i] : 0.f;
            pp[u] = in ? pa.p[i] : 0.f;
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int i = base + u * nt;
            if (i < hi[r]) adam_pack_apply(pa, i, gg[u], mm[u], vv[u], pp[u], coef);
          }
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // next step: every wave of this workgroup reads the rewritten packs (same CU, same L1)
    SSTAMP(6)
  }
#ifdef MOBROB_SMALL_STAMPS
  if (tid0 == 0)
    for (int k = 0; k < 8; ++k) a.stats[8 * net + k] = (float)st_acc[k];  // diagnostic build: over the rows of steps 0 / 1
#endif
}

inline size_t train_small_lds_bytes(int Dp, int nw) {
  return (size_t)(nw * (GR * (Dp + 4) + 2 * GR * GLDH + GR * FLDO + 64) + 96 + 96) * sizeof(float);
}
// tile waves that fit 160 KB of LDS next to the constants and scratch
inline int train_small_max_waves(int Dp) {
  const int wave = GR * (Dp + 4) + 2 * GR * GLDH + GR * FLDO + 64;
  return std::min(kSmallMaxWaves, (40960 - 96 - 96) / wave);
}

}  // namespace mobrob
