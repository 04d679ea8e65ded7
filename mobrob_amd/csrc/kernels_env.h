// Device-resident goal environment: the env-side elementwise math of the reference's EnvWrapper on the GPU
// (SURVEY.md §8f rank 4).  What is mirrored, per environment and step:
//     reward_fn   (/root/reference/src/mobrob/envs/wrapper.py:137-154): |goal - prev_pos| - |goal - pos|, +5 when reached
//                 (+10 more for the drone, wrapper.py:491-496)
//     step        (wrapper.py:156-171): terminated = terminate_on_goal and reached()
//     reached     (wrapper.py:203-207): |pos - goal| < 0.3
//     reset       (wrapper.py:173-201): lazy reset -- a robot that has just reached its goal keeps its pose and only
//                 gets a new goal; otherwise pose <- init_space.sample(); goal <- goal_space.sample(); prev_pos <- pos
//     TimeLimit   (gymnasium wrapper used by get_env, wrapper.py:549-571): truncated once `time_limit` steps elapsed
//     VecEnv auto-reset with terminal_observation / TimeLimit.truncated and the PPO time-limit bootstrap
//                 r += gamma * V(terminal_obs)  (SB3 collect_rollouts; oracle bootstrap_reward)
// The physics underneath (MuJoCo / Bullet in the reference) is NOT reproduced: the robot is the velocity-controlled
// point of mobrob_amd/envs/wrapper.py::KinematicSim (same constants), so that the host VecEnv and this kernel are
// the same task.  One thread per (env, 16-byte observation chunk) for coalesced stores; every chunk thread
// re-derives the env's tiny state update, the chunk-0 thread commits it (state is double buffered).
#pragma once
#include "kernels_generic.h"

namespace mobrob {

constexpr uint32_t kStreamEnvReset = 0x52535431u;
constexpr int kGoalStateFloats = 12;  // pos[3] vel[3] goal[3] ep_return ep_len pad

struct GoalEnvParams {
  int P;                     // position dimensions (2 or 3)
  int terminate_on_goal, time_limit;
  float dt, extent, reach, bonus, extra_bonus, noise;
  float mix[3][32];          // velocity command = mix . clip(action)
};

// Monitor ring: (return, length) of finished episodes.  ep_stats[4] counts the records ever written (a double, like its
// neighbours: one atomic per finished episode), slot = count mod kEpRing.  A record is ONE 64-bit word (two floats) and
// goes out as one 8-byte store: a rollout finishes far more than kEpRing episodes, so pushes k and k + kEpRing land on the
// same slot from different workgroups -- with one store per record they can replace each other but never pair one
// episode's return with another's length.  Slot order is the order of the atomics across independently running
// workgroups, i.e. the ring is a SAMPLE of recently finished episodes (all envs, roughly the last ones), not SB3's
// strictly newest-100 deque.
constexpr int kEpRing = 128;
constexpr int kEpStatsDoubles = 5 + kEpRing;
__device__ __forceinline__ void ep_ring_push(double* ep_stats, float ep_ret, float ep_len) {
  const unsigned slot = (unsigned)((unsigned long long)atomicAdd(&ep_stats[4], 1.0) % (unsigned)kEpRing);
  const unsigned long long rec = (unsigned long long)__float_as_uint(ep_ret) | ((unsigned long long)__float_as_uint(ep_len) << 32);
  reinterpret_cast<unsigned long long*>(ep_stats)[5 + slot] = rec;
}

struct GoalEnvArgs {
  uint64_t seed; uint32_t step_rel; const uint32_t* step_base;
  int N, D, Dp, A;
  GoalEnvParams p;
  const float* act;                    // [N][A] clipped actions of this step
  const float* st_in; float* st_out;   // [N][kGoalStateFloats]
  float* obs_next; float* term_obs;
  const float* prev_dones; float* next_dones; uint8_t* trunc; float* rew_out; float* es_out;
  double* ep_stats;                    // [4] finished episodes, sum of returns, sum of lengths, goals reached;
                                       // then the Monitor ring: [4] records written so far, [5 + k] = (return, length) as two floats
};

struct GoalState {
  float pos[3], vel[3], goal[3], ep_ret;
  int ep_len;
};

__device__ __forceinline__ GoalState goal_load(const float* s) {
  GoalState g;
#pragma unroll
  for (int j = 0; j < 3; ++j) { g.pos[j] = s[j]; g.vel[j] = s[3 + j]; g.goal[j] = s[6 + j]; }
  g.ep_ret = s[9];
  g.ep_len = (int)s[10];
  return g;
}
__device__ __forceinline__ void goal_store(float* s, const GoalState& g) {
#pragma unroll
  for (int j = 0; j < 3; ++j) { s[j] = g.pos[j]; s[3 + j] = g.vel[j]; s[6 + j] = g.goal[j]; }
  s[9] = g.ep_ret;
  s[10] = (float)g.ep_len;
  s[11] = 0.f;
}
// (all loops over position components are fully unrolled over 3 with a `j < P` guard: no dynamic register indexing;
//  HIP's __fmul_rn / __fsub_rn are plain operators after inlining and hipcc contracts across statements
//  (-ffp-contract=fast ignores pragmas), so a product that later feeds a subtraction -- the freshly drawn pose /
//  goal in goal_reset -- is pinned to its rounded value with an empty asm; otherwise goal - extent * u is fused
//  into one fma in one kernel and not in the other.  With that the per-step kernel and the persistent rollout
//  kernel step bit-identical environments.)
__device__ __forceinline__ float rounded(float x) {
  asm volatile("" : "+v"(x));
  return x;
}
__device__ __forceinline__ float goal_dist(const float* a, const float* b, int P) {
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 3; ++j)
    if (j < P) {
      const float d = __fsub_rn(a[j], b[j]);
      s = __fmaf_rn(d, d, s);
    }
  return (float)sqrt((double)s);  // via f64: hipcc may pick a 2.5-ulp f32 sqrt / divide per context
}
// observation features 4c..4c+3 of state g (KinematicSim.obs): [rel / (|rel| + 1e-6), vel, pos, noise ...]
__device__ __forceinline__ f32x4 goal_features(const GoalState& g, int P, int D, int c, const float z[4], float noise) {
  const float d = __fadd_rn(goal_dist(g.goal, g.pos, P), 1e-6f);
  f32x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int f = 4 * c + j;
    float v = rounded(__fmul_rn(noise, z[j]));
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      if (q < P) {
        if (f == q) v = (float)((double)__fsub_rn(g.goal[q], g.pos[q]) / (double)d);
        if (f == P + q) v = g.vel[q];
        if (f == 2 * P + q) v = g.pos[q];
      }
    }
    o[j] = f < D ? v : 0.f;
  }
  return o;
}
struct GoalOutcome {
  float reward;
  bool reached, term, tr, done;
};
// one env step: dynamics + reward + termination rules (no reset)
// (GP: GoalEnvParams, or the same struct behind a kernel-argument reference -- constant address space: the persistent rollout
//  kernels read the parameters, the 3 x 32 mix matrix with its run-time column index among them, from the kernarg segment where
//  they use them instead of holding ~100 scalars in registers through the step loop)
template <class GP>
__device__ __forceinline__ GoalOutcome goal_advance(GoalState& g, const GP& p, const float* act, int A) {
  float cmd[3] = {0.f, 0.f, 0.f};
  for (int k = 0; k < A; ++k) {
    const float a = act[k];
#pragma unroll
    for (int j = 0; j < 3; ++j) cmd[j] = __fmaf_rn(p.mix[j][k], a, cmd[j]);
  }
  const float d0 = goal_dist(g.goal, g.pos, p.P);
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    if (j < p.P) {
      g.vel[j] = __fmaf_rn(0.8f, g.vel[j], __fmul_rn(0.2f, cmd[j]));
      g.pos[j] = fminf(fmaxf(__fmaf_rn(p.dt, g.vel[j], g.pos[j]), -p.extent), p.extent);
    }
  }
  const float d1 = goal_dist(g.goal, g.pos, p.P);
  GoalOutcome o;
  o.reached = d1 < p.reach;
  o.reward = __fadd_rn(__fsub_rn(d0, d1), o.reached ? __fadd_rn(p.bonus, p.extra_bonus) : 0.f);
  o.term = p.terminate_on_goal && o.reached;
  g.ep_len += 1;
  g.ep_ret = __fadd_rn(g.ep_ret, o.reward);
  o.tr = g.ep_len >= p.time_limit && !o.term;
  o.done = o.term || o.tr;
  return o;
}
// EnvWrapper.reset: lazy pose reset, always a new goal
template <class GP>
__device__ __forceinline__ void goal_reset(GoalState& g, const GP& p, bool reached, uint32_t n, uint32_t step,
                                           uint32_t k0, uint32_t k1) {
  const Philox4 a = philox4x32_10(n, 0u, step, kStreamEnvReset, k0, k1);
  const Philox4 b = philox4x32_10(n, 1u, step, kStreamEnvReset, k0, k1);
  const float ua[3] = {u32_to_unit_open(a.x), u32_to_unit_open(a.y), u32_to_unit_open(a.z)};
  const float ub[3] = {u32_to_unit_open(b.x), u32_to_unit_open(b.y), u32_to_unit_open(b.z)};
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    if (j < p.P) {
      if (!reached) {
        g.vel[j] = 0.f;
        g.pos[j] = rounded(__fmul_rn(p.extent, __fsub_rn(ua[j], 0.5f)));      // init_space = [-extent/2, extent/2]
      }
      g.goal[j] = rounded(__fmul_rn(p.extent, __fmaf_rn(2.0f, ub[j], -1.0f)));  // goal_space = [-extent, extent]
    }
  }
  g.ep_ret = 0.f;
  g.ep_len = 0;
}

__global__ __launch_bounds__(256) void k_goal_env_reset(uint64_t seed, int N, int D, int Dp, GoalEnvParams p,
                                                        float* __restrict__ state, float* __restrict__ obs0) {
  const int per = Dp / 4;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * per) return;
  const int n = i / per, c = i - n * per;
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  GoalState g{};
  goal_reset(g, p, false, (uint32_t)n, 0xFFFFFFFFu, k0, k1);
  float z[4];
  box_muller4(philox4x32_10((uint32_t)n, (uint32_t)c, 0xFFFFFFFFu, kStreamEnvObs, k0, k1), z);
  reinterpret_cast<f32x4*>(obs0)[(size_t)n * per + c] = goal_features(g, p.P, D, c, z, p.noise);
  if (c == 0) goal_store(state + (size_t)n * kGoalStateFloats, g);
}

// env.step(clipped actions) + VecEnv auto-reset + rollout_buffer.add scalars + time-limit bootstrap, one launch
__global__ __launch_bounds__(256) void k_goal_env_step_store(GoalEnvArgs a, BootNetArgs bt) {
  extern __shared__ float sm[];  // x[Dp] | activations[width_sum] | red[16] | cnt[4] | env[kBootMaxEnvs] | rew[kBootMaxEnvs]
  float* x = sm;
  float* hbuf = x + a.Dp;
  float* red = hbuf + bt.vn.width_sum;
  int* cnt = reinterpret_cast<int*>(red + 16);
  int* lenv = cnt + 4;
  float* lrew = reinterpret_cast<float*>(lenv + kBootMaxEnvs);
  if (threadIdx.x == 0) *cnt = 0;
  __syncthreads();
  const int per = a.Dp / 4;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t k0 = (uint32_t)a.seed, k1 = (uint32_t)(a.seed >> 32);
  const uint32_t step = a.step_rel + (a.step_base ? *a.step_base : 0u);
  if (i < a.N * per) {
    const int n = i / per, c = i - n * per;
    GoalState g = goal_load(a.st_in + (size_t)n * kGoalStateFloats);
    const GoalOutcome o = goal_advance(g, a.p, a.act + (size_t)n * a.A, a.A);
    float z[4];
    box_muller4(philox4x32_10((uint32_t)n, (uint32_t)c, step, kStreamEnvObs, k0, k1), z);
    f32x4 ob = goal_features(g, a.p.P, a.D, c, z, a.p.noise);  // observation after the step (terminal one if done)
    const float ep_ret = g.ep_ret;
    const int ep_len = g.ep_len;
    if (o.done) {
      if (o.tr) reinterpret_cast<f32x4*>(a.term_obs)[(size_t)n * per + c] = ob;
      goal_reset(g, a.p, o.reached, (uint32_t)n, step, k0, k1);
      box_muller4(philox4x32_10((uint32_t)n, (uint32_t)c, step, kStreamEnvTerm, k0, k1), z);
      ob = goal_features(g, a.p.P, a.D, c, z, a.p.noise);
    }
    reinterpret_cast<f32x4*>(a.obs_next)[(size_t)n * per + c] = ob;
    if (c == 0) {
      goal_store(a.st_out + (size_t)n * kGoalStateFloats, g);
      if (o.tr) {  // reward is written after the bootstrap below
        const int q = atomicAdd(cnt, 1);
        lenv[q] = n;
        lrew[q] = o.reward;
      } else {
        a.rew_out[n] = o.reward;
      }
      a.es_out[n] = a.prev_dones[n];
      a.next_dones[n] = o.done ? 1.f : 0.f;
      a.trunc[n] = o.tr ? 1 : 0;
      if (o.done) {  // Monitor-style episode statistics (diagnostics: order of the atomics is irrelevant)
        atomicAdd(&a.ep_stats[0], 1.0);
        atomicAdd(&a.ep_stats[1], (double)ep_ret);
        atomicAdd(&a.ep_stats[2], (double)ep_len);
        if (o.reached) atomicAdd(&a.ep_stats[3], 1.0);
        ep_ring_push(a.ep_stats, ep_ret, (float)ep_len);
      }
    }
  }
  __syncthreads();
  const int m = *cnt;
  for (int q = 0; q < m; ++q) {
    const int n = lenv[q];
    if ((int)threadIdx.x < per) {  // rebuild the terminal observation of env n in LDS
      const int c = threadIdx.x;
      GoalState g = goal_load(a.st_in + (size_t)n * kGoalStateFloats);
      (void)goal_advance(g, a.p, a.act + (size_t)n * a.A, a.A);
      float z[4];
      box_muller4(philox4x32_10((uint32_t)n, (uint32_t)c, step, kStreamEnvObs, k0, k1), z);
      const f32x4 ob = goal_features(g, a.p.P, a.D, c, z, a.p.noise);
#pragma unroll
      for (int j = 0; j < 4; ++j) x[4 * c + j] = ob[j];
    }
    __syncthreads();
    const float v = value_net_row(x, hbuf, red, bt.vn, a.D);
    if (threadIdx.x == 0) {
      bt.term_val[n] = v;
      a.rew_out[n] = (float)((double)lrew[q] + (double)__fmul_rn(bt.gamma, v));
    }
    __syncthreads();
  }
}

}  // namespace mobrob
