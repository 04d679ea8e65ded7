/* Native vectorised HOST environment: thousands of goal-reaching robots stepped in one
 * address space by a thread pool, writing observations straight into a (pinned) staging buffer that the PPO engine
 * uploads with hipMemcpyAsync -- the replacement of the reference's SubprocVecEnv pipes
 * (/root/reference/src/mobrob/rl_control/ppo.py:30-33).  The rules are those of the reference's EnvWrapper:
 *     reward_fn (wrapper.py:137-154; drone +10: :491-496), step / terminate_on_goal (:156-171), lazy reset with a
 *     new goal (:173-201), reached (:203-207), gymnasium TimeLimit as applied by get_env (:549-571), and SB3's
 *     VecEnv auto-reset with terminal_observation / TimeLimit.truncated and Monitor episode statistics.
 * The robot is the kinematic stand-in of mobrob_amd/envs/wrapper.py::KinematicSim (same constants, float64 state):
 * the reference's MuJoCo / Bullet physics is out of scope, so this is the host-side counterpart of
 * csrc/kernels_env.h, not a simulator port.  This file is the ENVIRONMENT (what sits below the drop-in boundary);
 * it contains no PPO arithmetic and is not a fallback for anything in libmobrob_ppo.
 *
 * gcc -O3 -ffast-math -mavx2 -mfma -fopenmp -shared -fPIC -o libmobrob_hostenv.so host_env.c -lm
 *                                                                          (built by __graft_entry__.build())
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  double pos[3], vel[3], goal[3];
  double ep_ret;
  int32_t ep_len;
  uint64_t rng[2]; /* xoroshiro128+ */
} env_state;

typedef struct mobrob_hostenv {
  int32_t n, obs_dim, act_dim, pos_dim, terminate_on_goal, time_limit;
  double dt, extent, reach, bonus, extra_bonus, noise;
  double mix[3][32];
  env_state* st;
  int32_t threads; /* OpenMP team size: one thread per >= 64 envs, at most 16 (a step is ~0.4 us per env; larger teams
                      lose more to wake-up and barrier than they gain -- measured on the 256-core GPU host) */
  /* Monitor statistics since the last read */
  int64_t episodes, goals;
  int64_t ring_written, ring_read;   /* Monitor ring: (return, length) of the last EP_RING finished episodes */
  uint64_t ring[128];                /* one 64-bit word per record (two floats): written with ONE store, so records of
                                        two threads that wrap onto the same slot replace each other but never mix */
  double ret_sum, len_sum;
} mobrob_hostenv;

static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static inline uint64_t next_u64(uint64_t* s) {
  const uint64_t s0 = s[0];
  uint64_t s1 = s[1];
  const uint64_t r = s0 + s1;
  s1 ^= s0;
  s[0] = rotl(s0, 24) ^ s1 ^ (s1 << 16);
  s[1] = rotl(s1, 37);
  return r;
}
static inline double next_unit(uint64_t* s) { return ((double)(next_u64(s) >> 11) + 0.5) * (1.0 / 9007199254740992.0); }
/* Observation padding noise: `count` standard normals scaled by `scale`, two per 64 random bits (Box-Muller in float:
 * the padding features are sensor noise).  The random words are drawn first (the generator is serial), the
 * transcendental part is a branch-free loop over arrays that the compiler vectorises (-O3 -mavx2 -mfma): ln and
 * sin/cos are short polynomials (relative error ~1e-6, far below what a noise source needs).  This is 80 % of an env
 * step for doggo (49 padding features). */
#define MAX_PAIRS 64
static inline void fill_normals(uint64_t* s, float* out, int count, float scale) {
  uint32_t ua[MAX_PAIRS], va[MAX_PAIRS];
  float c[MAX_PAIRS], d[MAX_PAIRS];
  const int np = (count + 1) / 2;
  for (int i = 0; i < np; ++i) {
    const uint64_t r = next_u64(s);
    ua[i] = (uint32_t)(r >> 40);
    va[i] = (uint32_t)((r >> 8) & 0xFFFFFF);
  }
  for (int i = 0; i < np; ++i) {
    /* u in (0,1): -2 ln u = -2 (e ln2 + ln m), m in [1,2); ln m = 2 atanh((m-1)/(m+1)) */
    const float u = ((float)ua[i] + 0.5f) * (1.0f / 16777216.0f);
    union { float f; uint32_t i; } b;
    b.f = u;
    const float e = (float)((int)(b.i >> 23) - 127);
    b.i = (b.i & 0x007FFFFFu) | 0x3F800000u;
    const float q = (b.f - 1.0f) / (b.f + 1.0f), q2 = q * q;
    const float lnm = 2.0f * q * (1.0f + q2 * (1.0f / 3.0f + q2 * (1.0f / 5.0f + q2 * (1.0f / 7.0f + q2 * (1.0f / 9.0f + q2 * (1.0f / 11.0f))))));
    const float mag = scale * sqrtf(-2.0f * (e * 0.69314718056f + lnm));
    /* angle 2 pi v: quadrant k = floor(4v), f = 4v - k in [0,1); sin/cos of f pi/2 by polynomials, then rotate */
    const uint32_t vi = va[i];
    const int k = (int)(vi >> 22);
    const float f = ((float)(vi & 0x3FFFFF) + 0.5f) * (1.0f / 4194304.0f);
    const float x = f * 1.57079632679f, x2 = x * x;
    const float sn = x * (1.0f + x2 * (-1.0f / 6.0f + x2 * (1.0f / 120.0f + x2 * (-1.0f / 5040.0f + x2 * (1.0f / 362880.0f)))));
    const float cs = 1.0f + x2 * (-0.5f + x2 * (1.0f / 24.0f + x2 * (-1.0f / 720.0f + x2 * (1.0f / 40320.0f + x2 * (-1.0f / 3628800.0f)))));
    const float a0 = (k & 1) ? -sn : cs, b0 = (k & 1) ? cs : sn; /* rotate by k quarter turns */
    c[i] = mag * ((k & 2) ? -a0 : a0);
    d[i] = mag * ((k & 2) ? -b0 : b0);
  }
  const int nd = count - np;
  for (int i = 0; i < np; ++i) out[i] = c[i];
  for (int i = 0; i < nd; ++i) out[np + i] = d[i];
}
static inline uint64_t splitmix(uint64_t* x) {
  uint64_t z = (*x += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

static double dist(const double* a, const double* b, int p) {
  double s = 0.0;
  for (int j = 0; j < p; ++j) s += (a[j] - b[j]) * (a[j] - b[j]);
  return sqrt(s);
}

/* KinematicSim.obs: [rel / (|rel| + 1e-6), vel, pos, 0.1 N(0,1) padding] */
static void write_obs(const mobrob_hostenv* e, env_state* s, float* o) {
  const int p = e->pos_dim, d = e->obs_dim;
  const double dn = dist(s->goal, s->pos, p) + 1e-6;
  int k = 0;
  for (int j = 0; j < p && k < d; ++j) o[k++] = (float)((s->goal[j] - s->pos[j]) / dn);
  for (int j = 0; j < p && k < d; ++j) o[k++] = (float)s->vel[j];
  for (int j = 0; j < p && k < d; ++j) o[k++] = (float)s->pos[j];
  if (k < d) fill_normals(s->rng, o + k, d - k, (float)e->noise);
}

/* EnvWrapper.reset: pose only if the goal was not reached (lazy reset), always a new goal */
static void reset_env(const mobrob_hostenv* e, env_state* s, int reached) {
  for (int j = 0; j < e->pos_dim; ++j) {
    if (!reached) {
      s->vel[j] = 0.0;
      s->pos[j] = e->extent * (next_unit(s->rng) - 0.5); /* init_space = [-extent/2, extent/2] */
    }
    s->goal[j] = e->extent * (2.0 * next_unit(s->rng) - 1.0); /* goal_space = [-extent, extent] */
  }
  s->ep_ret = 0.0;
  s->ep_len = 0;
}

mobrob_hostenv* mobrob_hostenv_create(int32_t n, int32_t obs_dim, int32_t act_dim, int32_t pos_dim,
                                      int32_t terminate_on_goal, int32_t time_limit, double dt, double extent,
                                      double reach, double bonus, double extra_bonus, double noise,
                                      const double* mix /* [pos_dim][act_dim] */, uint64_t seed) {
  if (obs_dim - 3 * pos_dim > 2 * MAX_PAIRS) return NULL;
  if (n < 1 || obs_dim < 3 * pos_dim || pos_dim < 1 || pos_dim > 3 || act_dim < 1 || act_dim > 32 || time_limit < 1) return NULL;
  mobrob_hostenv* e = (mobrob_hostenv*)calloc(1, sizeof *e);
  if (!e) return NULL;
  e->n = n; e->obs_dim = obs_dim; e->act_dim = act_dim; e->pos_dim = pos_dim;
  e->terminate_on_goal = terminate_on_goal; e->time_limit = time_limit;
  e->dt = dt; e->extent = extent; e->reach = reach; e->bonus = bonus; e->extra_bonus = extra_bonus; e->noise = noise;
  for (int j = 0; j < pos_dim; ++j)
    for (int k = 0; k < act_dim; ++k) e->mix[j][k] = mix[j * act_dim + k];
  e->st = (env_state*)calloc((size_t)n, sizeof(env_state));
  if (!e->st) { free(e); return NULL; }
  e->threads = n / 64 < 1 ? 1 : n / 64;
  if (e->threads > 16) e->threads = 16;
  if (e->threads > omp_get_num_procs()) e->threads = omp_get_num_procs(); /* not omp_get_max_threads(): torchrun exports OMP_NUM_THREADS=1 */
  uint64_t sm = seed;
  for (int i = 0; i < n; ++i) { /* make_vec_env: env i is seeded with seed + i */
    uint64_t x = sm + (uint64_t)i * 0xD1342543DE82EF95ull;
    e->st[i].rng[0] = splitmix(&x);
    e->st[i].rng[1] = splitmix(&x) | 1ull;
  }
  return e;
}

void mobrob_hostenv_destroy(mobrob_hostenv* e) {
  if (!e) return;
  free(e->st);
  free(e);
}

/* VecEnv.reset(): obs[n][obs_dim] */
void mobrob_hostenv_reset(mobrob_hostenv* e, float* obs) {
#pragma omp parallel for schedule(static) num_threads(e->threads)
  for (int i = 0; i < e->n; ++i) {
    env_state* s = &e->st[i];
    memset(s->pos, 0, sizeof s->pos); memset(s->vel, 0, sizeof s->vel); memset(s->goal, 0, sizeof s->goal);
    reset_env(e, s, 0);
    write_obs(e, s, obs + (size_t)i * e->obs_dim);
  }
  e->episodes = e->goals = 0;
  e->ret_sum = e->len_sum = 0.0;
}

/* VecEnv.step(actions): next obs (post-reset where an episode ended), rewards, dones, TimeLimit.truncated flags and
 * the terminal observation of truncated rows (term_obs rows of other envs are left untouched).  Returns the number
 * of truncated envs.  Episode statistics accumulate in the handle (mobrob_hostenv_episode_stats). */
/* ... for the envs [i0, i1) only: all arrays are the full [n][...] buffers, rows outside the range are not touched.
 * This is what lets a pipelined collector step one half of the robots while the GPU evaluates the policy for the
 * other half (mobrob_ppo_act_part / store_part). */
int32_t mobrob_hostenv_step_range(mobrob_hostenv* e, int32_t i0, int32_t i1, const float* actions, float* obs,
                                  float* rewards, uint8_t* dones, uint8_t* truncated, float* term_obs) {
  int64_t episodes = 0, goals = 0;
  int32_t ntrunc = 0;
  double ret_sum = 0.0, len_sum = 0.0;
  if (i0 < 0) i0 = 0;
  if (i1 > e->n) i1 = e->n;
  int team = (i1 - i0) / 64 < 1 ? 1 : (i1 - i0) / 64;
  if (team > e->threads) team = e->threads;
#pragma omp parallel for schedule(static) num_threads(team) reduction(+ : episodes, goals, ntrunc, ret_sum, len_sum)
  for (int i = i0; i < i1; ++i) {
    env_state* s = &e->st[i];
    const float* a = actions + (size_t)i * e->act_dim;
    const int p = e->pos_dim;
    double cmd[3] = {0.0, 0.0, 0.0};
    for (int k = 0; k < e->act_dim; ++k) {
      double ak = a[k];
      ak = ak < -1.0 ? -1.0 : (ak > 1.0 ? 1.0 : ak);
      for (int j = 0; j < p; ++j) cmd[j] += e->mix[j][k] * ak;
    }
    const double d0 = dist(s->goal, s->pos, p);
    for (int j = 0; j < p; ++j) {
      s->vel[j] = 0.8 * s->vel[j] + 0.2 * cmd[j];
      double x = s->pos[j] + e->dt * s->vel[j];
      s->pos[j] = x < -e->extent ? -e->extent : (x > e->extent ? e->extent : x);
    }
    const double d1 = dist(s->goal, s->pos, p);
    const int reached = d1 < e->reach;
    const double r = d0 - d1 + (reached ? e->bonus + e->extra_bonus : 0.0);
    const int term = e->terminate_on_goal && reached;
    s->ep_len += 1;
    s->ep_ret += r;
    const int tr = s->ep_len >= e->time_limit && !term;
    const int done = term || tr;
    float* o = obs + (size_t)i * e->obs_dim;
    if (done) {
      if (tr) {
        write_obs(e, s, term_obs + (size_t)i * e->obs_dim);
        ntrunc += 1;
      }
      episodes += 1; goals += reached; ret_sum += s->ep_ret; len_sum += s->ep_len;
      {
        int64_t k;
#pragma omp atomic capture
        k = e->ring_written++;
        {
          const float rl[2] = {(float)s->ep_ret, (float)s->ep_len};
          uint64_t rec;
          memcpy(&rec, rl, sizeof rec);
#pragma omp atomic write
          e->ring[k % 128] = rec;
        }
      }
      reset_env(e, s, reached);
    }
    write_obs(e, s, o);
    rewards[i] = (float)r;
    dones[i] = (uint8_t)done;
    truncated[i] = (uint8_t)tr;
  }
  e->episodes += episodes; e->goals += goals; e->ret_sum += ret_sum; e->len_sum += len_sum;
  return ntrunc;
}

int32_t mobrob_hostenv_step(mobrob_hostenv* e, const float* actions, float* obs, float* rewards, uint8_t* dones,
                            uint8_t* truncated, float* term_obs) {
  return mobrob_hostenv_step_range(e, 0, e->n, actions, obs, rewards, dones, truncated, term_obs);
}

/* out[4] = episodes, goals, sum of returns, sum of lengths since the last call with reset != 0 */
void mobrob_hostenv_episode_stats(mobrob_hostenv* e, double* out, int32_t reset) {
  out[0] = (double)e->episodes; out[1] = (double)e->goals; out[2] = e->ret_sum; out[3] = e->len_sum;
  if (reset) { e->episodes = e->goals = 0; e->ret_sum = e->len_sum = 0.0; }
}

/* (return, length) of the episodes finished since the previous call, oldest first, the newest `max_records` at most
 * (the last 128 are kept); returns the count.  Within one step the order of simultaneous finishes is the order
 * the threads got to them. */
int32_t mobrob_hostenv_episode_records(mobrob_hostenv* e, double* out, int32_t max_records) {
  int64_t first = e->ring_read;
  if (e->ring_written - first > 128) first = e->ring_written - 128;
  if (e->ring_written - first > max_records) first = e->ring_written - max_records;
  int32_t n = 0;
  for (int64_t k = first; k < e->ring_written; ++k, ++n) {
    float rl[2];
    memcpy(rl, &e->ring[k % 128], sizeof rl);
    out[2 * n] = rl[0];
    out[2 * n + 1] = rl[1];
  }
  e->ring_read = e->ring_written;
  return n;
}

void mobrob_hostenv_set_threads(mobrob_hostenv* e, int32_t t) { e->threads = t < 1 ? 1 : t; }
int32_t mobrob_hostenv_get_threads(const mobrob_hostenv* e) { return e->threads; }

/* state of env i for tests: pos[3] vel[3] goal[3] */
void mobrob_hostenv_get_state(const mobrob_hostenv* e, int32_t i, double* out9) {
  memcpy(out9, e->st[i].pos, 3 * sizeof(double));
  memcpy(out9 + 3, e->st[i].vel, 3 * sizeof(double));
  memcpy(out9 + 6, e->st[i].goal, 3 * sizeof(double));
}
