// Fused fast path for hidden width 256 (BASELINE configs 3-4: doggo 58/12, 2x256).
//
// One persistent 256-thread workgroup per CU (4 waves, one per SIMD, up to 512 unified VGPR/AGPR each), specialised on one of the two
// independent networks (policy / value).  For each 64-row tile of the minibatch the whole
// forward -> loss -> backward chain stays on chip:
//     LDS (160 KB):  X tile [64][Dp+4] | h1 [64][260] | h2 [64][260] | head/dout [64][36] | row scalars
//     activations never touch HBM; dz2 / dz1 overwrite h2 / h1 in place
//     weights are streamed from L2 in an MFMA-fragment-ordered packing (1 KiB per wave load)
//     dW2 (256x256) is accumulated in registers across all tiles of the workgroup (256 registers per lane),
//     dW1 / dW3 / biases likewise; one slab store per workgroup at the end, then a deterministic
//     slab reduction kernel (no float atomics -> run-to-run reproducible gradients).
// All contractions use v_mfma_f32_32x32x2_f32 (exact f32).  Fragment maps: see kernels_generic.h.
#pragma once
#include "device_utils.h"

namespace mobrob {

constexpr int FH = 256;          // hidden width of the fused path
constexpr int FR = 64;           // rows per tile
constexpr int FLDH = FH + 4;     // padded LDS row stride of h1/h2 (conflict-free ds_read_b128: 260 % 64 == 4)
constexpr int FLDO = 36;         // head tile [64][32] + pad
constexpr int FTHREADS = 256;   // 4 waves: wave w owns output columns [64w, 64w+64)

// packed weights of one network (device pointers)
struct FusedNet {
  const f32x4* W1f;  // [H/32][Dp/8][64]   fwd pack of W1 [H][Dp]
  const f32x4* W2f;  // [H/32][H/8][64]    fwd pack of W2 [H][H]
  const f32x4* W3f;  // [1][H/8][64]       fwd pack of head [32 (zero padded)][H]
  const f32x4* W3h;  // [H/16][64]         16x16x4 fwd pack of head [16 (zero padded)][H] (heads <= 16 wide)
  const f32x4* W2b;  // [H/32][H/8][64]    bwd pack: B[k=n][j] = W2[n][j]
  const f32x4* W3b;  // [H/32][32/8][64]   bwd pack: B[k=a][j] = head[a][j], a < 32 zero padded
  const float* b1s; const float* b2s;  // hidden biases pre-multiplied by kTanhScale (see fast_tanh_scaled)
  const float* b3;                     // canonical head bias
  int head;          // A for the policy net, 1 for the value net
  // x3 packs of the two hidden layers (forward-only kernels; null: f32 matrix pipe), see gemm_x3_r32
  const unsigned* W1x;  // [H/32][Dp/16][3][64] x 16 bytes
  const unsigned* W2x;  // [H/32][H/16][3][64] x 16 bytes
  const unsigned* W2bx; // the same for the backward operand B[k = n][j] = W2[n][j] (dh1 = dz2 . W2), unscaled
  // chain packs (k_chain_train, kernels_chain.h; null: not maintained): A operands of v_mfma_f32_16x16x32_bf16 in the
  // permuted k order of the register chain, [k step][neuron tile 16][piece 3][lane 64] x 16 bytes
  const unsigned* W1c;  // layer 1 (scaled)
  const unsigned* W2c;  // layer 2 (scaled)
  const unsigned* W2bc; // dh1 = W2^T dz2
  const float* W3c;     // head forward, float32 16x16x4: [tile 16][lane 64][4]
  const float* W3bc;    // dh2 = W3^T dout, float32 16x16x4: [tile 16][lane 64][4]
};

struct FusedTrainArgs {
  FusedNet net[2];               // 0 = policy, 1 = value
  // rollout storage
  const float* obs; int Dp;
  int A;
  const float* rec; int RW;      // training records [T*N][RW]: what the loss stage reads of a row, on one line (train_rec_*)
  const int* rows;               // permuted row indices of this minibatch
  int count;                     // rows in this minibatch (local)
  const float* log_std;
  const double* advstat;         // (sum, sumsq, n, -) of the GLOBAL minibatch
  int normalize;
  float clip, vf_coef, ent_coef, inv_bg;
  float clip_vf;                 // clip_range_vf (< 0: none)
  float* slabs;                  // [gridDim.x][slab_floats]
  int slab_floats;
  float* sums;                   // [8] loss statistics (written by k_slab_reduce from the slabs' loss entries)
  unsigned long long* stamps;    // diagnostic build only (MOBROB_STAMPS): per-phase cycle sums
};

// Training records.  The loss stage needs, of every row of a minibatch, the action taken, the old log-probability, the
// advantage, the return and the old value prediction: five arrays, i.e. five random cache lines per row for 64 useful
// bytes (the PMC passes of rounds 1-2 counted 2 x 30.6 MB fetched per launch against 19 MB of rows).  They are packed
// once per train() into one record per row, a whole number of 64-byte lines:
//   [0, A) action | zeros | [RW-4] old log-prob | [RW-3] advantage | [RW-2] return | [RW-1] old value
__host__ __device__ inline int train_rec_width(int A) { return (A + 4 + 15) / 16 * 16; }
struct TrainRecArgs {
  const float* actions; const float* old_logp; const float* adv; const float* ret; const float* values;
  int A, RW, rows;
  float* rec;
};
__global__ __launch_bounds__(256) void k_build_train_records(TrainRecArgs a) {
  const int per = a.RW / 4;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)a.rows * per) return;
  const int row = (int)(i / per), g = (int)(i - (long long)row * per);
  f32x4 v;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int col = 4 * g + c;
    float x = 0.f;
    if (col < a.A) x = a.actions[(size_t)row * a.A + col];
    else if (col == a.RW - 4) x = a.old_logp[row];
    else if (col == a.RW - 3) x = a.adv[row];
    else if (col == a.RW - 2) x = a.ret[row];
    else if (col == a.RW - 1) x = a.values[row];
    v[c] = x;
  }
  *reinterpret_cast<f32x4*>(a.rec + (size_t)row * a.RW + 4 * g) = v;
}

// slab layout (floats): dW2 [H][H] | dW1 [H][64] | dW3 [32][H] | db2 [H] | db1 [H] | db3 [32] | dls [32] | loss sums [8]
// The three weight regions are stored in MFMA *fragment order* (the slab is private scratch, only k_slab_reduce
// reads it): wave w, 32x32 tile t, accumulator register i, lane l  ->  ((w*NT + t)*4 + i/4)*256 + l*4 + i%4,
// so that every lane moves its 16 registers of a tile with four coalesced 16-byte accesses.
//   dW2: NT = 16, t = ib*8 + jb   (neuron block ib of the wave's 64, input block jb of 8)
//   dW1: NT = 4,  t = ib*2 + jb   (input block jb of 2; columns >= Dp hold junk and are never read)
//   dW3: NT = 2,  t = jb          (rows = 32 head outputs, columns 64w + 32jb + r)
__host__ __device__ inline int frag_off(int w, int nt, int t, int i, int lane) {
  return ((w * nt + t) * 4 + (i >> 2)) * 256 + lane * 4 + (i & 3);
}
// inverse C-layout map: row within a 32x32 tile -> (accumulator register i, lane half h)
__host__ __device__ inline void row_to_ih(int row, int* i, int* h) {
  *h = (row >> 2) & 1;
  *i = (row & 3) + 4 * (row >> 3);
}
__host__ __device__ inline int slab_off_w2() { return 0; }
__host__ __device__ inline int slab_off_w1() { return FH * FH; }
__host__ __device__ inline int slab_off_w3(int) { return FH * FH + FH * 64; }
__host__ __device__ inline int slab_off_b2(int Dp) { return slab_off_w3(Dp) + 32 * FH; }
__host__ __device__ inline int slab_off_b1(int Dp) { return slab_off_b2(Dp) + FH; }
__host__ __device__ inline int slab_off_b3(int Dp) { return slab_off_b1(Dp) + FH; }
__host__ __device__ inline int slab_off_ls(int Dp) { return slab_off_b3(Dp) + 32; }
__host__ __device__ inline int slab_off_st(int Dp) { return slab_off_ls(Dp) + 32; }  // loss sums: pl, vl, kl, clip count
__host__ __device__ inline int slab_size(int Dp) { return slab_off_st(Dp) + 8; }

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// tanh(x) = 1 - 2 / (exp(2x) + 1) on the hardware exp2/rcp units (v_exp_f32, v_rcp_f32: ~1 ulp each).
// Absolute error <= ~2e-7 over the whole range (saturates correctly to +-1); the relative error grows for
// |x| < 1e-3 where tanh(x) ~ x, which is irrelevant at the 1e-4 parity tolerance of O(1) activations.
__device__ __forceinline__ float fast_tanh(float x) {
  const float t = __builtin_amdgcn_exp2f(x * 2.88539008177792681472f);  // exp(2x)
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(t + 1.0f);
}
// The forward weight packs of the hidden layers and their biases are stored pre-multiplied by 2*log2(e), and the
// GEMM accumulators start from the (scaled) bias, so the layer epilogue is exp2 / add / rcp / fma per value.
constexpr float kTanhScale = 2.88539008177792681472f;
__device__ __forceinline__ float fast_tanh_scaled(float xs) {  // xs = 2*log2(e) * x
  const float t = __builtin_amdgcn_exp2f(xs);
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(t + 1.0f);
}
__device__ __forceinline__ f32x16 splat16(float v) {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = v;
  return z;
}

// All LDS accesses go through ONE extern array with integer (float-unit) offsets, so that every address is
// "per-lane base + compile-time constant" and folds into the DS instructions' 16-bit immediate.  Per-lane bases
// are passed through opaque() inside the tile loop: without it LLVM's LICM hoists hundreds of loop-invariant
// address computations out of the tile loop and spills them.
extern __shared__ __attribute__((aligned(16))) float lds[];
__device__ __forceinline__ int opaque(int x) {
  asm volatile("" : "+v"(x));
  return x;
}
// Global accesses as "uniform base (SGPR pair) + 32-bit unsigned byte offset (one VGPR)": keeps per-lane 64-bit
// pointers (two VGPRs each, hoisted out of the tile loop and spilled) out of the register file.
__device__ __forceinline__ unsigned opaque_u(unsigned x) {
  asm volatile("" : "+v"(x));
  return x;
}
// Workgroup barrier for LDS hand-offs only.  __syncthreads() is a workgroup-scope fence as well: it drains vmcnt, i.e. it waits for
// every global load AND STORE the wave has in flight (a store's acknowledgement comes from L2 / HBM) -- which a hand-off through LDS
// does not need.  Global stores that only a later kernel reads need no wait inside this one.
#define LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
__device__ __forceinline__ f32x4 ldg16(const void* base, unsigned byte_off) {
  return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ void stg16(void* base, unsigned byte_off, const f32x4& v) {
  *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(base) + byte_off) = v;
}
// Agent-scope ("sc1": past the vector L1, write-through) accesses -- what a hand-off between workgroups INSIDE one launch is built
// from (k_epoch64, kernels_epoch64.h).  COH = false is the plain access every other kernel uses: a kernel boundary orders those.
// All of them are relaxed atomics of 4 or 8 bytes (compiler-tracked waits, no inline asm); a 16-byte access is two 8-byte ones
// (MI355X_MICROARCH.md, inter-workgroup visibility: stores and loads of the handed-off bytes all sc1, 4- / 8- / 16-byte forms).
template <bool COH, class T>
__device__ __forceinline__ T ldc(const T* p) {
  if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else return *p;
}
template <bool COH, class T>
__device__ __forceinline__ void stc(T* p, T v) {
  if constexpr (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}
template <bool COH>
__device__ __forceinline__ f32x4 ldg16c(const void* base, unsigned byte_off) {
  if constexpr (!COH) return ldg16(base, byte_off);
  else {
    const unsigned long long* p = reinterpret_cast<const unsigned long long*>(reinterpret_cast<const char*>(base) + byte_off);
    const unsigned long long a = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long b = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return f32x4{__uint_as_float((unsigned)a), __uint_as_float((unsigned)(a >> 32)), __uint_as_float((unsigned)b), __uint_as_float((unsigned)(b >> 32))};
  }
}
template <bool COH>
__device__ __forceinline__ void stg16c(void* base, unsigned byte_off, const f32x4& v) {
  if constexpr (!COH) stg16(base, byte_off, v);
  else {
    unsigned long long* p = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(base) + byte_off);
    __hip_atomic_store(p, (unsigned long long)__float_as_uint(v[0]) | ((unsigned long long)__float_as_uint(v[1]) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p + 1, (unsigned long long)__float_as_uint(v[2]) | ((unsigned long long)__float_as_uint(v[3]) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// C layout: element i of a 32x32 accumulator sits at row crc(i) + 4*h, column r of the tile
__host__ __device__ __forceinline__ constexpr int crc(int i) { return (i & 3) + 8 * (i >> 2); }

// acc[cb][rb] += A[rb*32 + 0..31][0..8*nkg) . Bpacked[cb]  for this wave's two 32-column blocks.
// a_off: LDS offset of the A tile (row stride lda floats); Bp0/Bp1: packed fragments [nkg][64] of the two blocks.
// B fragments are prefetched two k-groups ahead (global/L2 latency); A fragments one k-group ahead (LDS).
#define MFMA_KG(u, v, p, q)                     \
  _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) { \
    c00 = MFMA32(u[s_], p[s_], c00);            \
    c01 = MFMA32(v[s_], p[s_], c01);            \
    c10 = MFMA32(u[s_], q[s_], c10);            \
    c11 = MFMA32(v[s_], q[s_], c11);            \
  }
// nkg must be even.  The k-group loop is unrolled by two with ping-pong operand buffers (no register rotation:
// a rotation makes the compiler copy -- and therefore wait for -- loads that were only just issued).
// first k-group's weight fragments of a GEMM phase, fetched one phase early (hides the L2 latency that a
// single wave per SIMD cannot cover otherwise)
struct Frag2 {
  f32x4 p, q;
};
__device__ __forceinline__ Frag2 prefetch_frag(const f32x4* __restrict__ Bp0, const f32x4* __restrict__ Bp1, int lane) {
  const unsigned bo = opaque_u((unsigned)lane * 16u);
  return Frag2{ldg16(Bp0, bo), ldg16(Bp1, bo)};
}
template <int LDA>
__device__ __forceinline__ void gemm_lds_packed(int a_off, const f32x4* __restrict__ Bp0,
                                                const f32x4* __restrict__ Bp1, int nkg, f32x16& c00, f32x16& c01,
                                                f32x16& c10, f32x16& c11, int lane, const Frag2& first) {
  const int r = lane & 31, h = lane >> 5;
  const int ab = 4 * opaque((a_off + r * LDA + 4 * h) >> 2);  // provably 16-byte aligned -> ds_read_b128
  unsigned bo = opaque_u((unsigned)lane * 16u);  // byte offset of this lane's fragment; +1024 per k-group
  f32x4 pA = first.p, qA = first.q, pB, qB;
  f32x4 uA = *reinterpret_cast<const f32x4*>(&lds[ab]);
  f32x4 vA = *reinterpret_cast<const f32x4*>(&lds[ab + 32 * LDA]);
  f32x4 uB, vB;
  int ao = ab;
  // branch-free steady state (exact counted waits), last pair peeled
#pragma unroll 1
  for (int kg = 0; kg < nkg - 2; kg += 2) {
    pB = ldg16(Bp0, bo + 1024u);
    qB = ldg16(Bp1, bo + 1024u);
    uB = *reinterpret_cast<const f32x4*>(&lds[ao + 8]);
    vB = *reinterpret_cast<const f32x4*>(&lds[ao + 32 * LDA + 8]);
    MFMA_KG(uA, vA, pA, qA)
    pA = ldg16(Bp0, bo + 2048u);
    qA = ldg16(Bp1, bo + 2048u);
    uA = *reinterpret_cast<const f32x4*>(&lds[ao + 16]);
    vA = *reinterpret_cast<const f32x4*>(&lds[ao + 32 * LDA + 16]);
    MFMA_KG(uB, vB, pB, qB)
    bo += 2048u;
    ao += 16;
  }
  pB = ldg16(Bp0, bo + 1024u);
  qB = ldg16(Bp1, bo + 1024u);
  uB = *reinterpret_cast<const f32x4*>(&lds[ao + 8]);
  vB = *reinterpret_cast<const f32x4*>(&lds[ao + 32 * LDA + 8]);
  MFMA_KG(uA, vA, pA, qA)
  MFMA_KG(uB, vB, pB, qB)
}

// Deep-pipelined variant for the K = 256 contractions (nkg % 4 == 0, nkg >= 8): weight fragments are fetched THREE
// k-groups (~3 us of MFMA work) ahead through a 4-deep register ring -- one k-group of lookahead does not cover
// the L2 latency when all 256 CUs stream weights -- A fragments one k-group ahead from LDS.
template <int LDA>
__device__ __forceinline__ void gemm_lds_packed_deep(int a_off, const f32x4* __restrict__ Bp0,
                                                     const f32x4* __restrict__ Bp1, int nkg, f32x16& c00,
                                                     f32x16& c01, f32x16& c10, f32x16& c11, int lane,
                                                     const Frag2& first) {
  const int r = lane & 31, h = lane >> 5;
  const int ab = 4 * opaque((a_off + r * LDA + 4 * h) >> 2);
  unsigned bo = opaque_u((unsigned)lane * 16u);
  f32x4 p0 = first.p, q0 = first.q;
  f32x4 p1 = ldg16(Bp0, bo + 1024u), q1 = ldg16(Bp1, bo + 1024u);
  f32x4 p2 = ldg16(Bp0, bo + 2048u), q2 = ldg16(Bp1, bo + 2048u);
  f32x4 p3, q3;
  f32x4 uA = *reinterpret_cast<const f32x4*>(&lds[ab]);
  f32x4 vA = *reinterpret_cast<const f32x4*>(&lds[ab + 32 * LDA]);
  f32x4 uB, vB;
  int ao = ab;
#define LDA_NEXT(U, V, off)                                             \
  U = *reinterpret_cast<const f32x4*>(&lds[ao + (off)]);                \
  V = *reinterpret_cast<const f32x4*>(&lds[ao + 32 * LDA + (off)]);
#pragma unroll 1
  for (int kg = 0; kg < nkg - 4; kg += 4) {
    p3 = ldg16(Bp0, bo + 3072u); q3 = ldg16(Bp1, bo + 3072u);
    LDA_NEXT(uB, vB, 8)
    MFMA_KG(uA, vA, p0, q0)
    p0 = ldg16(Bp0, bo + 4096u); q0 = ldg16(Bp1, bo + 4096u);
    LDA_NEXT(uA, vA, 16)
    MFMA_KG(uB, vB, p1, q1)
    p1 = ldg16(Bp0, bo + 5120u); q1 = ldg16(Bp1, bo + 5120u);
    LDA_NEXT(uB, vB, 24)
    MFMA_KG(uA, vA, p2, q2)
    p2 = ldg16(Bp0, bo + 6144u); q2 = ldg16(Bp1, bo + 6144u);
    LDA_NEXT(uA, vA, 32)
    MFMA_KG(uB, vB, p3, q3)
    bo += 4096u;
    ao += 32;
  }
  p3 = ldg16(Bp0, bo + 3072u); q3 = ldg16(Bp1, bo + 3072u);
  LDA_NEXT(uB, vB, 8)
  MFMA_KG(uA, vA, p0, q0)
  LDA_NEXT(uA, vA, 16)
  MFMA_KG(uB, vB, p1, q1)
  LDA_NEXT(uB, vB, 24)
  MFMA_KG(uA, vA, p2, q2)
  MFMA_KG(uB, vB, p3, q3)
#undef LDA_NEXT
}

// ------------------------------------------------------------------------------------------------
// dW2 accumulators: 16 tiles of 32x32 (this wave's 64 neurons x 256 inputs) pinned to the accumulator
// register file for the whole kernel.  The library is compiled with -mllvm -amdgpu-mfma-vgpr-form, so every
// compiler-selected MFMA keeps its accumulator in arch VGPRs (work tiles, <= 64 registers); the persistent
// tiles are only ever touched by the statement below, whose "+a" constraints make the register allocator keep
// all 256 AGPRs occupied by them (so it cannot park anything else there).  Tile t = ib*8 + jb.
// ------------------------------------------------------------------------------------------------
#define MFA(t, x, y) "v_mfma_f32_32x32x2_f32 %" #t ", %" #x ", %" #y ", %" #t "\n\t"
// one k-step (two batch rows) of dW2 += dz2^T . h1: 16 MFMAs; x = A operands (2 neuron blocks), y = B operands
__device__ __forceinline__ void dw2_kstep(f32x16 (&g)[16], float x0, float x1, float y0, float y1, float y2,
                                          float y3, float y4, float y5, float y6, float y7) {
  asm volatile(
      "s_nop 1\n\t"  // VALU-written operand -> MFMA
      MFA(0, 16, 18) MFA(8, 17, 18) MFA(1, 16, 19) MFA(9, 17, 19) MFA(2, 16, 20) MFA(10, 17, 20)
      MFA(3, 16, 21) MFA(11, 17, 21) MFA(4, 16, 22) MFA(12, 17, 22) MFA(5, 16, 23) MFA(13, 17, 23)
      MFA(6, 16, 24) MFA(14, 17, 24) MFA(7, 16, 25) MFA(15, 17, 25)
      "s_nop 1"
      : "+a"(g[0]), "+a"(g[1]), "+a"(g[2]), "+a"(g[3]), "+a"(g[4]), "+a"(g[5]), "+a"(g[6]), "+a"(g[7]), "+a"(g[8]),
        "+a"(g[9]), "+a"(g[10]), "+a"(g[11]), "+a"(g[12]), "+a"(g[13]), "+a"(g[14]), "+a"(g[15])
      : "v"(x0), "v"(x1), "v"(y0), "v"(y1), "v"(y2), "v"(y3), "v"(y4), "v"(y5), "v"(y6), "v"(y7));
}

// Small TN phases (dW3, dW1) keep their accumulators in arch VGPRs; the MFMAs of one k-step are issued as one
// opaque statement so that the compiler cannot split the interleaved chains into separate passes (it did, spilling
// every second operand).  "s_nop 1": VALU-written operand -> MFMA.  The accumulators are only read at kernel end.
__device__ __forceinline__ void mfma_x1y2(f32x16& c0, f32x16& c1, float x, float y0, float y1) {
  asm volatile("s_nop 1\n\t"
               "v_mfma_f32_32x32x2_f32 %0, %2, %3, %0\n\t"
               "v_mfma_f32_32x32x2_f32 %1, %2, %4, %1"
               : "+v"(c0), "+v"(c1)
               : "v"(x), "v"(y0), "v"(y1));
}
__device__ __forceinline__ void mfma_x2y1(f32x16& c0, f32x16& c1, float x0, float x1, float y) {
  asm volatile("s_nop 1\n\t"
               "v_mfma_f32_32x32x2_f32 %0, %2, %4, %0\n\t"
               "v_mfma_f32_32x32x2_f32 %1, %3, %4, %1"
               : "+v"(c0), "+v"(c1)
               : "v"(x0), "v"(x1), "v"(y));
}
__device__ __forceinline__ void mfma_x2y2(f32x16& c00, f32x16& c10, f32x16& c01, f32x16& c11, float x0, float x1,
                                          float y0, float y1) {
  asm volatile("s_nop 1\n\t"
               "v_mfma_f32_32x32x2_f32 %0, %4, %6, %0\n\t"
               "v_mfma_f32_32x32x2_f32 %1, %5, %6, %1\n\t"
               "v_mfma_f32_32x32x2_f32 %2, %4, %7, %2\n\t"
               "v_mfma_f32_32x32x2_f32 %3, %5, %7, %3"
               : "+v"(c00), "+v"(c10), "+v"(c01), "+v"(c11)
               : "v"(x0), "v"(x1), "v"(y0), "v"(y1));
}

// v_mfma_f32_16x16x4_f32 (A: lane l -> A[l&15][l>>4], B: lane l -> B[l>>4][l&15], C: reg e -> C[4(l>>4)+e][l&15]):
// same MAC rate as 32x32x2 but N = 16, so a head of <= 16 outputs costs half the cycles of a 32-wide tile.
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)
__device__ __forceinline__ void mfma16_x1y4(f32x4& c0, f32x4& c1, f32x4& c2, f32x4& c3, float x, float y0, float y1,
                                            float y2, float y3) {
  asm volatile("s_nop 1\n\t"
               "v_mfma_f32_16x16x4_f32 %0, %4, %5, %0\n\t"
               "v_mfma_f32_16x16x4_f32 %1, %4, %6, %1\n\t"
               "v_mfma_f32_16x16x4_f32 %2, %4, %7, %2\n\t"
               "v_mfma_f32_16x16x4_f32 %3, %4, %8, %3"
               : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)
               : "v"(x), "v"(y0), "v"(y1), "v"(y2), "v"(y3));
}

// ------------------------------------------------------------------------------------------------
// epilogue helpers for the wave's 64x64 output block (2 column blocks x 2 row blocks, C layout)
// ------------------------------------------------------------------------------------------------
// lds[dst][row][col] = tanh(acc / kTanhScale)   (acc already holds kTanhScale * (x.W^T + b))
__device__ __forceinline__ void store_tanh(int dst_off, int wave, int lane, const f32x16& c00, const f32x16& c01,
                                           const f32x16& c10, const f32x16& c11) {
  const int r = lane & 31, h = lane >> 5;
  const int o = opaque(dst_off + 4 * h * FLDH + 64 * wave + r);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    lds[o + crc(i) * FLDH] = fast_tanh_scaled(c00[i]);
    lds[o + (32 + crc(i)) * FLDH] = fast_tanh_scaled(c01[i]);
    lds[o + crc(i) * FLDH + 32] = fast_tanh_scaled(c10[i]);
    lds[o + (32 + crc(i)) * FLDH + 32] = fast_tanh_scaled(c11[i]);
  }
}
// lds[hs][row][col] <- acc * (1 - hs^2) in place
__device__ __forceinline__ void dtanh_inplace(int hs_off, int wave, int lane, const f32x16& c00, const f32x16& c01,
                                              const f32x16& c10, const f32x16& c11) {
  const int r = lane & 31, h = lane >> 5;
  const int o = opaque(hs_off + 4 * h * FLDH + 64 * wave + r);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    float hv;
    hv = lds[o + crc(i) * FLDH];             lds[o + crc(i) * FLDH] = c00[i] * (1.0f - hv * hv);
    hv = lds[o + (32 + crc(i)) * FLDH];      lds[o + (32 + crc(i)) * FLDH] = c01[i] * (1.0f - hv * hv);
    hv = lds[o + crc(i) * FLDH + 32];        lds[o + crc(i) * FLDH + 32] = c10[i] * (1.0f - hv * hv);
    hv = lds[o + (32 + crc(i)) * FLDH + 32]; lds[o + (32 + crc(i)) * FLDH + 32] = c11[i] * (1.0f - hv * hv);
  }
}
// sum of column `tid` of a [64][FLDH] LDS tile over its 64 rows, fixed order (bias gradients)
__device__ __forceinline__ float column_sum(int hs_off, int tid) {
  const int o = opaque(hs_off + tid);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll 2
  for (int rr = 0; rr < FR; rr += 4) {
    s0 += lds[o + rr * FLDH];
    s1 += lds[o + (rr + 1) * FLDH];
    s2 += lds[o + (rr + 2) * FLDH];
    s3 += lds[o + (rr + 3) * FLDH];
  }
  return (s0 + s1) + (s2 + s3);
}

// LDS carve-up (float offsets) for a given padded observation width
template <int DP>
struct Lay {
  static constexpr int LDX = DP + 4;
  static constexpr int X = 0;
  static constexpr int H1 = X + FR * LDX;
  static constexpr int H2 = H1 + FR * FLDH;
  static constexpr int DO = H2 + FR * FLDH;
  static constexpr int CST = DO + FR * FLDO;  // [3][32]: 1/var, log(sd)+log(sqrt(2pi)), head bias
  static constexpr int GACC = CST + 96;       // [4 waves][2][32]: per-wave head-bias / log_std gradient sums
  static constexpr int STAT = GACC + 256;     // [4 waves][4]: per-wave loss sums (kernel end)
  static constexpr int END = STAT + 16;
};

// ------------------------------------------------------------------------------------------------
// Diagnostic build (-DMOBROB_STAMPS): s_memtime stamps at phase boundaries, summed over waves into
// a.stamps[phase] (cdna_hip_programming.md §7 'In-kernel stamps').  Never compiled into the product library.
// ------------------------------------------------------------------------------------------------
// Timing-only ablation builds (-DMOBROB_SKIP=<mask>): skip one phase of k_fused_train to price it in situ
// (outputs are wrong by construction; never compiled into the product library).
#ifndef MOBROB_SKIP
#define MOBROB_SKIP 0
#endif
#define PHASE_ON(bit) (!((MOBROB_SKIP) & (bit)))
#ifndef MOBROB_DW3_INSIDE_DH2   // A/B switch: 0 = dW3 and dh2 as two phases (rounds 1-2)
#define MOBROB_DW3_INSIDE_DH2 1
#endif
#ifdef MOBROB_STAMPS
// The deltas are accumulated in (scalar) registers and flushed once at the end of the kernel: a global atomic per
// stamp would sit in the in-order vmcnt queue in front of the next phase's weight-fragment loads and charge its own
// latency to that phase.
#define STAMP(id)                                                                                  \
  {                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    unsigned long long t_;                                                                         \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                     \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    tacc_[id] += t_ - tprev_;                                                                      \
    tprev_ = t_;                                                                                   \
  }
#define STAMP_INIT()                                                                               \
  unsigned long long tprev_, tclk0_, treal0_;                                                      \
  unsigned long long tacc_[26];                                                                    \
  _Pragma("unroll") for (int i_ = 0; i_ < 26; ++i_) tacc_[i_] = 0;                                 \
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(treal0_)::"memory");              \
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev_)::"memory");                   \
  tclk0_ = tprev_;
// slots 24 / 25: the wave's shader-clock cycles and 100 MHz ticks between STAMP_INIT and STAMP_FLUSH -> the in-kernel clock
// (MI355X_MICROARCH.md, DVFS give-back item 6)
#define STAMP_FLUSH()                                                                              \
  {                                                                                                \
    unsigned long long t1_, r1_;                                                                   \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1_)::"memory");                    \
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1_)::"memory");                \
    tacc_[24] = t1_ - tclk0_; tacc_[25] = r1_ - treal0_;                                           \
  }                                                                                                \
  if ((threadIdx.x & 63) == 0) {                                                                   \
    _Pragma("unroll") for (int i_ = 0; i_ < 26; ++i_) atomicAdd(&stamps_[i_], tacc_[i_]);          \
  }
#define STAMP_PARAMS , unsigned long long (&tacc_)[26], unsigned long long &tprev_
#define STAMP_ARGS , tacc_, tprev_
#else
#define STAMP(id)
#define STAMP_INIT()
#define STAMP_FLUSH()
#define STAMP_PARAMS
#define STAMP_ARGS
#endif

// ------------------------------------------------------------------------------------------------
// forward of one 64-row tile through one network; leaves h1, h2 in LDS and the raw head tile
// (without bias) in the head tile [64][FLDO].  All 4 waves participate; ends with a barrier.
// ------------------------------------------------------------------------------------------------
template <int DP, bool H16>
__device__ __forceinline__ Frag2 tile_layers(const FusedNet& W, int wave, int lane, const Frag2& f1 STAMP_PARAMS) {
  using L = Lay<DP>;
  constexpr int nkg2 = FH / 8;
  const f32x4* w2a = W.W2f + (size_t)(2 * wave) * nkg2 * 64;
  const f32x4* w2b = W.W2f + (size_t)(2 * wave + 1) * nkg2 * 64;
  Frag2 f2;
  const int r_ = lane & 31;
  {  // layer 1: K = DP
    const float bz0 = W.b1s[64 * wave + r_], bz1 = W.b1s[64 * wave + 32 + r_];
    f32x16 c00 = splat16(bz0), c01 = splat16(bz0), c10 = splat16(bz1), c11 = splat16(bz1);
    constexpr int nkg = DP / 8;
    if (PHASE_ON(2))
      gemm_lds_packed<L::LDX>(L::X, W.W1f + (size_t)(2 * wave) * nkg * 64, W.W1f + (size_t)(2 * wave + 1) * nkg * 64,
                              nkg, c00, c01, c10, c11, lane, f1);
    f2 = prefetch_frag(w2a, w2b, lane);
    STAMP(1)
    if (PHASE_ON(1024)) store_tanh(L::H1, wave, lane, c00, c01, c10, c11);
    else asm volatile("" ::"v"(c00), "v"(c01), "v"(c10), "v"(c11));
    STAMP(2)
  }
  __syncthreads();
  STAMP(3)
  // head GEMM operands: K split in two halves; wave = (khalf << 1) | rowblock
  const f32x4* bp = H16 ? W.W3h : W.W3f + (size_t)((wave >> 1) * 16) * 64;
  Frag2 f3;
  {  // layer 2: K = H
    const float bz0 = W.b2s[64 * wave + r_], bz1 = W.b2s[64 * wave + 32 + r_];
    f32x16 c00 = splat16(bz0), c01 = splat16(bz0), c10 = splat16(bz1), c11 = splat16(bz1);
    if (PHASE_ON(4)) gemm_lds_packed_deep<FLDH>(L::H1, w2a, w2b, nkg2, c00, c01, c10, c11, lane, f2);
    f3 = prefetch_frag(bp, bp + 64, lane);
    STAMP(4)
    if (PHASE_ON(1024)) store_tanh(L::H2, wave, lane, c00, c01, c10, c11);
    else asm volatile("" ::"v"(c00), "v"(c01), "v"(c10), "v"(c11));
    STAMP(5)
  }
  __syncthreads();
  STAMP(6)
  return f3;
}
template <int DP>
__device__ __forceinline__ void tile_head(const FusedNet& W, int wave, int lane, const Frag2& f3 STAMP_PARAMS) {
  using L = Lay<DP>;
  const int r = lane & 31, h = lane >> 5;
  const int rb = wave & 1, ks = wave >> 1;
  const f32x4* bp = W.W3f + (size_t)(ks * 16) * 64;
  {  // head: [64 x 32] = h2 . W3^T
    f32x16 acc = zero16(), acc2 = zero16();  // two independent chains (even / odd k-groups)
    const int ab = 4 * opaque((L::H2 + (rb * 32 + r) * FLDH + ks * 128 + 4 * h) >> 2);
    unsigned bo = opaque_u((unsigned)lane * 16u);
    f32x4 bA = f3.p, bB = f3.q;
    f32x4 aA = *reinterpret_cast<const f32x4*>(&lds[ab]), aB = *reinterpret_cast<const f32x4*>(&lds[ab + 8]);
    int ao = ab;
#pragma unroll 1
    for (int kg = 0; kg < 14; kg += 2) {
      const f32x4 b0 = bA, b1 = bB, a0 = aA, a1 = aB;
      bA = ldg16(bp, bo + 2048u);
      bB = ldg16(bp, bo + 3072u);
      aA = *reinterpret_cast<const f32x4*>(&lds[ao + 16]);
      aB = *reinterpret_cast<const f32x4*>(&lds[ao + 24]);
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) {
        acc = MFMA32(a0[s_], b0[s_], acc);
        acc2 = MFMA32(a1[s_], b1[s_], acc2);
      }
      bo += 2048u;
      ao += 16;
    }
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) {
      acc = MFMA32(aA[s_], bA[s_], acc);
      acc2 = MFMA32(aB[s_], bB[s_], acc2);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] += acc2[i];
    STAMP(7)
    // deterministic cross-wave reduction through the head tile: one K-half per round
    const int o = opaque(L::DO + (rb * 32 + 4 * h) * FLDO + r);
#pragma unroll
    for (int round = 0; round < 2; ++round) {
      if (ks == round) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          if (round == 0) lds[o + crc(i) * FLDO] = acc[i];
          else lds[o + crc(i) * FLDO] += acc[i];
        }
      }
      __syncthreads();
    }
  }
}

// Heads of <= 16 outputs: wave w computes rows [16w, 16w+16) of the head tile over the full K = 256 with
// v_mfma_f32_16x16x4_f32 (half the MFMA cycles of the zero-padded 32-wide tile).  No cross-wave reduction, and the
// loss stage of wave w (4 lanes per row, rows 16w..16w+15) consumes exactly the rows this wave wrote -> no barrier.
template <int DP>
__device__ __forceinline__ void tile_head16(const FusedNet& W, int wave, int lane, const Frag2& f3) {
  using L = Lay<DP>;
  const int i = lane & 15, g = lane >> 4;
  const f32x4* bp = W.W3h;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};  // two independent chains (even / odd k-groups)
  const int ab = 4 * opaque((L::H2 + (16 * wave + i) * FLDH + 4 * g) >> 2);
  unsigned bo = opaque_u((unsigned)lane * 16u);
  f32x4 bA = f3.p, bB = f3.q;
  f32x4 aA = *reinterpret_cast<const f32x4*>(&lds[ab]), aB = *reinterpret_cast<const f32x4*>(&lds[ab + 16]);
  int ao = ab;
#pragma unroll 1
  for (int kg = 0; kg < FH / 16 - 2; kg += 2) {
    const f32x4 b0 = bA, b1 = bB, a0 = aA, a1 = aB;
    bA = ldg16(bp, bo + 2048u);
    bB = ldg16(bp, bo + 3072u);
    aA = *reinterpret_cast<const f32x4*>(&lds[ao + 32]);
    aB = *reinterpret_cast<const f32x4*>(&lds[ao + 48]);
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) {
      acc = MFMA16(a0[s_], b0[s_], acc);
      acc2 = MFMA16(a1[s_], b1[s_], acc2);
    }
    bo += 2048u;
    ao += 32;
  }
#pragma unroll
  for (int s_ = 0; s_ < 4; ++s_) {
    acc = MFMA16(aA[s_], bA[s_], acc);
    acc2 = MFMA16(aB[s_], bB[s_], acc2);
  }
  const int o = opaque(L::DO + (16 * wave + 4 * g) * FLDO + i);
#pragma unroll
  for (int e = 0; e < 4; ++e) lds[o + e * FLDO] = acc[e] + acc2[e];
}

// ------------------------------------------------------------------------------------------------
// "x3" GEMMs: a float32 product on the bf16 matrix pipe.  Every float32 operand is split into three bf16 pieces
// (x = x1 + x2 + x3, each the bf16 rounding of what the previous ones left: 24 mantissa bits, the residuals are exact in
// float32) and six of the nine piece products are kept (x1y1, x1y2, x2y1, x1y3, x3y1, x2y2; the dropped ones are below
// 2^-24 of the product), accumulated in float32 by v_mfma_f32_32x32x16_bf16.  The piece products are exact in float32, so a
// 16-k step rounds once where eight v_mfma_f32_32x32x2_f32 round eight times: measured against float64 the error is SMALLER
// than the f32 instruction's (scratch/bf16x3_probe.hip: 3.7e-7 against 5.2e-7 of max |C|) at 1.75-1.87x its rate.
// Used by the forward-only kernels (rollout policy forward, batched value pass); the gradient kernels stay on the f32 pipe.
//
// Weight packs (k_pack_x3_multi, kept current by k_adam_pack): [column block n/32][k step k/16][piece 3][lane 64] x 8 bf16 (16 bytes per lane):
//   lane = (n & 31) + 32 ((k & 15) >> 3), element j = k & 7      (B[k][n] = scale * W[n][k]: the 32x32x16 B operand)
// ------------------------------------------------------------------------------------------------
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MFMA32B(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), (c), 0, 0, 0)

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {  // round to nearest even, lo in the low half
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
// The same instruction selected by the compiler from a plain conversion.  Measured (round 4, profiles/r4/chain_variants.txt): in the
// MFMA-dense weight-gradient loops the asm form is the faster one -- with the conversion visible the compiler recomputes single
// conversions for the residuals and SLP-packs the subtractions into v_pk_add_f32, which is slow beside MFMAs (dW2 16.3 k -> 21.4 k
// cycles per tile) --, in the side work of the chain loops (few instructions between v_mfma_f32_16x16x32_bf16) the native form (dh1 25.0 k -> 23.3 k).
__device__ __forceinline__ unsigned cvt_pk_bf16_native(float lo, float hi) {
  const bf16x2_t v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ void x3_split2(float a, float b, unsigned& p1, unsigned& p2, unsigned& p3) {
  p1 = cvt_pk_bf16(a, b);
  const float ra = a - __uint_as_float(p1 << 16), rb = b - __uint_as_float(p1 & 0xffff0000u);   // exact
  p2 = cvt_pk_bf16(ra, rb);
  const float sa = ra - __uint_as_float(p2 << 16), sb = rb - __uint_as_float(p2 & 0xffff0000u);  // exact
  p3 = cvt_pk_bf16(sa, sb);
}
struct X3Frag { u32x4 p[3]; };  // eight consecutive k of one row (or column), three pieces
__device__ __forceinline__ X3Frag x3_split8(const f32x4& a, const f32x4& b) {
  X3Frag f;
  unsigned p1, p2, p3;
  x3_split2(a[0], a[1], p1, p2, p3); f.p[0][0] = p1; f.p[1][0] = p2; f.p[2][0] = p3;
  x3_split2(a[2], a[3], p1, p2, p3); f.p[0][1] = p1; f.p[1][1] = p2; f.p[2][1] = p3;
  x3_split2(b[0], b[1], p1, p2, p3); f.p[0][2] = p1; f.p[1][2] = p2; f.p[2][2] = p3;
  x3_split2(b[2], b[3], p1, p2, p3); f.p[0][3] = p1; f.p[1][3] = p2; f.p[2][3] = p3;
  return f;
}
__device__ __forceinline__ X3Frag x3_load_b(const u32x4* __restrict__ Bx, int ks, int lane) {
  X3Frag f;
#pragma unroll
  for (int pc = 0; pc < 3; ++pc) f.p[pc] = Bx[(size_t)(ks * 3 + pc) * 64 + lane];
  return f;
}
// the six kept products, small terms first
#define X3_MFMA6(A_, B_, C_)                    \
  C_ = MFMA32B(A_.p[1], B_.p[1], C_);           \
  C_ = MFMA32B(A_.p[0], B_.p[2], C_);           \
  C_ = MFMA32B(A_.p[2], B_.p[0], C_);           \
  C_ = MFMA32B(A_.p[0], B_.p[1], C_);           \
  C_ = MFMA32B(A_.p[1], B_.p[0], C_);           \
  C_ = MFMA32B(A_.p[0], B_.p[0], C_);
// c0 / c1 += A[32 rows][16 NKS] (float32 in LDS, row stride LDA) . B of two column blocks (x3 packs of NKS k steps each).
// Weight fragments are requested kAhead k steps before their use (a k step is 12 MFMAs = 384 cycles; L2 takes 500-800).
#ifndef MOBROB_X3_AHEAD
#define MOBROB_X3_AHEAD 2
#endif
#ifndef MOBROB_X3_PIPE
#define MOBROB_X3_PIPE 1
#endif
// Software pipeline of a k step: the MFMAs of step ks (operands split during step ks - 1) are issued with the ~50 VALU
// instructions that split the A fragment of step ks + 1 between them (an MFMA occupies the matrix pipe for 32 cycles, four
// VALU instructions fill them); without the explicit groups the compiler emitted the split as one block in front of a burst of
// MFMAs, each side waiting for the other.
template <int LDA, int NKS>
__device__ __forceinline__ void gemm_x3_r32(int a_off, const u32x4* __restrict__ Bx0, const u32x4* __restrict__ Bx1, f32x16& c0,
                                            f32x16& c1, int lane) {
  constexpr int kAhead = NKS < MOBROB_X3_AHEAD ? NKS : MOBROB_X3_AHEAD;
  const int r = lane & 31, h = lane >> 5;
  const int ab = 4 * opaque((a_off + r * LDA + 8 * h) >> 2);
  X3Frag P[kAhead + 1], Q[kAhead + 1];
#pragma unroll
  for (int k = 0; k < kAhead; ++k) { P[k] = x3_load_b(Bx0, k, lane); Q[k] = x3_load_b(Bx1, k, lane); }
  X3Frag U = x3_split8(*reinterpret_cast<const f32x4*>(&lds[ab]), *reinterpret_cast<const f32x4*>(&lds[ab + 4]));
  f32x4 ua = {0.f, 0.f, 0.f, 0.f}, ub = ua;  // float32 A fragment of step ks + 1, read one step before it is split
  if (NKS > 1) {
    ua = *reinterpret_cast<const f32x4*>(&lds[ab + 16]);
    ub = *reinterpret_cast<const f32x4*>(&lds[ab + 20]);
  }
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
#if MOBROB_X3_PIPE
    __builtin_amdgcn_sched_barrier(0);
#endif
    f32x4 na = ua, nb = ub;
    if (ks + 2 < NKS) {
      na = *reinterpret_cast<const f32x4*>(&lds[ab + 16 * (ks + 2)]);
      nb = *reinterpret_cast<const f32x4*>(&lds[ab + 16 * (ks + 2) + 4]);
    }
    if (ks + kAhead < NKS) {
      P[(ks + kAhead) % (kAhead + 1)] = x3_load_b(Bx0, ks + kAhead, lane);
      Q[(ks + kAhead) % (kAhead + 1)] = x3_load_b(Bx1, ks + kAhead, lane);
    }
    const X3Frag& p = P[ks % (kAhead + 1)];
    const X3Frag& q = Q[ks % (kAhead + 1)];
    X3_MFMA6(U, p, c0)
    X3_MFMA6(U, q, c1)
    if (ks + 1 < NKS) U = x3_split8(ua, ub);
    ua = na; ub = nb;
#if MOBROB_X3_PIPE
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // the two LDS reads of the A fragment two steps ahead
    __builtin_amdgcn_sched_group_barrier(0x020, 6, 0);  // the six weight-fragment loads kAhead steps ahead
#pragma unroll
    for (int g = 0; g < 12; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);  // four VALU
    }
#endif
  }
}
// The same arithmetic (same k order, same six products per step: the same bits) with a bounded register footprint for the
// training kernel: a runtime k loop over PAIRS of steps, weight fragments one step ahead in two fixed slots, the A fragment read
// and split one step ahead (c 32 + U 12 + float32 fragment 8 + two slots 48 registers).
// one product of the step for both column blocks: c0 += U[ia] . P[ib], c1 += U[ia] . Q[ib]  (accumulators in arch VGPRs)
#define X3R32_MFMA(Uf, Pf, Qf, ia, ib)                                                  \
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\t"                            \
               "v_mfma_f32_32x32x16_bf16 %1, %2, %4, %1"                                \
               : "+v"(c0), "+v"(c1)                                                     \
               : "v"(Uf.p[ia]), "v"(Pf.p[ib]), "v"(Qf.p[ib]))
#define X3_SPLIT_PAIR(dst, jp, lo, hi)                                                  \
  {                                                                                     \
    unsigned p1_, p2_, p3_;                                                             \
    x3_split2(lo, hi, p1_, p2_, p3_);                                                   \
    dst.p[0][jp] = p1_; dst.p[1][jp] = p2_; dst.p[2][jp] = p3_;                         \
  }
// a step: the twelve MFMAs in six statements, the four pair-splits of the NEXT A fragment between the first five (an MFMA
// holds the matrix pipe for 32 cycles; a pair-split is 11 VALU instructions) -- written out by hand: left to the scheduler
// (sched_group_barrier), the first step of every call came out as a burst of MFMAs followed by a block of splits
#define X3R32_STEP(Pf, Qf)                                   \
  {                                                          \
    X3Frag Un_;                                              \
    asm volatile("s_nop 1");                                 \
    X3R32_MFMA(U, Pf, Qf, 1, 1);                             \
    X3_SPLIT_PAIR(Un_, 0, ua[0], ua[1])                      \
    X3R32_MFMA(U, Pf, Qf, 0, 2);                             \
    X3_SPLIT_PAIR(Un_, 1, ua[2], ua[3])                      \
    X3R32_MFMA(U, Pf, Qf, 2, 0);                             \
    X3_SPLIT_PAIR(Un_, 2, ub[0], ub[1])                      \
    X3R32_MFMA(U, Pf, Qf, 0, 1);                             \
    X3_SPLIT_PAIR(Un_, 3, ub[2], ub[3])                      \
    X3R32_MFMA(U, Pf, Qf, 1, 0);                             \
    X3R32_MFMA(U, Pf, Qf, 0, 0);                             \
    U = Un_;                                                 \
  }
template <int LDA, int NKS>
__device__ __forceinline__ void gemm_x3_r32_lean(int a_off, const u32x4* __restrict__ Bx0, const u32x4* __restrict__ Bx1, f32x16& c0,
                                                 f32x16& c1, int lane) {
  const int r = lane & 31, h = lane >> 5;
  int ao = 4 * opaque((a_off + r * LDA + 8 * h) >> 2);
  const u32x4* b0 = Bx0 + lane;
  const u32x4* b1 = Bx1 + lane;
  X3Frag P0, Q0, P1, Q1;
#pragma unroll
  for (int pc = 0; pc < 3; ++pc) { P0.p[pc] = b0[pc * 64]; Q0.p[pc] = b1[pc * 64]; }
  X3Frag U = x3_split8(*reinterpret_cast<const f32x4*>(&lds[ao]), *reinterpret_cast<const f32x4*>(&lds[ao + 4]));
#pragma unroll 1
  for (int ks = 0; ks + 2 <= NKS; ks += 2) {
    {  // step ks from slot 0; step ks + 1 exists
      const f32x4 ua = *reinterpret_cast<const f32x4*>(&lds[ao + 16]);
      const f32x4 ub = *reinterpret_cast<const f32x4*>(&lds[ao + 20]);
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) { P1.p[pc] = b0[(3 + pc) * 64]; Q1.p[pc] = b1[(3 + pc) * 64]; }
      X3R32_STEP(P0, Q0)
    }
    {  // step ks + 1 from slot 1; step ks + 2 may not exist: its (unused) operands are then those of the last step again
      const int adv = ks + 2 < NKS ? 1 : 0;
      const f32x4 ua = *reinterpret_cast<const f32x4*>(&lds[ao + 16 + 16 * adv]);
      const f32x4 ub = *reinterpret_cast<const f32x4*>(&lds[ao + 20 + 16 * adv]);
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) { P0.p[pc] = b0[(3 + 3 * adv + pc) * 64]; Q0.p[pc] = b1[(3 + 3 * adv + pc) * 64]; }
      X3R32_STEP(P1, Q1)
    }
    ao += 32;
    b0 += 6 * 64;
    b1 += 6 * 64;
  }
  if (NKS & 1) {  // last step of an odd count: slot 0 and U hold it
    asm volatile("s_nop 1");
    X3R32_MFMA(U, P0, Q0, 1, 1); X3R32_MFMA(U, P0, Q0, 0, 2); X3R32_MFMA(U, P0, Q0, 2, 0);
    X3R32_MFMA(U, P0, Q0, 0, 1); X3R32_MFMA(U, P0, Q0, 1, 0); X3R32_MFMA(U, P0, Q0, 0, 0);
  }
  // The MFMAs above are opaque statements: the compiler's hazard recognizer does not know that c0 / c1 were written by the
  // matrix pipe, and the caller's epilogue reads them with VALU instructions right away (XDL write -> VALU read needs up to 19
  // wait states for a 16-pass instruction).
  asm volatile("s_nop 15\n\ts_nop 7" : "+v"(c0), "+v"(c1));
}
// the 64-row form: two row blocks share every weight fragment (c[column block][row block]); same pipeline, 24 MFMAs per step
template <int LDA, int NKS>
__device__ __forceinline__ void gemm_x3_r64(int a_off, const u32x4* __restrict__ Bx0, const u32x4* __restrict__ Bx1, f32x16& c00,
                                            f32x16& c01, f32x16& c10, f32x16& c11, int lane) {
  constexpr int kAhead = NKS < MOBROB_X3_AHEAD ? NKS : MOBROB_X3_AHEAD;
  const int r = lane & 31, h = lane >> 5;
  const int ab = 4 * opaque((a_off + r * LDA + 8 * h) >> 2);
  X3Frag P[kAhead + 1], Q[kAhead + 1];
#pragma unroll
  for (int k = 0; k < kAhead; ++k) { P[k] = x3_load_b(Bx0, k, lane); Q[k] = x3_load_b(Bx1, k, lane); }
  X3Frag U = x3_split8(*reinterpret_cast<const f32x4*>(&lds[ab]), *reinterpret_cast<const f32x4*>(&lds[ab + 4]));
  X3Frag V = x3_split8(*reinterpret_cast<const f32x4*>(&lds[ab + 32 * LDA]), *reinterpret_cast<const f32x4*>(&lds[ab + 32 * LDA + 4]));
  f32x4 ua = {0.f, 0.f, 0.f, 0.f}, ub = ua, va = ua, vb = ua;  // float32 A fragments of step ks + 1
  if (NKS > 1) {
    ua = *reinterpret_cast<const f32x4*>(&lds[ab + 16]);
    ub = *reinterpret_cast<const f32x4*>(&lds[ab + 20]);
    va = *reinterpret_cast<const f32x4*>(&lds[ab + 32 * LDA + 16]);
    vb = *reinterpret_cast<const f32x4*>(&lds[ab + 32 * LDA + 20]);
  }
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
#if MOBROB_X3_PIPE
    __builtin_amdgcn_sched_barrier(0);
#endif
    f32x4 na = ua, nb = ub, ma = va, mb = vb;
    if (ks + 2 < NKS) {
      na = *reinterpret_cast<const f32x4*>(&lds[ab + 16 * (ks + 2)]);
      nb = *reinterpret_cast<const f32x4*>(&lds[ab + 16 * (ks + 2) + 4]);
      ma = *reinterpret_cast<const f32x4*>(&lds[ab + 32 * LDA + 16 * (ks + 2)]);
      mb = *reinterpret_cast<const f32x4*>(&lds[ab + 32 * LDA + 16 * (ks + 2) + 4]);
    }
    if (ks + kAhead < NKS) {
      P[(ks + kAhead) % (kAhead + 1)] = x3_load_b(Bx0, ks + kAhead, lane);
      Q[(ks + kAhead) % (kAhead + 1)] = x3_load_b(Bx1, ks + kAhead, lane);
    }
    const X3Frag& p = P[ks % (kAhead + 1)];
    const X3Frag& q = Q[ks % (kAhead + 1)];
    X3_MFMA6(U, p, c00)
    X3_MFMA6(V, p, c01)
    X3_MFMA6(U, q, c10)
    X3_MFMA6(V, q, c11)
    if (ks + 1 < NKS) { U = x3_split8(ua, ub); V = x3_split8(va, vb); }
    ua = na; ub = nb; va = ma; vb = mb;
#if MOBROB_X3_PIPE
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
    __builtin_amdgcn_sched_group_barrier(0x020, 6, 0);
#pragma unroll
    for (int g = 0; g < 24; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
    }
#endif
  }
}
// the x3 packs of up to four matrices in one launch (blockIdx.y = matrix): both hidden layers of both networks
struct PackX3Args {
  const float* W[6]; int N[6], K[6], ld[6], NB[6], KS[6], trans[6];  // trans: the operand is W^T (B[k][n] = W[k][n])
  float scale[6];
  unsigned short* out[6];
};
__global__ __launch_bounds__(256) void k_pack_x3_multi(PackX3Args a) {
  const int m = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // (cb, ks, lane, j)
  const int KS = a.KS[m];
  if (i >= a.NB[m] * KS * 512) return;
  const int j = i & 7, lane = (i >> 3) & 63, ks = (i >> 9) % KS, cb = (i >> 9) / KS;
  const int n = cb * 32 + (lane & 31), k = ks * 16 + 8 * (lane >> 5) + j;
  float x = 0.f;
  if (n < a.N[m] && k < a.K[m]) x = a.scale[m] * (a.trans[m] ? a.W[m][(size_t)k * a.ld[m] + n] : a.W[m][(size_t)n * a.ld[m] + k]);
  asm volatile("" : "+v"(x));  // the rounded float32 product is what gets split (see x3_pack_store)
  unsigned p1, p2, p3;
  x3_split2(x, 0.f, p1, p2, p3);
  unsigned short* out = a.out[m];
  const size_t base = ((size_t)(cb * KS + ks) * 3) * 512 + (size_t)lane * 8 + j;
  out[base] = (unsigned short)(p1 & 0xffffu);
  out[base + 512] = (unsigned short)(p2 & 0xffffu);
  out[base + 1024] = (unsigned short)(p3 & 0xffffu);
}

// forward of the two hidden layers of a 64-row tile on the bf16 pipe (the x3 form of tile_layers; forward-only kernels)
template <int DP>
__device__ __forceinline__ Frag2 tile_layers_x3(const FusedNet& W, int wave, int lane) {
  using L = Lay<DP>;
  const int r_ = lane & 31;
  const u32x4* W1x = reinterpret_cast<const u32x4*>(W.W1x);
  const u32x4* W2x = reinterpret_cast<const u32x4*>(W.W2x);
  {
    const float bz0 = W.b1s[64 * wave + r_], bz1 = W.b1s[64 * wave + 32 + r_];
    f32x16 c00 = splat16(bz0), c01 = splat16(bz0), c10 = splat16(bz1), c11 = splat16(bz1);
    gemm_x3_r64<L::LDX, DP / 16>(L::X, W1x + (size_t)(2 * wave) * (DP / 16) * 192, W1x + (size_t)(2 * wave + 1) * (DP / 16) * 192,
                                 c00, c01, c10, c11, lane);
    store_tanh(L::H1, wave, lane, c00, c01, c10, c11);
  }
  __syncthreads();
  const f32x4* bp = W.W3h;
  Frag2 f3;
  {
    const float bz0 = W.b2s[64 * wave + r_], bz1 = W.b2s[64 * wave + 32 + r_];
    f32x16 c00 = splat16(bz0), c01 = splat16(bz0), c10 = splat16(bz1), c11 = splat16(bz1);
    gemm_x3_r64<FLDH, FH / 16>(L::H1, W2x + (size_t)(2 * wave) * (FH / 16) * 192, W2x + (size_t)(2 * wave + 1) * (FH / 16) * 192,
                               c00, c01, c10, c11, lane);
    f3 = prefetch_frag(bp, bp + 64, lane);
    store_tanh(L::H2, wave, lane, c00, c01, c10, c11);
  }
  __syncthreads();
  return f3;
}

// the same forward for the training kernel: one 32-row block at a time (two accumulators instead of four: the training kernel
// holds dW1 / dW3 and the next tile's operands in registers and has no room for the 64-row form's operand ring); every
// element sees the arithmetic of gemm_x3_r32, i.e. the bits of the rollout kernel's forward
__device__ __forceinline__ void store_tanh_r32(int dst_off, int rb, int wave, int lane, const f32x16& c0, const f32x16& c1) {
  const int r = lane & 31, h = lane >> 5;
  const int o = opaque(dst_off + (32 * rb + 4 * h) * FLDH + 64 * wave + r);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    lds[o + crc(i) * FLDH] = fast_tanh_scaled(c0[i]);
    lds[o + crc(i) * FLDH + 32] = fast_tanh_scaled(c1[i]);
  }
}
// lds[hs][rows of block rb][this wave's 64 columns] <- acc * (1 - hs^2) in place (the one-row-block form of dtanh_inplace)
__device__ __forceinline__ void dtanh_inplace_r32(int hs_off, int rb, int wave, int lane, const f32x16& c0, const f32x16& c1) {
  const int r = lane & 31, h = lane >> 5;
  const int o = opaque(hs_off + (32 * rb + 4 * h) * FLDH + 64 * wave + r);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    float hv;
    hv = lds[o + crc(i) * FLDH];      lds[o + crc(i) * FLDH] = c0[i] * (1.0f - hv * hv);
    hv = lds[o + crc(i) * FLDH + 32]; lds[o + crc(i) * FLDH + 32] = c1[i] * (1.0f - hv * hv);
  }
}
template <int DP, bool H16>
__device__ __forceinline__ Frag2 tile_layers_x3_train(const FusedNet& W, int wave, int lane) {
  using L = Lay<DP>;
  const int r_ = lane & 31;
  const u32x4* W1x = reinterpret_cast<const u32x4*>(W.W1x);
  const u32x4* W2x = reinterpret_cast<const u32x4*>(W.W2x);
  {
    const float bz0 = W.b1s[64 * wave + r_], bz1 = W.b1s[64 * wave + 32 + r_];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      f32x16 c0 = splat16(bz0), c1 = splat16(bz1);
      gemm_x3_r32_lean<L::LDX, DP / 16>(L::X + rb * 32 * L::LDX, W1x + (size_t)(2 * wave) * (DP / 16) * 192,
                                        W1x + (size_t)(2 * wave + 1) * (DP / 16) * 192, c0, c1, lane);
      store_tanh_r32(L::H1, rb, wave, lane, c0, c1);
    }
  }
  __syncthreads();
  const f32x4* bp = H16 ? W.W3h : W.W3f + (size_t)((wave >> 1) * 16) * 64;
  Frag2 f3;
  {
    const float bz0 = W.b2s[64 * wave + r_], bz1 = W.b2s[64 * wave + 32 + r_];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      f32x16 c0 = splat16(bz0), c1 = splat16(bz1);
      gemm_x3_r32_lean<FLDH, FH / 16>(L::H1 + rb * 32 * FLDH, W2x + (size_t)(2 * wave) * (FH / 16) * 192,
                                      W2x + (size_t)(2 * wave + 1) * (FH / 16) * 192, c0, c1, lane);
      if (rb == 1) f3 = prefetch_frag(bp, bp + 64, lane);
      store_tanh_r32(L::H2, rb, wave, lane, c0, c1);
    }
  }
  __syncthreads();
  return f3;
}

// dW2 += dz2^T . h1 on the bf16 pipe: one product (A piece, B piece) of the tile pair (neuron block 0 / 1, input block jb); the
// accumulators stay pinned in AGPRs ("+a") like those of dw2_kstep.  Two MFMAs per statement so that the VALU work that
// splits the next operands can sit between the statements.
#define DW2X_MFMA(ia, ib)                                                               \
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %4, %0\n\t"                            \
               "v_mfma_f32_32x32x16_bf16 %1, %3, %4, %1"                                \
               : "+a"(g0), "+a"(g1)                                                     \
               : "v"(A0.p[ia]), "v"(A1.p[ia]), "v"(B.p[ib]))
// dW1 += dz1^T . X on the bf16 pipe: one product for the four (two) tiles; accumulators in arch VGPRs ("+v") like mfma_x2y2's
#define DW1X_MFMA4(ia, ib)                                                              \
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %4, %6, %0\n\t"                            \
               "v_mfma_f32_32x32x16_bf16 %1, %5, %6, %1\n\t"                            \
               "v_mfma_f32_32x32x16_bf16 %2, %4, %7, %2\n\t"                            \
               "v_mfma_f32_32x32x16_bf16 %3, %5, %7, %3"                                \
               : "+v"(gW1a), "+v"(gW1b), "+v"(gW1c), "+v"(gW1d)                         \
               : "v"(A0.p[ia]), "v"(A1.p[ia]), "v"(B0.p[ib]), "v"(B1.p[ib]))
#define DW1X_MFMA2(ia, ib)                                                              \
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %4, %0\n\t"                            \
               "v_mfma_f32_32x32x16_bf16 %1, %3, %4, %1"                                \
               : "+v"(gW1a), "+v"(gW1b)                                                 \
               : "v"(A0.p[ia]), "v"(A1.p[ia]), "v"(B0.p[ib]))
// eight consecutive batch rows of one column: the float32 side of an x3 fragment of a TRANSPOSED operand (k = batch row)
struct ColFrag { float v[8]; };
template <int LD = FLDH>
__device__ __forceinline__ ColFrag col_frag_load(int off) {
  ColFrag f;
#pragma unroll
  for (int j = 0; j < 8; ++j) f.v[j] = lds[off + j * LD];
  return f;
}
__device__ __forceinline__ void col_frag_split_pair(const ColFrag& f, int jp, X3Frag& out) {  // elements 2 jp, 2 jp + 1
  unsigned p1, p2, p3;
  x3_split2(f.v[2 * jp], f.v[2 * jp + 1], p1, p2, p3);
  out.p[0][jp] = p1; out.p[1][jp] = p2; out.p[2][jp] = p3;
}
__device__ __forceinline__ X3Frag col_frag_split(const ColFrag& f) {
  X3Frag o;
#pragma unroll
  for (int jp = 0; jp < 4; ++jp) col_frag_split_pair(f, jp, o);
  return o;
}
// the tile pair of input block jb for one 16-row k step; meanwhile the fragment of the next input block is split
// (four pair-splits between the six two-MFMA statements) and the one after it is read from LDS
__device__ __forceinline__ void dw2_x3_block(f32x16& g0, f32x16& g1, const X3Frag& A0, const X3Frag& A1, const X3Frag& B,
                                             const ColFrag& next_raw, X3Frag& next_split) {
  asm volatile("s_nop 1");  // VALU-written operand -> MFMA
  DW2X_MFMA(1, 1);
  col_frag_split_pair(next_raw, 0, next_split);
  DW2X_MFMA(0, 2);
  col_frag_split_pair(next_raw, 1, next_split);
  DW2X_MFMA(2, 0);
  col_frag_split_pair(next_raw, 2, next_split);
  DW2X_MFMA(0, 1);
  col_frag_split_pair(next_raw, 3, next_split);
  DW2X_MFMA(1, 0);
  DW2X_MFMA(0, 0);
}

// ------------------------------------------------------------------------------------------------
// The training kernel.  H16: both heads are <= 16 wide (A <= 16) -> 16x16x4 head / dW3, dh2 over K = 16.
// ------------------------------------------------------------------------------------------------
template <int DP, bool H16, bool X3 = false>
__global__ __launch_bounds__(FTHREADS, 1) void k_fused_train(FusedTrainArgs a) {
  using L = Lay<DP>;
  constexpr int ldx = L::LDX, per = DP / 4;
  const int tid0 = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);  // wave-uniform -> SGPR addressing of weights/slabs
  // blockIdx -> (network, tile sequence).  The hardware deals workgroups round-robin over the 8 XCDs (XCD = blockIdx
  // mod 8, each with a private L2).  Blocks are taken in groups of 16: the first 8 run the policy network for tile
  // sequences 8g .. 8g+7, the next 8 the value network for the SAME sequences, so the two workgroups that gather the
  // same observation rows sit on the same XCD and the second gather hits that XCD's L2 instead of HBM (with the old
  // `blockIdx & 1` mapping they sat on different XCDs and every X row came from HBM twice).  A last partial group of
  // r blocks is split r/2 : r/2.  Each L2 holds the fragment packs of both networks (1.2 MB of 4 MB).
  const int nwg = gridDim.x >> 1;
  int net, wg;
  {
    const int b = blockIdx.x, g16 = b >> 4, o = b & 15;
    const int gsz = min(16, (int)gridDim.x - 16 * g16), half = gsz >> 1;
    net = o >= half ? 1 : 0;
    wg = 8 * g16 + (o - net * half);
  }
  const int slab_id = 2 * wg + net;   // k_slab_reduce expects the slabs of a network at stride 2
  const FusedNet W = a.net[net];
  const int ntiles = (a.count + FR - 1) / FR;

  // dW2 lives in the 256 accumulator registers for the whole kernel; dW1 / dW3 are small and are accumulated
  // per tile into the workgroup's private slab (read-modify-write by the owning lane, L2 resident).
  f32x16 gW2[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) gW2[t] = zero16();
  // dW1 (4 tiles) and dW3 (2 tiles) accumulate in arch VGPRs across all tiles of the workgroup (96 registers):
  // the hot GEMM loops need ~100 VGPRs, so everything fits without spilling and no per-tile slab traffic remains.
  f32x16 gW1a = zero16(), gW1b = zero16(), gW1c = zero16(), gW1d = zero16();  // [ib][jb] = 00, 10, 01, 11
  f32x16 gW3a = zero16(), gW3b = zero16();  // 32-wide heads: [32][this wave's 64 columns]
  f32x4 gW3h0 = {0.f, 0.f, 0.f, 0.f}, gW3h1 = gW3h0, gW3h2 = gW3h0, gW3h3 = gW3h0;  // H16: [16][64] as 4 16x16 tiles
  float* slab = a.slabs + (size_t)slab_id * a.slab_floats;
  float* slab_w1 = slab + slab_off_w1();
  float* slab_w3 = slab + slab_off_w3(DP);
  float gb2 = 0.f, gb1 = 0.f;  // bias gradients of hidden column `tid`
  float s_pl = 0.f, s_vl = 0.f, s_kl = 0.f, s_cf = 0.f;  // loss statistics (lanes with q == 0)
  // per-action constants of the Gaussian head and the per-wave head-gradient accumulators
  if (tid0 < 32) {
    const int k = tid0;
    float iv = 0.f, lc = 0.f, bb = 0.f;
    if (net == 0 && k < a.A) {
      const float sd = expf(a.log_std[k]);
      iv = 1.0f / (sd * sd);
      lc = logf(sd) + 0.91893853320467274178f;
    }
    if (k < W.head) bb = W.b3[k];
    lds[L::CST + k] = iv;
    lds[L::CST + 32 + k] = lc;
    lds[L::CST + 64 + k] = bb;
  }
  lds[L::GACC + tid0] = 0.f;

  float adv_mean = 0.f, adv_sd = 1.f;
  bool adv_on = false;
  {
    const double n = a.advstat[2];
    adv_on = n > 1.0;
    const double m = a.advstat[0] / (n > 0 ? n : 1.0);
    double var = adv_on ? (a.advstat[1] - n * m * m) / (n - 1.0) : 0.0;
    if (var < 0.0) var = 0.0;
    adv_mean = (float)m;
    adv_sd = (float)sqrt(var);
  }

#ifdef MOBROB_STAMPS
  unsigned long long* stamps_ = a.stamps;
#endif
  STAMP_INIT()
  // software-pipelined gather: xr holds the X rows of the tile about to be processed (zero beyond the minibatch)
  constexpr int NG = (FR * per + FTHREADS - 1) / FTHREADS;
  static_assert((FR * per) % FTHREADS == 0, "gather assumes a whole number of 16-byte chunks per thread");
  f32x4 xr[NG];
#pragma unroll
  for (int u = 0; u < NG; ++u) {
    const int i = tid0 + u * FTHREADS, rr = i / per, c = i - rr * per;
    xr[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (PHASE_ON(1) && wg < ntiles && wg * FR + rr < a.count)
      xr[u] = ldg16(a.obs, (unsigned)a.rows[wg * FR + rr] * (unsigned)(DP * 4) + (unsigned)(c * 16));
  }
  // Operands of the loss stage (4 lanes per row, q = action residue mod 4) of the tile about to be processed: a
  // two-level gather (row index, then actions / advantage / old log-prob or return of that row).  vmcnt retires in
  // order, so a long-latency gather that is in flight when a GEMM phase waits for its next weight fragment stalls
  // that phase for the whole HBM / TLB latency.  All gathers for tile t+1 are therefore issued inside the dW2 phase
  // of tile t -- 13 us of MFMAs fed from LDS only, no global load to wait for -- row indices at its start, the
  // dependent rows at its mid-point; here: the first tile (exposed once per launch).
  constexpr int NJ = H16 ? 4 : 8;  // action columns per lane (4 lanes per row)
  // Wide heads on wide observations (A > 16, D > 32: none of the reference robots) are out of registers with eight
  // action values per lane held across the dW2 phase: they keep the ROW INDEX instead and load the actions in the loss
  // stage itself (an exposed gather per tile instead of register spills).
  constexpr bool kActAhead = H16 || DP <= 32;
  float l_adv = 0.f, l_old = 0.f, l_act[NJ];
  int l_src = -1;
  auto load_actions = [&](int src_or_neg, int lq_) {
    const bool live = src_or_neg >= 0, pol = net == 0;
    const unsigned src = live ? (unsigned)src_or_neg : 0u;
    const unsigned aoff = (src * (unsigned)a.RW + (unsigned)lq_) * 4u;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      l_act[j] = (pol && 4 * j + lq_ < a.A && live)
                     ? *reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.rec) + (aoff + 16u * j))
                     : 0.f;
  };
  auto gather_loss = [&](int src_or_neg, int lq_) {  // one load per destination register, each after its zero init
    const bool live = src_or_neg >= 0, pol = net == 0;
    const unsigned src = live ? (unsigned)src_or_neg : 0u;
    if constexpr (kActAhead) load_actions(src_or_neg, lq_);
    else l_src = src_or_neg;
    const float* tail = a.rec + (size_t)src * a.RW + (a.RW - 4);  // [old log-prob, advantage, return, old value]
    l_old = live ? tail[pol ? 0 : 2] : 0.f;                        // policy: old log-prob; value net: return target
    l_adv = live ? (pol ? tail[1] : (a.clip_vf >= 0.f ? tail[3] : 0.f)) : 0.f;  // value net: old value (vf clipping)
  };
  {
    const int lrr0 = tid0 >> 2;
    gather_loss((wg < ntiles && wg * FR + lrr0 < a.count) ? a.rows[wg * FR + lrr0] : -1, tid0 & 3);
  }
  for (int tile = wg; tile < ntiles; tile += nwg) {
    // Per-lane indices are re-derived from an opaque copy of the thread id in every tile: otherwise LLVM hoists
    // every address base that depends only on the lane out of this loop and keeps ~40 of them live (and spilled).
    const int tid = opaque(tid0), lane = tid & 63;
    const int r = lane & 31, h = lane >> 5;
    const int row0 = tile * FR;
    Frag2 f1;
    if constexpr (!X3) f1 = prefetch_frag(W.W1f + (size_t)(2 * wave) * (DP / 8) * 64,
                                          W.W1f + (size_t)(2 * wave + 1) * (DP / 8) * 64, lane);
    // ---- the observation rows of this tile were fetched during the previous tile's backward pass ----
#pragma unroll
    for (int u = 0; u < NG; ++u) {
      const int i = tid + u * FTHREADS, rr = i / per, c = i - rr * per;
      *reinterpret_cast<f32x4*>(&lds[L::X + rr * ldx + 4 * c]) = xr[u];
    }
    const int nrow0 = (tile + nwg) * FR;
    const bool has_next = PHASE_ON(1) && tile + nwg < ntiles;
    const int lrr = tid >> 2, lq = tid & 3;
    const bool llive = row0 + lrr < a.count;
    __syncthreads();
    STAMP(0)
    Frag2 f3;
    if constexpr (X3) f3 = tile_layers_x3_train<DP, H16>(W, wave, lane);  // forward on the bf16 pipe (float32 operands split three ways)
    else f3 = tile_layers<DP, H16>(W, wave, lane, f1 STAMP_ARGS);
    if (PHASE_ON(8)) {
      if constexpr (H16) tile_head16<DP>(W, wave, lane, f3);
      else tile_head<DP>(W, wave, lane, f3 STAMP_ARGS);
    }
    STAMP(8)

    const Frag2 fh2 = prefetch_frag(W.W3b + (size_t)(2 * wave) * 4 * 64, W.W3b + (size_t)(2 * wave + 1) * 4 * 64, lane);
    // ---- loss: 4 lanes per row (q = action residue mod 4), all waves; writes dL/d(head output) into the
    //      head tile (zero padded) and accumulates the head-bias / log_std gradient sums per wave ----
    if (PHASE_ON(16)) {
      const int rr = lrr, q = lq;
      const bool live = llive;
      const int db = opaque(L::DO + rr * FLDO + q);  // head tile row, this lane's action residue
      const int cb = opaque(L::CST + q);             // per-action constants: [0]=1/var [32]=log terms [64]=bias
      const int gb = opaque(L::GACC + wave * 64 + q);
      const int A = a.A;
      if (net == 0) {
        if constexpr (!kActAhead) load_actions(l_src, q);
        float lp = 0.f;
        float dk[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          float d = 0.f;
          if (4 * j + q < A && live) {
            d = l_act[j] - (lds[db + 4 * j] + lds[cb + 64 + 4 * j]);
            lp += -(d * d) * (0.5f * lds[cb + 4 * j]) - lds[cb + 32 + 4 * j];
          }
          dk[j] = d;
        }
        lp += xor_lane(lp, 1);
        lp += xor_lane(lp, 2);
        float g_logp = 0.f;
        if (live) {
          float adv = l_adv;
          if (a.normalize && adv_on) adv = (adv - adv_mean) / (adv_sd + 1e-8f);
          const float log_ratio = lp - l_old;
          const float ratio = expf(log_ratio);
          const float lo = 1.0f - a.clip, hi = 1.0f + a.clip;
          const float s1 = adv * ratio, s2 = adv * fminf(fmaxf(ratio, lo), hi);
          if (q == 0) {
            s_pl += fminf(s1, s2);
            s_cf += (fabsf(ratio - 1.0f) > a.clip) ? 1.f : 0.f;
            s_kl += (ratio - 1.0f) - log_ratio;
          }
          const float in_range = (ratio >= lo && ratio <= hi) ? 1.f : 0.f;
          const float w1 = (s1 < s2) ? 1.f : ((s1 > s2) ? 0.f : 0.5f);
          g_logp = -(w1 * adv + (1.0f - w1) * adv * in_range) * a.inv_bg * ratio;
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const float iv = lds[cb + 4 * j];
          float gm = g_logp * dk[j] * iv;  // zero for k >= A (dk = 0, iv = 0)
          float gl = (4 * j + q < A) ? g_logp * (dk[j] * dk[j] * iv - 1.0f) : 0.f;
          lds[db + 4 * j] = gm;
          if (4 * j < A) {  // wave-uniform: sum over the 16 rows of this wave (lanes with equal q)
#pragma unroll
            for (int o = 4; o < 64; o <<= 1) {
              gm += xor_lane(gm, o);
              gl += xor_lane(gl, o);
            }
            if (lane < 4) {
              lds[gb + 4 * j] += gm;
              lds[gb + 32 + 4 * j] += gl;
            }
          }
        }
      } else {
        float dv = 0.f;
        if (live && q == 0) {
          float sq, gv_;
          value_loss_terms(lds[db] + lds[cb + 64], l_old, l_adv, a.clip_vf, sq, gv_);
          s_vl += sq;
          dv = a.vf_coef * gv_ * a.inv_bg;
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) lds[db + 4 * j] = (j == 0) ? dv : 0.f;  // q != 0 lanes hold dv = 0
        const float t = wave_sum(dv);
        if (lane == 0) lds[gb] += t;
      }
    }
    __syncthreads();
    STAMP(9)

    // ---- dW3 += dout^T . h2  (M = 32 or 16 head rows, this wave's 64 columns, K = 64 rows) and
    //      dh2 = dout . W3 (K = 16 or 32), then dz2 = dh2 * (1 - h2^2) in place ----
    // Narrow heads (H16): the dh2 GEMM is two k-groups of 16 MFMAs, too short to cover the L2 latency of its own second
    // weight fragment, and dW3 is fed from LDS only.  They are therefore issued as ONE phase: dh2 k-group 0 (fragment
    // prefetched before the loss stage) -> request k-group 1's fragment -> the whole dW3 loop -> dh2 k-group 1, whose
    // fragment has had 2 000 MFMA cycles to arrive (as separate phases: dW3 + dh2 ran at ~63 % of the MFMA rate).
    auto dw3_h16 = [&]() {
      // lane group g = lane>>4 takes batch rows kk + 4g (kk = 16s + j): LDS bank offset 16g -> conflict-free
      const int i16 = lane & 15, g4 = 4 * (lane >> 4);
      const int ao = opaque(L::DO + g4 * FLDO + i16);              // A[i=a][k=row] = dout[row][a]
      const int bo = opaque(L::H2 + g4 * FLDH + 64 * wave + i16);  // B[k=row][j]  = h2[row][j], 4 column tiles
      float x = lds[ao], y0 = lds[bo], y1 = lds[bo + 16], y2 = lds[bo + 32], y3 = lds[bo + 48];
#pragma unroll 5
      for (int t = 1; t < 16; ++t) {
        const int kk = (t >> 2) * 16 + (t & 3);
        const float xn = lds[ao + kk * FLDO];
        const float* bk = &lds[bo + kk * FLDH];
        const float y0n = bk[0], y1n = bk[16], y2n = bk[32], y3n = bk[48];
        mfma16_x1y4(gW3h0, gW3h1, gW3h2, gW3h3, x, y0, y1, y2, y3);
        x = xn; y0 = y0n; y1 = y1n; y2 = y2n; y3 = y3n;
      }
      mfma16_x1y4(gW3h0, gW3h1, gW3h2, gW3h3, x, y0, y1, y2, y3);
    };
    if (PHASE_ON(32)) {
      if constexpr (!H16) {
        const int ao = opaque(L::DO + h * FLDO + r);              // A[i=a][k=row] = dout[row][a]
        const int bo = opaque(L::H2 + h * FLDH + 64 * wave + r);  // B[k=row][j]  = h2[row][j]
        float x = lds[ao], y0 = lds[bo], y1 = lds[bo + 32];
#pragma unroll 4
        for (int k = 0; k < FR - 2; k += 2) {
          const float xn = lds[ao + (k + 2) * FLDO], y0n = lds[bo + (k + 2) * FLDH], y1n = lds[bo + (k + 2) * FLDH + 32];
          mfma_x1y2(gW3a, gW3b, x, y0, y1);
          x = xn; y0 = y0n; y1 = y1n;
        }
        mfma_x1y2(gW3a, gW3b, x, y0, y1);
      } else if (!MOBROB_DW3_INSIDE_DH2) {
        dw3_h16();
      }
    }
    STAMP(10)
    {
      f32x16 c00 = zero16(), c01 = zero16(), c10 = zero16(), c11 = zero16();
      if (PHASE_ON(64)) {
        if constexpr (H16 && MOBROB_DW3_INSIDE_DH2) {
          const f32x4* Bp0 = W.W3b + (size_t)(2 * wave) * 4 * 64;
          const f32x4* Bp1 = W.W3b + (size_t)(2 * wave + 1) * 4 * 64;
          const int ab = 4 * opaque((L::DO + r * FLDO + 4 * h) >> 2);
          const unsigned bo = opaque_u((unsigned)lane * 16u);
          const f32x4 pA = fh2.p, qA = fh2.q;
          const f32x4 pB = ldg16(Bp0, bo + 1024u), qB = ldg16(Bp1, bo + 1024u);
          const f32x4 uA = *reinterpret_cast<const f32x4*>(&lds[ab]);
          const f32x4 vA = *reinterpret_cast<const f32x4*>(&lds[ab + 32 * FLDO]);
          const f32x4 uB = *reinterpret_cast<const f32x4*>(&lds[ab + 8]);
          const f32x4 vB = *reinterpret_cast<const f32x4*>(&lds[ab + 32 * FLDO + 8]);
          MFMA_KG(uA, vA, pA, qA)
          if (PHASE_ON(32)) dw3_h16();
          MFMA_KG(uB, vB, pB, qB)
        } else {
          gemm_lds_packed<FLDO>(L::DO, W.W3b + (size_t)(2 * wave) * 4 * 64, W.W3b + (size_t)(2 * wave + 1) * 4 * 64,
                                H16 ? 2 : 4,  // k-groups of 8: head columns beyond `head` are zero padding
                                c00, c01, c10, c11, lane, fh2);
        }
      }
      STAMP(11)
      __syncthreads();  // every wave is done reading h2 (dW3) before it is overwritten
      STAMP(12)
      if (PHASE_ON(1024)) dtanh_inplace(L::H2, wave, lane, c00, c01, c10, c11);
      else asm volatile("" ::"v"(c00), "v"(c01), "v"(c10), "v"(c11));
      STAMP(13)
    }
    __syncthreads();
    STAMP(14)
    // ---- dW2 += dz2^T . h1  (this wave: 64 neurons x 256 inputs, K = 64 rows) ----
    Frag2 fh1;
    int nsrc_x[NG], lsrc_next_x = -1;  // X3: row indices of the next tile (loaded in the dW2 phase, used at the start of dW1)
    if constexpr (!X3) fh1 = prefetch_frag(W.W2b + (size_t)(2 * wave) * (FH / 8) * 64, W.W2b + (size_t)(2 * wave + 1) * (FH / 8) * 64, lane);
    if constexpr (X3) {
      // On the bf16 pipe: four k steps of 16 batch rows.  Both operands are TRANSPOSED reads of LDS tiles (k = batch row): a
      // lane's fragment is eight rows of one column (col_frag_load: eight ds_read_b32, the same number of LDS reads per row
      // as the f32 loop), split into three bf16 pieces in registers.  Per step: the two neuron-block fragments of dz2, then
      // for each of the eight input blocks of h1 twelve MFMAs (two tiles x six products) with the next block's split
      // between them.  The bias gradient gb2 and the next tile's gathers ride in the loop as in the f32 form.
      const int ao = opaque(L::H2 + 8 * h * FLDH + 64 * wave + r);
      const int bo = opaque(L::H1 + 8 * h * FLDH + r);
      const int co = opaque(L::H2 + tid);
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      // level 1 of the next tile's gathers (row indices) here; level 2 (the rows) at the start of the dW1 phase: held from
      // the middle of this phase, as the f32 form does, the sixteen registers of the rows spilled at DP = 64
#pragma unroll
      for (int u = 0; u < NG; ++u) {
        const int rr = (tid + u * FTHREADS) / per;
        nsrc_x[u] = (has_next && nrow0 + rr < a.count) ? a.rows[nrow0 + rr] : -1;
      }
      lsrc_next_x = (has_next && nrow0 + lrr < a.count) ? a.rows[nrow0 + lrr] : -1;
#pragma unroll 1
      for (int ks = 0; ks < FR / 16; ++ks) {
        const int ro = 16 * ks * FLDH;
        const X3Frag A0 = col_frag_split(col_frag_load(ao + ro)), A1 = col_frag_split(col_frag_load(ao + ro + 32));
#pragma unroll
        for (int j = 0; j < 16; j += 4) {  // rows = 0..3 (mod 4) -> s0..s3: column_sum()'s order
          s0 += lds[co + ro + j * FLDH];
          s1 += lds[co + ro + (j + 1) * FLDH];
          s2 += lds[co + ro + (j + 2) * FLDH];
          s3 += lds[co + ro + (j + 3) * FLDH];
        }
        X3Frag B = col_frag_split(col_frag_load(bo + ro));
        ColFrag raw = col_frag_load(bo + ro + 32);
#pragma unroll
        for (int jb = 0; jb < 8; ++jb) {
          X3Frag Bn;
          __builtin_amdgcn_sched_barrier(0);  // one block's reads at a time (hoisted, the eight blocks' 64 values spilled at DP = 64)
          const ColFrag raw2 = col_frag_load(bo + ro + 32 * (jb + 2 < 8 ? jb + 2 : 7));  // two input blocks ahead: no LDS wait in front of the splits
          dw2_x3_block(gW2[jb], gW2[8 + jb], A0, A1, B, raw, Bn);
          B = Bn;
          raw = raw2;
        }
      }
      gb2 += (s0 + s1) + (s2 + s3);
    } else if (PHASE_ON(128)) {
      const int ao = opaque(L::H2 + h * FLDH + 64 * wave + r);
      const int bo = opaque(L::H1 + h * FLDH + r);
      // The bias gradient gb2 (sum of column `tid` of dz2 over the 64 rows) rides in this loop like gb1 rides in the
      // dW1 loop: four partial sums over rows = 0..3 (mod 4), added as (s0 + s1) + (s2 + s3) -- column_sum()'s order,
      // bit for bit -- with their LDS reads under the MFMAs of the previous k-step instead of in front of the phase.
      const int co = opaque(L::H2 + tid);
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      if (PHASE_ON(2048)) { s0 += lds[co]; s1 += lds[co + FLDH]; }
#define GB2_ROWS(k)                                                                    \
  if (PHASE_ON(2048)) {                                                                \
    if (((k) & 2) == 0) { s0 += lds[co + (k) * FLDH]; s1 += lds[co + ((k) + 1) * FLDH]; } \
    else { s2 += lds[co + (k) * FLDH]; s3 += lds[co + ((k) + 1) * FLDH]; }             \
  }
      // operands of k-step k+2 are read from LDS before the 16 MFMAs of k-step k are issued (the MFMA statement is
      // opaque to the scheduler, so the lookahead is written out by hand)
      float x0 = lds[ao], x1 = lds[ao + 32];
      float y0 = lds[bo], y1 = lds[bo + 32], y2 = lds[bo + 64], y3 = lds[bo + 96], y4 = lds[bo + 128],
            y5 = lds[bo + 160], y6 = lds[bo + 192], y7 = lds[bo + 224];
      // level 1 of the next tile's gathers: row indices (X rows of the 64-row tile, loss-stage row of this lane)
      int nsrc[NG];
#pragma unroll
      for (int u = 0; u < NG; ++u) {
        const int rr = (tid + u * FTHREADS) / per;
        nsrc[u] = (has_next && nrow0 + rr < a.count) ? a.rows[nrow0 + rr] : -1;
      }
      const int lsrc_next = (has_next && nrow0 + lrr < a.count) ? a.rows[nrow0 + lrr] : -1;
#pragma unroll 2
      for (int k = 2; k < FR / 2; k += 2) {
        const float* bk = &lds[bo + k * FLDH];
        const float nx0 = lds[ao + k * FLDH], nx1 = lds[ao + k * FLDH + 32];
        const float n0 = bk[0], n1 = bk[32], n2 = bk[64], n3 = bk[96], n4 = bk[128], n5 = bk[160], n6 = bk[192],
                    n7 = bk[224];
        GB2_ROWS(k)
        dw2_kstep(gW2, x0, x1, y0, y1, y2, y3, y4, y5, y6, y7);
        x0 = nx0; x1 = nx1; y0 = n0; y1 = n1; y2 = n2; y3 = n3; y4 = n4; y5 = n5; y6 = n6; y7 = n7;
      }
      // level 2 (the indices have had half of the phase to arrive): the rows themselves, in flight during the
      // second half; written to LDS / consumed by the loss stage in the next tile
#pragma unroll
      for (int u = 0; u < NG; ++u) {
        const int c = (tid + u * FTHREADS) % per;
        xr[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (nsrc[u] >= 0) xr[u] = ldg16(a.obs, (unsigned)nsrc[u] * (unsigned)(DP * 4) + (unsigned)(c * 16));
      }
      gather_loss(lsrc_next, lq);
#pragma unroll 2
      for (int k = FR / 2; k < FR; k += 2) {
        const float* bk = &lds[bo + k * FLDH];
        const float nx0 = lds[ao + k * FLDH], nx1 = lds[ao + k * FLDH + 32];
        const float n0 = bk[0], n1 = bk[32], n2 = bk[64], n3 = bk[96], n4 = bk[128], n5 = bk[160], n6 = bk[192],
                    n7 = bk[224];
        GB2_ROWS(k)
        dw2_kstep(gW2, x0, x1, y0, y1, y2, y3, y4, y5, y6, y7);
        x0 = nx0; x1 = nx1; y0 = n0; y1 = n1; y2 = n2; y3 = n3; y4 = n4; y5 = n5; y6 = n6; y7 = n7;
      }
      dw2_kstep(gW2, x0, x1, y0, y1, y2, y3, y4, y5, y6, y7);
#undef GB2_ROWS
      gb2 += (s0 + s1) + (s2 + s3);
    } else if (PHASE_ON(2048)) {
      gb2 += column_sum(L::H2, tid);
    }
    STAMP(15)
    // ---- dh1 = dz2 . W2 (K = 256), then dz1 = dh1 * (1 - h1^2) in place ----
    if constexpr (X3) {  // on the bf16 pipe, one 32-row block at a time (gemm_x3_r32_lean)
      const u32x4* W2bx = reinterpret_cast<const u32x4*>(W.W2bx);
#pragma unroll 1
      for (int rb = 0; rb < 2; ++rb) {
        f32x16 c0 = zero16(), c1 = zero16();
        gemm_x3_r32_lean<FLDH, FH / 16>(L::H2 + rb * 32 * FLDH, W2bx + (size_t)(2 * wave) * (FH / 16) * 192,
                                        W2bx + (size_t)(2 * wave + 1) * (FH / 16) * 192, c0, c1, lane);
        if (rb == 0) __syncthreads();  // dW2 reads of h1 complete everywhere
        dtanh_inplace_r32(L::H1, rb, wave, lane, c0, c1);
      }
      STAMP(16)
    } else {
      f32x16 c00 = zero16(), c01 = zero16(), c10 = zero16(), c11 = zero16();
      constexpr int nkg = FH / 8;
      if (PHASE_ON(256))
        gemm_lds_packed_deep<FLDH>(L::H2, W.W2b + (size_t)(2 * wave) * nkg * 64,
                                   W.W2b + (size_t)(2 * wave + 1) * nkg * 64, nkg, c00, c01, c10, c11, lane, fh1);
      STAMP(16)
      __syncthreads();  // dW2 reads of h1 complete everywhere
      STAMP(17)
      if (PHASE_ON(1024)) dtanh_inplace(L::H1, wave, lane, c00, c01, c10, c11);
      else asm volatile("" ::"v"(c00), "v"(c01), "v"(c10), "v"(c11));
      STAMP(18)
    }
    __syncthreads();
    STAMP(19)
    // ---- dW1 += dz1^T . X  (this wave: 64 neurons x DP inputs, K = 64 rows) ----
    if constexpr (X3) {  // level 2 of the next tile's gathers: in flight under this phase's MFMAs (LDS operands only)
#pragma unroll
      for (int u = 0; u < NG; ++u) {
        const int c = (tid + u * FTHREADS) % per;
        xr[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (nsrc_x[u] >= 0) xr[u] = ldg16(a.obs, (unsigned)nsrc_x[u] * (unsigned)(DP * 4) + (unsigned)(c * 16));
      }
      gather_loss(lsrc_next_x, lq);
    }
    if constexpr (X3) {  // four k steps of 16 batch rows on the bf16 pipe; both operands are column fragments (dz1, X)
      constexpr bool two = DP > 32;
      const int ao = opaque(L::H1 + 8 * h * FLDH + 64 * wave + r);
      const int c0 = (r < DP) ? r : 0;
      const int c1 = (32 + r < DP) ? 32 + r : c0;  // clamped columns are never read back
      const int b0o = opaque(L::X + 8 * h * ldx + c0), b1o = opaque(L::X + 8 * h * ldx + c1);
      const int co = opaque(L::H1 + tid);
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll 1
      for (int ks = 0; ks < FR / 16; ++ks) {
        const int ro = 16 * ks * FLDH, rx = 16 * ks * ldx;
        const X3Frag A0 = col_frag_split(col_frag_load(ao + ro)), A1 = col_frag_split(col_frag_load(ao + ro + 32));
        const X3Frag B0 = col_frag_split(col_frag_load<ldx>(b0o + rx));
        X3Frag B1 = B0;
        if (two) B1 = col_frag_split(col_frag_load<ldx>(b1o + rx));
#pragma unroll
        for (int j = 0; j < 16; j += 4) {  // gb1: rows = 0..3 (mod 4) -> s0..s3, column_sum()'s order
          s0 += lds[co + ro + j * FLDH];
          s1 += lds[co + ro + (j + 1) * FLDH];
          s2 += lds[co + ro + (j + 2) * FLDH];
          s3 += lds[co + ro + (j + 3) * FLDH];
        }
        asm volatile("s_nop 1");
        if (two) {
          DW1X_MFMA4(1, 1); DW1X_MFMA4(0, 2); DW1X_MFMA4(2, 0); DW1X_MFMA4(0, 1); DW1X_MFMA4(1, 0); DW1X_MFMA4(0, 0);
        } else {
          DW1X_MFMA2(1, 1); DW1X_MFMA2(0, 2); DW1X_MFMA2(2, 0); DW1X_MFMA2(0, 1); DW1X_MFMA2(1, 0); DW1X_MFMA2(0, 0);
        }
      }
      asm volatile("s_nop 15\n\ts_nop 7");  // opaque MFMA statements: after the last tile the slab store reads gW1 (XDL write -> VMEM read)
      gb1 += (s0 + s1) + (s2 + s3);
    } else if (PHASE_ON(512)) {
      constexpr bool two = DP > 32;
      const int ao = opaque(L::H1 + h * FLDH + 64 * wave + r);
      const int c0 = (r < DP) ? r : 0;
      const int c1 = (32 + r < DP) ? 32 + r : c0;  // clamped columns are never read back
      const int b0o = opaque(L::X + h * ldx + c0), b1o = opaque(L::X + h * ldx + c1);
      // The bias gradient gb1 (sum of column `tid` of dz1 over the 64 rows) rides in the same loop: four partial sums
      // over rows = 0..3 (mod 4), added as (s0 + s1) + (s2 + s3) -- the order of column_sum() -- with their LDS reads
      // under the MFMAs instead of in front of them.  Two k-steps (4 rows) per iteration, ping-pong operand sets.
      const int co = opaque(L::H1 + tid);
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      float x0 = lds[ao], x1 = lds[ao + 32], y0 = lds[b0o], y1 = two ? lds[b1o] : 0.f;
      float xa, xb, ya, yb;
#define DW1_STEP(X0, X1, Y0, Y1)                                  \
  if (two) mfma_x2y2(gW1a, gW1b, gW1c, gW1d, X0, X1, Y0, Y1);     \
  else mfma_x2y1(gW1a, gW1b, X0, X1, Y0);
#pragma unroll 2
      for (int k = 0; k < FR - 4; k += 4) {
        xa = lds[ao + (k + 2) * FLDH]; xb = lds[ao + (k + 2) * FLDH + 32];
        ya = lds[b0o + (k + 2) * ldx]; yb = two ? lds[b1o + (k + 2) * ldx] : 0.f;
        if (PHASE_ON(2048)) { s0 += lds[co + k * FLDH]; s1 += lds[co + (k + 1) * FLDH]; }
        DW1_STEP(x0, x1, y0, y1)
        x0 = lds[ao + (k + 4) * FLDH]; x1 = lds[ao + (k + 4) * FLDH + 32];
        y0 = lds[b0o + (k + 4) * ldx]; y1 = two ? lds[b1o + (k + 4) * ldx] : 0.f;
        if (PHASE_ON(2048)) { s2 += lds[co + (k + 2) * FLDH]; s3 += lds[co + (k + 3) * FLDH]; }
        DW1_STEP(xa, xb, ya, yb)
      }
      xa = lds[ao + (FR - 2) * FLDH]; xb = lds[ao + (FR - 2) * FLDH + 32];
      ya = lds[b0o + (FR - 2) * ldx]; yb = two ? lds[b1o + (FR - 2) * ldx] : 0.f;
      if (PHASE_ON(2048)) { s0 += lds[co + (FR - 4) * FLDH]; s1 += lds[co + (FR - 3) * FLDH]; }
      DW1_STEP(x0, x1, y0, y1)
      if (PHASE_ON(2048)) { s2 += lds[co + (FR - 2) * FLDH]; s3 += lds[co + (FR - 1) * FLDH]; }
      DW1_STEP(xa, xb, ya, yb)
#undef DW1_STEP
      gb1 += (s0 + s1) + (s2 + s3);
    } else if (PHASE_ON(2048)) {
      gb1 += column_sum(L::H1, tid);
    }
    STAMP(20)
    __syncthreads();  // X / h1 / h2 are rewritten by the next tile
    STAMP(21)
  }

  const int tid = tid0, lane = tid & 63;
  // ---- store this workgroup's partial gradients to its slab ----
  STAMP(22)
  STAMP_FLUSH()
  if (PHASE_ON(16384) || slab_id < 2) {  // (ablation bit 16384: only two workgroups write their slabs)
    asm volatile("s_nop 15\n\ts_nop 3");  // last asm MFMA's D -> v_accvgpr_read (16-pass XDL)
    {  // dW1 / dW3 tiles, fragment order: [w][tile][quad][lane] x 16 B
      const unsigned s1 = (unsigned)(wave * 4 * 4 * 64 + lane) * 16u, s3 = (unsigned)(wave * 2 * 4 * 64 + lane) * 16u;
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gW1a[4 * qd + e];
        stg16(slab_w1, s1 + (0 * 4 + qd) * 1024u, v);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gW1b[4 * qd + e];
        stg16(slab_w1, s1 + (2 * 4 + qd) * 1024u, v);
        if (DP > 32) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = gW1c[4 * qd + e];
          stg16(slab_w1, s1 + (1 * 4 + qd) * 1024u, v);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = gW1d[4 * qd + e];
          stg16(slab_w1, s1 + (3 * 4 + qd) * 1024u, v);
        }
        if constexpr (!H16) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = gW3a[4 * qd + e];
          stg16(slab_w3, s3 + qd * 1024u, v);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = gW3b[4 * qd + e];
          stg16(slab_w3, s3 + (4 + qd) * 1024u, v);
        }
      }
      if constexpr (H16) {  // [w][column tile b][lane] x 16 B
        const unsigned sh = (unsigned)(wave * 4 * 64 + lane) * 16u;
        stg16(slab_w3, sh, gW3h0);
        stg16(slab_w3, sh + 1024u, gW3h1);
        stg16(slab_w3, sh + 2048u, gW3h2);
        stg16(slab_w3, sh + 3072u, gW3h3);
      }
    }
    float* w2base = slab + slab_off_w2();
    const unsigned s2 = (unsigned)(wave * 16 * 4 * 64 + lane) * 16u;  // [w][t][quad][lane] x 16 B
#pragma unroll
    for (int t = 0; t < 16; ++t) {
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gW2[t][4 * qd + e];
        stg16(w2base, s2 + (unsigned)(t * 4 + qd) * 1024u, v);
      }
      __builtin_amdgcn_sched_barrier(0);  // keep at most one tile of read-backs live
    }
    slab[slab_off_b2(DP) + tid] = gb2;
    slab[slab_off_b1(DP) + tid] = gb1;
    if (tid < 32) {  // fixed-order sum of the four per-wave partials (the loop-end barrier made them visible)
      float b3s = 0.f, lss = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        b3s += lds[L::GACC + w * 64 + tid];
        lss += lds[L::GACC + w * 64 + 32 + tid];
      }
      slab[slab_off_b3(DP) + tid] = b3s;
      slab[slab_off_ls(DP) + tid] = lss;
    }
  }
  {  // loss statistics: per-wave sums -> LDS -> fixed-order workgroup sum -> slab (k_slab_reduce adds the slabs up;
     // contended float atomics on four addresses cost ~25 us per launch and made the logged losses order dependent)
    const float t0 = wave_sum(s_pl), t1 = wave_sum(s_vl), t2 = wave_sum(s_kl), t3 = wave_sum(s_cf);
    if (lane == 0) {
      lds[L::STAT + wave * 4 + 0] = t0;
      lds[L::STAT + wave * 4 + 1] = t1;
      lds[L::STAT + wave * 4 + 2] = t2;
      lds[L::STAT + wave * 4 + 3] = t3;
    }
    __syncthreads();
    if (tid < 4)
      slab[slab_off_st(DP) + tid] = (lds[L::STAT + tid] + lds[L::STAT + 4 + tid]) + (lds[L::STAT + 8 + tid] + lds[L::STAT + 12 + tid]);
  }
}

// ------------------------------------------------------------------------------------------------
// Deterministic slab reduction into the canonical gradient vector.
// grid.x covers the P canonical elements; each thread sums its element over the slabs of its network.
// ------------------------------------------------------------------------------------------------
struct SlabReduceArgs {
  const float* slabs; int slab_floats; int nslabs;  // nslabs = gridDim of the train kernel (even: pi, odd: vf)
  float* grads; int P;
  int offs[14];   // canonical offsets
  int D, Dp, A;
  int h16;        // dW3 region written by the 16x16x4 variant (k_fused_train<DP, true>)
  float ent_coef, b_local, inv_bg;
  float* sums;    // sums[4] = rows (for the stats finaliser)
  double* rec_sum; int* rec_t;  // optional [2][gridDim.x][kNormRec]: sums of squares of what each block produced, by tensor
};

// ------------------------------------------------------------------------------------------------
// clip_grad_norm_ needs the sum of squares of every gradient tensor.  Inside mobrob_ppo_train (single rank, nothing
// touches the gradient between reduction and clip) the reduction kernels produce them on the fly instead of a
// separate k_sqnorm_chunks launch: every 256-thread block records (tensor, sum of squares in float64) of the
// entries it produced -- one record per block, except the last block of a network, which holds the small tensors
// (<= kNormRec of them).  The table layout is static, so the host lists for each tensor which records to fold, in
// order (norm_fold_table), and k_adam_pack folds them like it folds chunk partials.
// ------------------------------------------------------------------------------------------------
constexpr int kNormRec = 4;
template <class PI>   // const int*, or the same behind a kernel-argument reference (constant address space)
__host__ __device__ inline int tensor_of_canonical(PI offs, int P, int dst) {
  if (dst < 0 || dst >= P) return -1;
  int t = 0;
  for (int k = 1; k < 13; ++k) t += (dst >= offs[k]) ? 1 : 0;
  return t;
}
// records of one wave in lane order: first occurrence of a tensor opens a record (<= kNormRec distinct tensors)
template <bool COH = false>
__device__ __forceinline__ void block_norm_records(int t, float val, double* rec_sum, int* rec_t) {
  __shared__ double sc[4 * kNormRec];
  __shared__ int sct[4 * kNormRec];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const double sq = t >= 0 ? (double)val * (double)val : 0.0;
  int nrec = 0;
  unsigned long long todo = __ballot(t >= 0);
  while (todo != 0ull) {
    const int first = __ffsll((long long)todo) - 1;
    const int tsel = __shfl(t, first, 64);
    const bool m = t == tsel;
    const double sum = wave_sum_d(m ? sq : 0.0);
    if (lane == 0 && nrec < kNormRec) { sc[wave * kNormRec + nrec] = sum; sct[wave * kNormRec + nrec] = tsel; }
    ++nrec;
    todo &= ~__ballot(m);
  }
  if (lane == 0)
    for (int k = nrec; k < kNormRec; ++k) sct[wave * kNormRec + k] = -1;
  __syncthreads();
  if (threadIdx.x == 0) {
    int n = 0, cur = -1;
    double acc = 0.0;
    for (int k = 0; k < 4 * kNormRec; ++k) {
      const int tt = sct[k];
      if (tt < 0) continue;
      if (tt == cur) { acc += sc[k]; continue; }
      if (cur >= 0 && n < kNormRec) { stc<COH>(rec_sum + n, acc); stc<COH>(rec_t + n, cur); ++n; }
      cur = tt;
      acc = sc[k];
    }
    if (cur >= 0 && n < kNormRec) { stc<COH>(rec_sum + n, acc); stc<COH>(rec_t + n, cur); ++n; }
    for (; n < kNormRec; ++n) stc<COH>(rec_t + n, -1);
  }
}

// slab position -> canonical gradient index of network `net` (or -1 for padding / unused positions)
__host__ __device__ __forceinline__ int slab_to_canonical(const SlabReduceArgs& s, int net, int p) {
  const int T_W1 = net == 0 ? 1 : 5, T_B1 = net == 0 ? 2 : 6, T_W2 = net == 0 ? 3 : 7, T_B2 = net == 0 ? 4 : 8;
  const int T_W3 = net == 0 ? 9 : 11, T_B3 = net == 0 ? 10 : 12;
  const int head = net == 0 ? s.A : 1;
  auto frag = [](int q, int nt, int* w, int* t, int* i, int* lane) {
    *lane = (q >> 2) & 63;
    *i = (q & 3) + 4 * ((q >> 8) & 3);
    const int wt = q >> 10;
    *t = wt % nt;
    *w = wt / nt;
  };
  int w, t, i, lane;
  if (p < slab_off_w1()) {  // dW2
    frag(p, 16, &w, &t, &i, &lane);
    const int n = 64 * w + 32 * (t >> 3) + crc(i) + 4 * (lane >> 5), j = 32 * (t & 7) + (lane & 31);
    return s.offs[T_W2] + n * FH + j;
  }
  if (p < slab_off_w3(s.Dp)) {  // dW1
    frag(p - slab_off_w1(), 4, &w, &t, &i, &lane);
    const int n = 64 * w + 32 * (t >> 1) + crc(i) + 4 * (lane >> 5), j = 32 * (t & 1) + (lane & 31);
    return j < s.D ? s.offs[T_W1] + n * s.D + j : -1;
  }
  if (p < slab_off_b2(s.Dp) && s.h16) {  // dW3, 16x16 tiles: [w][b][lane][e] -> head row 4(lane>>4)+e, column 64w+16b+(lane&15)
    const int q = p - slab_off_w3(s.Dp);
    if (q >= 16 * FH) return -1;
    const int e = q & 3, l = (q >> 2) & 63, b = (q >> 8) & 3, w_ = q >> 10;
    const int a_ = 4 * (l >> 4) + e, j = 64 * w_ + 16 * b + (l & 15);
    return a_ < head ? s.offs[T_W3] + a_ * FH + j : -1;
  }
  if (p < slab_off_b2(s.Dp)) {  // dW3
    frag(p - slab_off_w3(s.Dp), 2, &w, &t, &i, &lane);
    const int a_ = crc(i) + 4 * (lane >> 5), j = 64 * w + 32 * t + (lane & 31);
    return a_ < head ? s.offs[T_W3] + a_ * FH + j : -1;
  }
  if (p < slab_off_b1(s.Dp)) return s.offs[T_B2] + (p - slab_off_b2(s.Dp));
  if (p < slab_off_b3(s.Dp)) return s.offs[T_B1] + (p - slab_off_b1(s.Dp));
  if (p < slab_off_ls(s.Dp)) {
    const int k = p - slab_off_b3(s.Dp);
    return k < head ? s.offs[T_B3] + k : -1;
  }
  if (p < slab_off_st(s.Dp)) {
    const int k = p - slab_off_ls(s.Dp);
    return (net == 0 && k < s.A) ? s.offs[0] + k : -1;
  }
  // loss sums live behind the gradient vector (sums = grads + P): policy slabs carry pl / kl / clip count, value slabs vl
  const int k = p - slab_off_st(s.Dp);
  const bool mine = net == 0 ? (k == 0 || k == 2 || k == 3) : k == 1;
  return mine ? s.P + k : -1;
}

#ifndef MOBROB_SLAB_REDUCE_UNROLL
#define MOBROB_SLAB_REDUCE_UNROLL 8
#endif
constexpr int kSlabReduceUnroll = MOBROB_SLAB_REDUCE_UNROLL;
// grid = (ceil(slab_floats / 256), 2 networks): thread p sums slab position p over the slabs of its network
// (coalesced reads, fixed order) and scatters the total to the canonical gradient vector.
__global__ __launch_bounds__(256) void k_slab_reduce(SlabReduceArgs s) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const int net = blockIdx.y;
  if (p == 0 && net == 0) s.sums[4] = s.b_local;
  const int dst = p < s.slab_floats ? slab_to_canonical(s, net, p) : -1;
  float acc = 0.f;
  if (dst >= 0) {
    const float* src = s.slabs + (size_t)net * s.slab_floats + p;
    const size_t stride = 2 * (size_t)s.slab_floats;
    const int n = (s.nslabs - net + 1) / 2;
    // Eight loads in flight per lane (A/B on one box, 93 MB per launch: 4-deep 20.4 us, 8-deep 18.8, 16-deep 19.3, 32-deep
    // 20.3 per launch incl. its event bracket; the kernel streams at ~5.5 of the ~6.1 TB/s a sweep reaches on this part).
    // Slab w is added to accumulator w mod 8, the accumulators by a fixed tree: one summation order whatever the grid.
    constexpr int U = kSlabReduceUnroll;
    float ac[U];
#pragma unroll
    for (int k = 0; k < U; ++k) ac[k] = 0.f;
    int w = 0;
    for (; w + U <= n; w += U) {
      float v[U];
#pragma unroll
      for (int k = 0; k < U; ++k) v[k] = src[(size_t)(w + k) * stride];
#pragma unroll
      for (int k = 0; k < U; ++k) ac[k] += v[k];
    }
#pragma unroll
    for (int k = 0; k < U; ++k)
      if (w + k < n) ac[k] += src[(size_t)(w + k) * stride];
#pragma unroll
    for (int d = U / 2; d >= 1; d >>= 1)
#pragma unroll
      for (int k = 0; k < d; ++k) ac[k] += ac[k + d];
    acc = ac[0];
    if (dst < s.offs[1]) acc = __fadd_rn(acc, __fmul_rn(__fmul_rn(s.ent_coef, -s.b_local), s.inv_bg));  // entropy bonus gradient on log_std
    s.grads[dst] = acc;
  }
  if (s.rec_sum != nullptr) {
    const size_t b = ((size_t)net * gridDim.x + blockIdx.x) * kNormRec;
    block_norm_records(tensor_of_canonical(s.offs, s.P, dst), acc, s.rec_sum + b, s.rec_t + b);
  }
}

// ------------------------------------------------------------------------------------------------
// Weight packing: canonical row-major W[N][K] (ld) -> MFMA fragment order.
//   fwd pack  Pf[nb][kg][lane][s] = W[nb*32 + r][kg*8 + 4h + s]      (B[k][n] = W[n][k])
//   bwd pack  Pb[jb][kg][lane][s] = W[kg*8 + 4h + s][jb*32 + r]      (B[k][j] = W[k][j])
// rows/cols outside the source matrix are zero.
// ------------------------------------------------------------------------------------------------
__global__ void k_pack_fwd(const float* __restrict__ W, int N, int K, int ld, float* __restrict__ out, int NB,
                           int KG, float scale) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= NB * KG * 256) return;
  const int s = i & 3, lane = (i >> 2) & 63, kg = (i >> 8) % KG, nb = (i >> 8) / KG;
  const int n = nb * 32 + (lane & 31), k = kg * 8 + 4 * (lane >> 5) + s;
  out[i] = (n < N && k < K) ? scale * W[(size_t)n * ld + k] : 0.f;
}
// 16x16x4 fwd pack of a head [N <= 16][K]: out[kg][lane][s] = W[lane & 15][16 kg + 4 (lane >> 4) + s]
__host__ __device__ inline int pack_h16_idx(int n, int k) { return (((k >> 4) * 64) + 16 * ((k >> 2) & 3) + n) * 4 + (k & 3); }
__global__ void k_pack_h16(const float* __restrict__ W, int N, int K, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (K / 16) * 256) return;
  const int s = i & 3, lane = (i >> 2) & 63, kg = i >> 8;
  const int n = lane & 15, k = 16 * kg + 4 * (lane >> 4) + s;
  out[i] = n < N ? W[(size_t)n * K + k] : 0.f;
}
__global__ void k_scale_copy(const float* __restrict__ src, float* __restrict__ dst, int n, float scale) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = scale * src[i];
}
__global__ void k_pack_bwd(const float* __restrict__ W, int N, int K, int ld, float* __restrict__ out, int JB,
                           int KG) {
  // W is [N rows = k index][K cols = j index]
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= JB * KG * 256) return;
  const int s = i & 3, lane = (i >> 2) & 63, kg = (i >> 8) % KG, jb = (i >> 8) / KG;
  const int k = kg * 8 + 4 * (lane >> 5) + s, j = jb * 32 + (lane & 31);
  out[i] = (k < N && j < K) ? W[(size_t)k * ld + j] : 0.f;
}

// ------------------------------------------------------------------------------------------------
// Rollout-time fused forward (+ sampling epilogue for the policy network).
// grid = 2 * ceil(rows / 64); block b: net = b & 1, tile = b >> 1.
// ------------------------------------------------------------------------------------------------
struct FusedActArgs {
  FusedNet net[2];
  const float* X; int Dp; int rows;
  int row0;                // env index of row 0 (sampling noise is a function of the env index)
  int want_pi, want_v;
  float* mu; int ldmu;     // optional raw mean output [rows][ldmu]
  float* v;                // [rows]
  // sampling (policy net) -- any of the outputs may be null
  int sample; int A;
  const float* log_std; const float* eps; uint64_t seed; uint32_t draw; const uint32_t* draw_base; float lo, hi;
  float* act_raw; float* act_clip; float* logp;
};

// 32-row variant of the packed GEMM: one row block, two column blocks per wave (8 MFMAs per k-group)
#define MFMA_KG1(u, p, q)                            \
  _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) { \
    c0 = MFMA32(u[s_], p[s_], c0);                   \
    c1 = MFMA32(u[s_], q[s_], c1);                   \
  }
template <int LDA>
__device__ __forceinline__ void gemm_lds_packed_r32(int a_off, const f32x4* __restrict__ Bp0,
                                                    const f32x4* __restrict__ Bp1, int nkg, f32x16& c0, f32x16& c1,
                                                    int lane) {
  const int r = lane & 31, h = lane >> 5;
  const int ab = 4 * opaque((a_off + r * LDA + 4 * h) >> 2);
  unsigned bo = opaque_u((unsigned)lane * 16u);
  f32x4 pA = ldg16(Bp0, bo), qA = ldg16(Bp1, bo), pB, qB;
  f32x4 uA = *reinterpret_cast<const f32x4*>(&lds[ab]), uB;
  int ao = ab;
#pragma unroll 1
  for (int kg = 0; kg < nkg - 2; kg += 2) {
    pB = ldg16(Bp0, bo + 1024u);
    qB = ldg16(Bp1, bo + 1024u);
    uB = *reinterpret_cast<const f32x4*>(&lds[ao + 8]);
    MFMA_KG1(uA, pA, qA)
    pA = ldg16(Bp0, bo + 2048u);
    qA = ldg16(Bp1, bo + 2048u);
    uA = *reinterpret_cast<const f32x4*>(&lds[ao + 16]);
    MFMA_KG1(uB, pB, qB)
    bo += 2048u;
    ao += 16;
  }
  pB = ldg16(Bp0, bo + 1024u);
  qB = ldg16(Bp1, bo + 1024u);
  uB = *reinterpret_cast<const f32x4*>(&lds[ao + 8]);
  MFMA_KG1(uA, pA, qA)
  MFMA_KG1(uB, pB, qB)
}

// Deep-pipelined 32-row variant for K = 256 (nkg % 4 == 0, nkg >= 8): one k-group is only 8 MFMAs (~0.2 us), far
// less than the L2 latency of the weight stream, so fragments are fetched three k-groups ahead (4-deep ring).
template <int LDA>
__device__ __forceinline__ void gemm_lds_packed_r32_deep(int a_off, const f32x4* __restrict__ Bp0,
                                                         const f32x4* __restrict__ Bp1, int nkg, f32x16& c0,
                                                         f32x16& c1, int lane) {
  const int r = lane & 31, h = lane >> 5;
  const int ab = 4 * opaque((a_off + r * LDA + 4 * h) >> 2);
  unsigned bo = opaque_u((unsigned)lane * 16u);
  f32x4 p0 = ldg16(Bp0, bo), q0 = ldg16(Bp1, bo);
  f32x4 p1 = ldg16(Bp0, bo + 1024u), q1 = ldg16(Bp1, bo + 1024u);
  f32x4 p2 = ldg16(Bp0, bo + 2048u), q2 = ldg16(Bp1, bo + 2048u);
  f32x4 p3, q3;
  f32x4 uA = *reinterpret_cast<const f32x4*>(&lds[ab]), uB;
  int ao = ab;
#pragma unroll 1
  for (int kg = 0; kg < nkg - 4; kg += 4) {
    p3 = ldg16(Bp0, bo + 3072u); q3 = ldg16(Bp1, bo + 3072u);
    uB = *reinterpret_cast<const f32x4*>(&lds[ao + 8]);
    MFMA_KG1(uA, p0, q0)
    p0 = ldg16(Bp0, bo + 4096u); q0 = ldg16(Bp1, bo + 4096u);
    uA = *reinterpret_cast<const f32x4*>(&lds[ao + 16]);
    MFMA_KG1(uB, p1, q1)
    p1 = ldg16(Bp0, bo + 5120u); q1 = ldg16(Bp1, bo + 5120u);
    uB = *reinterpret_cast<const f32x4*>(&lds[ao + 24]);
    MFMA_KG1(uA, p2, q2)
    p2 = ldg16(Bp0, bo + 6144u); q2 = ldg16(Bp1, bo + 6144u);
    uA = *reinterpret_cast<const f32x4*>(&lds[ao + 32]);
    MFMA_KG1(uB, p3, q3)
    bo += 4096u;
    ao += 32;
  }
  p3 = ldg16(Bp0, bo + 3072u); q3 = ldg16(Bp1, bo + 3072u);
  uB = *reinterpret_cast<const f32x4*>(&lds[ao + 8]);
  MFMA_KG1(uA, p0, q0)
  uA = *reinterpret_cast<const f32x4*>(&lds[ao + 16]);
  MFMA_KG1(uB, p1, q1)
  uB = *reinterpret_cast<const f32x4*>(&lds[ao + 24]);
  MFMA_KG1(uA, p2, q2)
  MFMA_KG1(uB, p3, q3)
}

// LDS carve-up of the 32-row rollout kernel
template <int DP>
struct Lay32 {
  static constexpr int R = 32;
  static constexpr int LDX = DP + 4;
  static constexpr int X = 0;
  static constexpr int H1 = X + R * LDX;
  static constexpr int H2 = H1 + R * FLDH;
  static constexpr int DO = H2 + R * FLDH;   // [4 K-slices][32][FLDO] partial head tiles
  static constexpr int END = DO + 4 * R * FLDO;
};

// grid = 2 * ceil(rows / 32); block b: net = b & 1, tile = b >> 1.  ~76 KB of LDS -> two blocks per CU.
template <int DP>
__global__ __launch_bounds__(FTHREADS, 2) void k_fused_act(FusedActArgs a) {
  using L = Lay32<DP>;
  constexpr int ldx = L::LDX, per = DP / 4, R = 32;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int net = blockIdx.x & 1, tile = blockIdx.x >> 1;
  if ((net == 0 && !a.want_pi) || (net == 1 && !a.want_v)) return;
  const FusedNet W = a.net[net];
  const int row0 = tile * R;
#pragma unroll
  for (int i = tid; i < R * per; i += FTHREADS) {
    const int rr = i / per, c = i - rr * per;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row0 + rr < a.rows) v = ldg16(a.X, (unsigned)(row0 + rr) * (unsigned)(DP * 4) + (unsigned)(c * 16));
    *reinterpret_cast<f32x4*>(&lds[L::X + rr * ldx + 4 * c]) = v;
  }
  __syncthreads();
  {  // layer 1
    f32x16 c0 = splat16(W.b1s[64 * wave + r]), c1 = splat16(W.b1s[64 * wave + 32 + r]);
    constexpr int nkg = DP / 8;
    gemm_lds_packed_r32<ldx>(L::X, W.W1f + (size_t)(2 * wave) * nkg * 64, W.W1f + (size_t)(2 * wave + 1) * nkg * 64, nkg,
                             c0, c1, lane);
    const int o = opaque(L::H1 + 4 * h * FLDH + 64 * wave + r);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      lds[o + crc(i) * FLDH] = fast_tanh_scaled(c0[i]);
      lds[o + crc(i) * FLDH + 32] = fast_tanh_scaled(c1[i]);
    }
  }
  __syncthreads();
  {  // layer 2
    f32x16 c0 = splat16(W.b2s[64 * wave + r]), c1 = splat16(W.b2s[64 * wave + 32 + r]);
    constexpr int nkg = FH / 8;
    gemm_lds_packed_r32_deep<FLDH>(L::H1, W.W2f + (size_t)(2 * wave) * nkg * 64,
                                   W.W2f + (size_t)(2 * wave + 1) * nkg * 64, nkg, c0, c1, lane);
    const int o = opaque(L::H2 + 4 * h * FLDH + 64 * wave + r);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      lds[o + crc(i) * FLDH] = fast_tanh_scaled(c0[i]);
      lds[o + crc(i) * FLDH + 32] = fast_tanh_scaled(c1[i]);
    }
  }
  __syncthreads();
  {  // head: K split over the 4 waves (64 each); partial tiles side by side, summed in the epilogue
    f32x16 acc = zero16(), acc2 = zero16();
    const int ab = 4 * opaque((L::H2 + r * FLDH + wave * 64 + 4 * h) >> 2);
    const f32x4* bp = W.W3f + (size_t)(wave * 8) * 64;
    const unsigned bo = opaque_u((unsigned)lane * 16u);
#pragma unroll
    for (int kg = 0; kg < 8; kg += 2) {
      const f32x4 b0 = ldg16(bp, bo + kg * 1024u), b1 = ldg16(bp, bo + (kg + 1) * 1024u);
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(&lds[ab + kg * 8]);
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(&lds[ab + kg * 8 + 8]);
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) {
        acc = MFMA32(a0[s_], b0[s_], acc);
        acc2 = MFMA32(a1[s_], b1[s_], acc2);
      }
    }
    const int o = opaque(L::DO + wave * R * FLDO + 4 * h * FLDO + r);
#pragma unroll
    for (int i = 0; i < 16; ++i) lds[o + crc(i) * FLDO] = acc[i] + acc2[i];
  }
  __syncthreads();
  if (wave != 0 || lane >= R) return;
  const int row = row0 + lane;
  if (row >= a.rows) return;
  const int db = opaque(L::DO + lane * FLDO);
  auto head = [&](int k) {  // fixed-order sum of the four K-slices + bias
    return ((lds[db + k] + lds[db + R * FLDO + k]) + (lds[db + 2 * R * FLDO + k] + lds[db + 3 * R * FLDO + k])) + W.b3[k];
  };
  if (net == 1) {
    a.v[row] = head(0);
    return;
  }
  float lp = 0.f;
  float z0 = 0.f, z1 = 0.f, z2 = 0.f, z3 = 0.f;
  for (int k = 0; k < a.A; ++k) {
    const float m = head(k);
    if (a.mu) a.mu[(size_t)row * a.ldmu + k] = m;
    if (a.sample) {
      float e;
      if (a.eps != nullptr) {
        e = a.eps[(size_t)row * a.A + k];
      } else {
        if ((k & 3) == 0) {
          float z[4];
          box_muller4(philox4x32_10((uint32_t)(row + a.row0), (uint32_t)(k >> 2), a.draw + (a.draw_base ? *a.draw_base : 0u),
                                    0x45505331u, (uint32_t)a.seed,
                                    (uint32_t)(a.seed >> 32)), z);
          z0 = z[0]; z1 = z[1]; z2 = z[2]; z3 = z[3];
        }
        const int q = k & 3;
        e = q == 0 ? z0 : (q == 1 ? z1 : (q == 2 ? z2 : z3));
      }
      const float sd = expf(a.log_std[k]);
      const float act = m + e * sd;
      const float d = act - m;
      lp += -(d * d) / (2.0f * (sd * sd)) - logf(sd) - 0.91893853320467274178f;
      if (a.act_raw) a.act_raw[(size_t)row * a.A + k] = act;
      if (a.act_clip) a.act_clip[(size_t)row * a.A + k] = fminf(fmaxf(act, a.lo), a.hi);
    }
  }
  if (a.sample && a.logp) a.logp[row] = lp;
}

// ------------------------------------------------------------------------------------------------
// clip_grad_norm_ + Adam + re-packing in three small launches [torch 2.0.1 semantics; oracle clip_grad_norm /
// adam_step]:
//   k_sqnorm_chunks : per-(tensor, chunk) sum of squares in f64 (many blocks; chunk table built on the host)
//   k_adam_pack     : every block re-derives the clip coefficient from the chunk partials (fixed order), then
//                     each thread updates one canonical parameter and scatters it into the zero-padded copies
//                     (generic path) and the MFMA-fragment packs (fused path) -- no separate pack launches.
// ------------------------------------------------------------------------------------------------
struct NormChunk { int tensor, start, end, pad; };

struct StatsArgs {
  float* stats_row; const float* loss_sums; const float* log_std; float ent_coef, vf_coef, inv_bg; int n_act;
  int sde;   // gSDE (generic chain only): the entropy is state dependent, its sum over the minibatch is loss_sums[5]
};
// logged loss statistics of one optimizer step from the loss sums behind the gradient vector
// sum over the actions of 0.5 + log sqrt(2 pi) + log sd [SB3 DiagGaussianDistribution.entropy], in action order.  COH: log_std was written
// by other workgroups of the same launch (k_epoch64).  The loads go out eight at a time (as relaxed atomics they are not hoisted out of a
// load - use - load - use loop: twelve dependent round trips on one lane).
template <bool COH = false, class PF = const float*>
__device__ __forceinline__ float entropy_of_log_std(PF log_std, int n_act) {
  float ent = 0.f;
  for (int k0 = 0; k0 < n_act; k0 += 8) {
    float x[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) x[u] = ldc<COH>(log_std + (k0 + u < n_act ? k0 + u : 0));
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (k0 + u < n_act) ent += (0.5f + 0.91893853320467274178f) + logf(expf(x[u]));
  }
  return ent;
}
// the logged statistics of a step from its loss sums and the entropy term of the PRE-update log_std (`ent`, entropy_of_log_std)
template <bool COH = false>
__device__ __forceinline__ void stats_row_from_sums(const StatsArgs& st, float ent) {
  const float s0 = ldc<COH>(st.loss_sums + 0), s1 = ldc<COH>(st.loss_sums + 1), s2 = ldc<COH>(st.loss_sums + 2), s3 = ldc<COH>(st.loss_sums + 3),
              s4 = ldc<COH>(st.loss_sums + 4);
  const float pl = -s0 * st.inv_bg;
  const float vl = s1 * st.inv_bg;
  const float el = st.sde ? -ldc<COH>(st.loss_sums + 5) * st.inv_bg : -(ent * s4) * st.inv_bg;
  st.stats_row[0] = pl; st.stats_row[1] = vl; st.stats_row[2] = el;
  st.stats_row[3] = __fmaf_rn(st.vf_coef, vl, __fmaf_rn(st.ent_coef, el, pl));   // (explicit: the same bits from every kernel this is compiled into)
  st.stats_row[4] = s2 * st.inv_bg;
  st.stats_row[5] = s3 * st.inv_bg;
  st.stats_row[7] = 0.f;
}
// thread `tid` of 256: its share of the sum of squares of one chunk (float64).  Shared by k_sqnorm_chunks and the
// persistent small-batch kernel (kernels_train_small.h) so that both sum in exactly the same order.
__device__ __forceinline__ double chunk_sumsq_thread(const float* __restrict__ g, const NormChunk c, int tid) {
  double a = 0.0;
  for (int i = c.start + tid; i < c.end; i += 256) {
    const double x = (double)g[i];
    a += x * x;
  }
  return a;
}
__global__ __launch_bounds__(256) void k_sqnorm_chunks(const float* __restrict__ g, const NormChunk* __restrict__ chunks,
                                                       double* __restrict__ partial, StatsArgs st) {
  __shared__ double sc[16];
  if (blockIdx.x == 0 && threadIdx.x == 0 && st.stats_row != nullptr) stats_row_from_sums(st, entropy_of_log_std(st.log_std, st.n_act));  // pre-update log_std
  const double tot = block_sum_d(chunk_sumsq_thread(g, chunks[blockIdx.x], threadIdx.x), sc);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// tensors of a policy: log_std, (W, b) of one to kMaxHidden (device_utils.h) hidden layers per network, the two heads (engine.hip
// keeps the table).  kFusedTensors: what at most three layers per network make -- the range the per-element search below unrolls.
constexpr int kMaxTensors = 1 + 4 * kMaxHidden + 4;   // 37
constexpr int kFusedTensors = 17;
struct AdamPackArgs {
  float* p; const float* g; float* m; float* v; int P;
  float* g_out;  // the same vector, writable (persistent small-batch kernel: it produces the gradient itself)
  const NormChunk* chunks; const double* partial; int nchunks;
  const int* fold_idx; int fold_start[kMaxTensors + 1];  // non-null: `partial` is a record table; tensor t folds partial[fold_idx[k]], k in [fold_start[t], fold_start[t+1])
  float max_norm, step_size, bc2_sqrt, beta1, beta2, eps;
  int offs[kMaxTensors + 1];    // canonical offsets of the engine's ntens tensors (9 .. 37: one to eight hidden layers per network), padded with P
  int ntens;       // 9 .. 37
  int two_by_two;  // two hidden layers in BOTH networks: tensor numbering 0 .. 12, the one the fused cases below are written for
                   // (1 + 3 layers also make thirteen tensors)
  int id_pw1, id_vw1, id_aw, id_vw;   // tensors with a zero-padded compute copy (first-layer weights, heads), whatever the depth
  int HL, GL;      // width of the last hidden layer of the policy / value network (row length of the heads)
  int D, Dp, A, Ap, H1, H2, G1, G2;
  // generic-path padded copies (always maintained: cheap, and k_value_flagged/predict fallbacks use canonical)
  float* pW1p; float* vW1p; float* aWp; float* vWp;
  // fused-path packs (null when the fused path is disabled)
  float* fW1f[2]; float* fW2f[2]; float* fW3f[2]; float* fW3h[2]; float* fW2b[2]; float* fW3b[2]; float* fb1s[2]; float* fb2s[2];
  unsigned short* xW1[2]; unsigned short* xW2[2]; unsigned short* xW2b[2];  // x3 packs kept current per step (null: not maintained here)
  unsigned short* cW1[2]; unsigned short* cW2[2]; unsigned short* cW2b[2]; float* cW3[2]; float* cW3b[2];  // chain packs (null: not maintained)
  float* stats_row;  // [6] <- total gradient norm
  float* loss_sums_zero;  // fused path: the 8 loss accumulators are re-zeroed here instead of by a memset launch
  StatsArgs st;           // st.stats_row != null: this kernel also writes the step's loss statistics (no k_sqnorm_chunks launch)
};

__device__ __forceinline__ int pack_fwd_idx(int n, int k, int KG) {
  return (((n >> 5) * KG + (k >> 3)) * 64 + (n & 31) + 32 * ((k >> 2) & 1)) * 4 + (k & 3);
}
__device__ __forceinline__ int pack_bwd_idx(int krow, int j, int KG) {
  return (((j >> 5) * KG + (krow >> 3)) * 64 + (j & 31) + 32 * ((krow >> 2) & 1)) * 4 + (krow & 3);
}

// element (column-block index n, k) of an x3 pack of KS k steps <- the three bf16 pieces of x (layout: k_pack_x3_multi)
__device__ __forceinline__ void x3_pack_store(unsigned short* out, int n, int k, int KS, float x) {
  // x arrives as a product (scale * parameter): without this the compiler contracts it into the first residual of the split
  // (fma(scale, p, -piece1)), i.e. splits the UNROUNDED product -- not the bits k_pack_x3_multi produces from the stored parameter
  asm volatile("" : "+v"(x));
  unsigned p1, p2, p3;
  x3_split2(x, 0.f, p1, p2, p3);
  const size_t base = ((size_t)((n >> 5) * KS + (k >> 4)) * 3) * 512 + (size_t)((n & 31) + 32 * ((k & 15) >> 3)) * 8 + (k & 7);
  out[base] = (unsigned short)(p1 & 0xffffu);
  out[base + 512] = (unsigned short)(p2 & 0xffffu);
  out[base + 1024] = (unsigned short)(p3 & 0xffffu);
}
// ---- chain packs (k_chain_train, kernels_chain.h; layouts stated and checked in tests/chain_model.py) ----
// k slot (lane group g, element j) of k step s carries
//   layer 1:       observation column 32 s + 8 g + j                      (natural order)
//   layer 2 / dh1: neuron 32 s + 16 (j >> 2) + 4 g + (j & 3)              (the order accumulator tiles hand their rows over in)
__host__ __device__ __forceinline__ void chain_kslot_of_neuron(int n, int* s, int* g, int* j) {
  *s = n >> 5;
  const int r = n & 31;
  *g = (r >> 2) & 3;
  *j = ((r >> 4) << 2) | (r & 3);
}
// bf16 index of piece 0 of element (row & 15, k slot (g, j)) of ring unit `unit` (3 KB each, in the order the kernel consumes
// them); pieces 1, 2 follow at + 512, + 1024
__host__ __device__ __forceinline__ size_t chain_pack_idx(int unit, int row, int g, int j) {
  return ((size_t)unit * 3) * 512 + (size_t)((row & 15) + 16 * g) * 8 + j;
}
// unit order of the three packs: layer 1 [neuron tile][k step]; layer 2 [k step][neuron tile]; dh1 [half of the tiles][k step][tile of the half]
__host__ __device__ __forceinline__ int chain_unit_w1(int tile, int s, int K1) { return tile * K1 + s; }
__host__ __device__ __forceinline__ int chain_unit_w2(int tile, int s) { return s * 16 + tile; }
__host__ __device__ __forceinline__ int chain_unit_w2b(int tile, int s) { return (tile >> 3) * 64 + s * 8 + (tile & 7); }
__device__ __forceinline__ void chain_pack_store(unsigned short* out, size_t base, float x) {
  asm volatile("" : "+v"(x));   // split the ROUNDED product (see x3_pack_store)
  unsigned p1, p2, p3;
  x3_split2(x, 0.f, p1, p2, p3);
  out[base] = (unsigned short)(p1 & 0xffffu);
  out[base + 512] = (unsigned short)(p2 & 0xffffu);
  out[base + 1024] = (unsigned short)(p3 & 0xffffu);
}
__device__ __forceinline__ void chain_store_w1(unsigned short* w1c, int n, int k, int K1, float scaled) {
  chain_pack_store(w1c, chain_pack_idx(chain_unit_w1(n >> 4, k >> 5, K1), n, (k >> 3) & 3, k & 7), scaled);
}
__device__ __forceinline__ void chain_store_w2(unsigned short* w2c, unsigned short* w2bc, int n, int k, float scaled, float raw) {
  int s, g, j;
  chain_kslot_of_neuron(k, &s, &g, &j);           // forward: A[row = n][k slot of input neuron k] = scale W2[n][k]
  chain_pack_store(w2c, chain_pack_idx(chain_unit_w2(n >> 4, s), n, g, j), scaled);
  chain_kslot_of_neuron(n, &s, &g, &j);           // dh1: A[row = k (input neuron)][k slot of output neuron n] = W2[n][k]
  chain_pack_store(w2bc, chain_pack_idx(chain_unit_w2b(k >> 4, s), k, g, j), raw);
}
// head [A <= 16][H]: forward pack [t][lane][i] = W3[lane & 15][16 t + 4 (lane >> 4) + i]; dh2 pack [t][lane][i] = W3[4 (lane >> 4) + i][16 t + (lane & 15)]
__host__ __device__ __forceinline__ int chain_head_fwd_idx(int a_, int k) { return ((k >> 4) * 64 + a_ + 16 * ((k >> 2) & 3)) * 4 + (k & 3); }
__host__ __device__ __forceinline__ int chain_head_bwd_idx(int a_, int k) { return ((k >> 4) * 64 + (k & 15) + 16 * (a_ >> 2)) * 4 + (a_ & 3); }

// clip + Adam of canonical parameter i and its scatter into the padded copies / fragment packs [torch 2.0.1
// single-tensor Adam; oracle adam_step].  Shared by k_adam_pack and the persistent small-batch kernel.
// the four operands already in registers (the caller batches its loads).  COH: parameters, moments and the 64-wide families' packs are
// read by other workgroups of the same launch (k_epoch64): stored at agent scope.  (The x3 / chain packs exist for 256-wide engines
// only, which never run that kernel: plain stores.)
// (step_size / bc2_sqrt: the per-step constants, passed beside `a`: k_epoch64 changes them per step and a modified local copy of the
//  argument struct, with its runtime-indexed arrays, would live in scratch memory)
template <bool COH = false, class TA = AdamPackArgs>   // TA: AdamPackArgs, or the same struct behind a kernel-argument reference (constant address space)
__device__ __forceinline__ void adam_pack_apply(const TA& a, int i, float graw, float m0, float v0, float p0, float coef, float step_size, float bc2_sqrt) {
  // Explicit roundings (no fp-contract): this function is compiled into several kernels (k_adam_pack, k_epoch64) that must produce the
  // SAME bits, and which of `m0 b1 + (1 - b1) g`'s two products the compiler fuses into an fma is its choice per instantiation.
  // The forms are torch's: exp_avg.mul_(b1).add_(g, alpha = 1 - b1); exp_avg_sq.mul_(b2).addcmul_(g, g, value = 1 - b2);
  // denom = exp_avg_sq.sqrt() / sqrt(bc2) + eps; param.addcdiv_(exp_avg, denom, value = -step_size).
  const float g = __fmul_rn(graw, coef);
  const float m = __fmaf_rn(g, 1.0f - a.beta1, __fmul_rn(m0, a.beta1));
  const float v = __fmaf_rn(__fmul_rn(g, g), 1.0f - a.beta2, __fmul_rn(v0, a.beta2));
  const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(v), bc2_sqrt), a.eps);
  const float pn = __fmaf_rn(-step_size, __fdiv_rn(m, denom), p0);
  int t = 0;
#pragma unroll
  for (int k = 1; k < kFusedTensors; ++k) t += (i >= a.offs[k]) ? 1 : 0;   // (offsets beyond the engine's tensors equal P)
  if (a.ntens > kFusedTensors)   // deeper generic networks only (uniform branch)
    for (int k = kFusedTensors; k < a.ntens; ++k) t += (i >= a.offs[k]) ? 1 : 0;
  const int e = i - a.offs[t];
  // Store policy inside a co-operative launch (COH): agent-scope (write-through, one fabric write per 4-byte store: what the barrier
  // behind this phase waits for) ONLY for what another workgroup reads before the launch ends -- the packs, and the canonical
  // log_std / head biases the gradient phase reads directly.  The moments and every other parameter are read back by THIS thread
  // only (past the vector L1: the line is in this XCD's L2), the zero-padded copies by later kernels: plain stores.
  a.m[i] = m;
  a.v[i] = v;
  if (COH && a.two_by_two && (t == 0 || t == 10 || t == 12)) stc<true>(a.p + i, pn);
  else a.p[i] = pn;
#ifdef ADAM_NO_PACK  // timing-only ablation: what the scattered pack stores cost
  return;
#endif
  if (!a.two_by_two) {  // other depths than two hidden layers: the generic GEMM chain's zero-padded copies only (no fused packs exist)
    if (t == a.id_pw1 || t == a.id_vw1) {
      const int n = e / a.D, k = e - n * a.D;
      (t == a.id_vw1 ? a.vW1p : a.pW1p)[n * a.Dp + k] = pn;
    } else if (t == a.id_aw || t == a.id_vw) {
      (t == a.id_vw ? a.vWp : a.aWp)[e] = pn;
    }
    return;
  }
  switch (t) {
    case 1: case 5: {  // W1 [H][D]
      const int net = t == 5, n = e / a.D, k = e - n * a.D;
      (net ? a.vW1p : a.pW1p)[n * a.Dp + k] = pn;
      if (a.fW1f[net]) stc<COH>(a.fW1f[net] + pack_fwd_idx(n, k, a.Dp / 8), kTanhScale * pn);
      if (a.xW1[net]) x3_pack_store(a.xW1[net], n, k, a.Dp / 16, kTanhScale * pn);
      if (a.cW1[net]) chain_store_w1(a.cW1[net], n, k, (a.Dp + 31) / 32, kTanhScale * pn);
    } break;
    case 3: case 7: {  // W2 [H2][H1]
      const int net = t == 7;
      if (a.fW2f[net]) {
        const int K = net ? a.G1 : a.H1, n = e / K, k = e - n * K;
        stc<COH>(a.fW2f[net] + pack_fwd_idx(n, k, K / 8), kTanhScale * pn);
        // (packs only a gradient kernel reads are null while another gradient kernel is the engine's: k_chain_train has its own)
        if (a.fW2b[net]) stc<COH>(a.fW2b[net] + pack_bwd_idx(n, k, (net ? a.G2 : a.H2) / 8), pn);
        if (a.xW2[net]) x3_pack_store(a.xW2[net], n, k, K / 16, kTanhScale * pn);                       // forward operand B[k][n] = scale W2[n][k]
        if (a.xW2b[net]) x3_pack_store(a.xW2b[net], k, n, (net ? a.G2 : a.H2) / 16, pn);                // backward operand B[n][k] = W2[n][k]
        if (a.cW2[net]) chain_store_w2(a.cW2[net], a.cW2b[net], n, k, kTanhScale * pn, pn);
      }
    } break;
    case 9: case 11: {  // head [A or 1][H2]
      const int net = t == 11, K = net ? a.G2 : a.H2, n = e / K, k = e - n * K;
      (net ? a.vWp : a.aWp)[e] = pn;
      if (a.fW3f[net]) {
        stc<COH>(a.fW3f[net] + pack_fwd_idx(n, k, K / 8), pn);
        if (a.fW3b[net]) stc<COH>(a.fW3b[net] + pack_bwd_idx(n, k, 4), pn);
        if (a.fW3h[net] && n < 16) stc<COH>(a.fW3h[net] + pack_h16_idx(n, k), pn);
        if (a.cW3[net] && n < 16) { a.cW3[net][chain_head_fwd_idx(n, k)] = pn; a.cW3b[net][chain_head_bwd_idx(n, k)] = pn; }
      }
    } break;
    case 2: case 6: if (a.fb1s[t == 6]) stc<COH>(a.fb1s[t == 6] + e, kTanhScale * pn); break;  // hidden biases (scaled copies)
    case 4: case 8: if (a.fb2s[t == 8]) stc<COH>(a.fb2s[t == 8] + e, kTanhScale * pn); break;
    default: break;
  }
}

// One 256-parameter block of clip + Adam + packs: block bx.  COH: gradient, norm records and loss sums come from other workgroups of
// the same launch, parameters / moments / packs go to them (k_epoch64): agent-scope accesses.
// (step_size / bc2_sqrt / stats_row / inv_bg: the per-step fields of `a` and `a.st`, passed beside it)
// The block's work with the element each thread updates CHOSEN BY THE CALLER: `i` = its canonical parameter (-1: none) and the four
// operands already requested (k_adam_pack: parameter bx 256 + thread, loaded here below; k_epoch64: the parameter whose gradient the
// thread has just reduced, still in its registers).  `lin` = bx 256 + thread: who does the block-0 duties (statistics, zeroing).
template <bool COH, class TA>
// `ent_pre` (used by the thread with lin == 0 only): entropy_of_log_std of the log_std this step's gradient was taken at -- the caller
// reads it BEFORE any thread of the launch can have updated log_std.
// `fidx_pre` (>= 0: a.fold_idx[threadIdx.x], fetched by the caller ahead of time; else fetched here): the index of the norm record this
// thread stages when the record table has at most 128 entries -- one dependent round trip less behind a grid barrier.
__device__ __forceinline__ void adam_pack_block_at(const TA& a, int lin, int i, float g_in, float m_in, float v_in, float p_in, float step_size,
                                                   float bc2_sqrt, float* stats_row, float inv_bg, float ent_pre, int fidx_pre = -1) {
  __shared__ double part[1024];
  __shared__ int tens[256];
  __shared__ float nts[kMaxTensors + 3];
  __shared__ float coef_s, total_s;
  if (lin == 0 && a.st.loss_sums != nullptr) {  // this kernel also writes the step's loss statistics (pre-update log_std)
    StatsArgs st;
    st.stats_row = stats_row; st.loss_sums = a.st.loss_sums; st.log_std = a.st.log_std;
    st.ent_coef = a.st.ent_coef; st.vf_coef = a.st.vf_coef; st.inv_bg = inv_bg; st.n_act = a.st.n_act; st.sde = a.st.sde;
    stats_row_from_sums<COH>(st, ent_pre);
  }
  if (a.fold_idx != nullptr) {  // norm records of the reduction kernel, listed per tensor by the host
    const int nrec = a.fold_start[13];   // (records exist with the fused kernels only: two hidden layers, 13 tensors)
    if (nrec <= 128) {  // 64-wide nets (~90 records): one lane per tensor, thirteen short serial folds side by side
      if ((int)threadIdx.x < nrec) part[threadIdx.x] = ldc<COH>(a.partial + (fidx_pre >= 0 ? fidx_pre : a.fold_idx[threadIdx.x]));   // (nrec <= 128 < blockDim)
      __syncthreads();
      if (threadIdx.x < 13) {
        double ts = 0.0;
        for (int c = a.fold_start[threadIdx.x]; c < a.fold_start[threadIdx.x + 1]; ++c) ts += part[c];
        nts[threadIdx.x] = (float)sqrt(ts);
      }
    } else {
      // 2 x 256 (712 records): wave w folds tensors w, w + 4, w + 8, w + 12 -- lane l adds records l, l + 64, ... of the
      // tensor's list in list order, then a fixed butterfly over the lanes: one fixed summation tree per tensor, and four
      // short wave reductions instead of a 256-long serial chain (what made this lose to k_sqnorm_chunks in round 2).
      // The records come through LDS: ONE round of global loads by all threads, then the four folds of a wave read LDS
      // (fetched inside the per-tensor loop, each tensor of a wave paid its own two dependent global round trips).
      const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
      const bool staged = nrec <= 1024;
      if (staged) {
        for (int c = threadIdx.x; c < nrec; c += blockDim.x) part[c] = ldc<COH>(a.partial + a.fold_idx[c]);
        __syncthreads();
      }
      for (int t = wv; t < 13; t += 4) {
        double ts = 0.0;
        if (staged)
          for (int c = a.fold_start[t] + lane; c < a.fold_start[t + 1]; c += 64) ts += part[c];
        else
          for (int c = a.fold_start[t] + lane; c < a.fold_start[t + 1]; c += 64) ts += ldc<COH>(a.partial + a.fold_idx[c]);
        ts = wave_sum_d(ts);
        if (lane == 0) nts[t] = (float)sqrt(ts);
      }
    }
  } else {
    for (int c = threadIdx.x; c < a.nchunks; c += blockDim.x) {  // one parallel round trip; the serial fold below is LDS only
      part[c] = ldc<COH>(a.partial + c);
      tens[c] = a.chunks[c].tensor;
    }
    __syncthreads();
    // total norm = norm of per-tensor norms (torch.norm(torch.stack(norms))).  One thread per tensor folds that
    // tensor's chunk partials in chunk order and takes the float64 square root (a software routine: thirteen of them
    // one after the other were half of this kernel), thread 0 then adds the thirteen squares in tensor order.
    if ((int)threadIdx.x < a.ntens) {
      double ts = 0.0;
      for (int c = 0; c < a.nchunks; ++c)
        if (tens[c] == (int)threadIdx.x) ts += part[c];
      nts[threadIdx.x] = (float)sqrt(ts);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot_sq = 0.f;
    for (int t = 0; t < a.ntens; ++t) tot_sq = __fmaf_rn(nts[t], nts[t], tot_sq);  // explicit: one rounding per tensor
    const float total = sqrtf(tot_sq);
    total_s = total;
    coef_s = fminf(a.max_norm / (total + 1e-6f), 1.0f);
  }
  __syncthreads();
  const float coef = coef_s;
  if (lin == 0 && stats_row != nullptr) stats_row[6] = total_s;
  if (lin < 8 && a.loss_sums_zero != nullptr) stc<COH>(a.loss_sums_zero + lin, 0.f);  // consumed by k_sqnorm_chunks; ready for the next step
  if (i < 0 || i >= a.P) return;
  adam_pack_apply<COH>(a, i, g_in, m_in, v_in, p_in, coef, step_size, bc2_sqrt);
}
template <bool COH, class TA>
__device__ __forceinline__ void adam_pack_block(const TA& a, int bx, float step_size, float bc2_sqrt, float* stats_row, float inv_bg) {
  // this thread's four operands are requested first: their memory latency runs under the norm fold
  const int i = bx * 256 + (int)threadIdx.x;
  float g_in = 0.f, m_in = 0.f, v_in = 0.f, p_in = 0.f;
  if (i < a.P) { g_in = ldc<COH>(a.g + i); m_in = ldc<COH>(a.m + i); v_in = ldc<COH>(a.v + i); p_in = ldc<COH>(a.p + i); }
  // (block 0 updates log_std itself, behind the barriers of the norm fold: its thread 0 reads the pre-update values here)
  const float ent = (i == 0 && a.st.loss_sums != nullptr) ? entropy_of_log_std<COH>(a.st.log_std, a.st.n_act) : 0.f;
  adam_pack_block_at<COH>(a, i, i, g_in, m_in, v_in, p_in, step_size, bc2_sqrt, stats_row, inv_bg, ent);
}
__global__ __launch_bounds__(256) void k_adam_pack(AdamPackArgs a) {
  adam_pack_block<false>(a, blockIdx.x, a.step_size, a.bc2_sqrt, a.stats_row, a.st.inv_bg);
}

// ------------------------------------------------------------------------------------------------
// host-side state of the fused path
// ------------------------------------------------------------------------------------------------
struct FusedState {
  bool enabled = false;
  int H = 0;  // hidden width served by the fused kernels: 256 (kernels_fused.h) or 64 (kernels_fused64.h)
  int D = 0, Dp = 0, A = 0;
  float* packed = nullptr;      // all packed weights
  size_t packed_floats = 0;
  FusedNet net[2];
  float* slabs = nullptr;
  float* train_rec = nullptr;   // H = 256: [T*N][train_rec_width(A)] (k_build_train_records)
  bool train_x3 = false;        // k_fused_train<.., X3 = true>: x3 packs refreshed after every optimizer step
  bool train_chain = false;     // k_chain_train (kernels_chain.h) instead: chain packs refreshed after every optimizer step
  size_t lds_chain_bytes = 0;
  unsigned long long* stamps = nullptr;  // diagnostic build only
  int slab_floats = 0, max_grid = 0;
  int pair_nseq_max = 0;  // 64-wide nets: two-wave workgroups per network of k_pair64_train (slabs are sized for them)
  size_t lds_bytes = 0, lds_act_bytes = 0;
};

inline size_t fused_lds_bytes(int Dp) { return (size_t)(FR * (Dp + 4) + 2 * FR * FLDH + FR * FLDO + 96 + 256 + 16) * sizeof(float); }

inline size_t fused_lds_act_bytes(int Dp) { return (size_t)(32 * (Dp + 4) + 2 * 32 * FLDH + 4 * 32 * FLDO) * sizeof(float); }

// Row stride of stored observations (floats): the fused kernels are instantiated for 16 / 32 / 48 / 64 columns, so every
// observation of up to 64 features is padded to the next of those (zeros); wider ones to a multiple of 8 (generic path).
inline int padded_obs_dim(int D) { return D <= 64 ? (D + 15) / 16 * 16 : (D + 7) / 8 * 8; }
inline bool fused_shape_ok(int D, int A, int H1, int H2, int G1, int G2) {
  const int Dp = padded_obs_dim(D);
  const bool same = H1 == H2 && H1 == G1 && H1 == G2 && (H1 == FH || H1 == 64);
  return same && A <= 32 && (Dp == 16 || Dp == 32 || Dp == 48 || Dp == 64);
}

// (re)build the packed weight copies from the canonical parameter vector
inline void fused_repack(FusedState& f, const float* params, const int* offs, hipStream_t st) {
  if (!f.enabled) return;
  const int Dp = f.Dp, D = f.D, A = f.A;
  auto fwd = [&](const float* W, int N, int K, int ld, const f32x4* out, int NB, int KG, float scale) {
    hipLaunchKernelGGL(k_pack_fwd, dim3((NB * KG * 256 + 255) / 256), dim3(256), 0, st, W, N, K, ld, (float*)out, NB, KG,
                       scale);
  };
  auto bwd = [&](const float* W, int N, int K, int ld, const f32x4* out, int JB, int KG) {
    hipLaunchKernelGGL(k_pack_bwd, dim3((JB * KG * 256 + 255) / 256), dim3(256), 0, st, W, N, K, ld, (float*)out, JB, KG);
  };
  // tensor ids: 0 log_std, 1 pW1, 2 pb1, 3 pW2, 4 pb2, 5 vW1, 6 vb1, 7 vW2, 8 vb2, 9 aW, 10 ab, 11 vW, 12 vb
  const int w1[2] = {1, 5}, w2[2] = {3, 7}, w3[2] = {9, 11}, heads[2] = {A, 1};
  const int H = f.H;
  for (int n = 0; n < 2; ++n) {
    fwd(params + offs[w1[n]], H, D, D, f.net[n].W1f, H / 32, Dp / 8, kTanhScale);
    fwd(params + offs[w2[n]], H, H, H, f.net[n].W2f, H / 32, H / 8, kTanhScale);
    fwd(params + offs[w3[n]], heads[n], H, H, f.net[n].W3f, 1, H / 8, 1.0f);
    if (heads[n] <= 16)  // 16x16x4 pack of narrow heads: k_fused_train<.., true> and k_pair64_train (NJ <= 6)
      hipLaunchKernelGGL(k_pack_h16, dim3(H / 16), dim3(256), 0, st, params + offs[w3[n]], heads[n], H, (float*)f.net[n].W3h);
    hipLaunchKernelGGL(k_scale_copy, dim3(1), dim3(256), 0, st, params + offs[w1[n] + 1], (float*)f.net[n].b1s, H, kTanhScale);
    hipLaunchKernelGGL(k_scale_copy, dim3(1), dim3(256), 0, st, params + offs[w2[n] + 1], (float*)f.net[n].b2s, H, kTanhScale);
    bwd(params + offs[w2[n]], H, H, H, f.net[n].W2b, H / 32, H / 8);
    bwd(params + offs[w3[n]], heads[n], H, H, f.net[n].W3b, H / 32, 4);
  }
}

inline void fused_launch_act(FusedState& f, FusedActArgs& a, hipStream_t st);

inline bool fused_forward(FusedState& f, const float* X, int rows, bool want_pi, float* mu_out, int ldmu, bool want_v,
                          float* v_out, hipStream_t st) {
  if (!f.enabled) return false;
  FusedActArgs a{};
  a.X = X; a.rows = rows; a.want_pi = want_pi; a.want_v = want_v; a.mu = mu_out; a.ldmu = ldmu; a.v = v_out;
  a.sample = 0; a.A = f.A;
  fused_launch_act(f, a, st);
  return true;
}

}  // namespace mobrob
