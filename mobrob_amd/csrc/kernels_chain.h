// k_chain_train: the gradient kernel of the 256-wide networks designed around the bf16 matrix pipe (round 4).
//
// k_fused_train<.., X3> (kernels_fused.h) is the f32 kernel with its GEMM loops swapped for "x3" loops (float32 products as six
// bf16 x bf16 MFMAs on three-way split operands): every wave owns 64 output columns, activations live in LDS as float32 and are
// split in the k loops, every weight fragment is streamed from L2 per 32 rows (64 B/clk per CU: the rate of the CU's vector
// memory port).  This kernel keeps the arithmetic (the same six products per float32 multiply-add, float32 accumulation, heads /
// loss / epilogues in float32) and changes who owns what:
//
//   * A wave owns 16 BATCH ROWS of a 64-row tile and runs the whole forward / backward ACTIVATION chain of its rows in
//     registers.  Every layer is computed transposed, C[neuron][row] = sum_k W[neuron][k] act[k][row] with the weights as the
//     MFMA A operand and the activations as the B operand (v_mfma_f32_16x16x32_bf16): an accumulator tile then has its neuron
//     in the registers / lane groups and its batch row on the lanes, which IS the B-operand layout of the next layer -- no
//     LDS round trip, no lane movement (cdna_hip_programming.md, 'An accumulator tile as the next MFMA's operand').  The price
//     is a permuted k order that the weight packs absorb (tests/chain_model.py states the maps and checks them on the CPU).
//     Activations are split into their three bf16 pieces ONCE, when a k step's B fragment is formed (44 VALU instructions per 96
//     MFMAs), not per use.
//   * All four waves need the same weights: they are streamed L2 -> LDS ONCE per tile by LDS-DMA (global_load_lds_dwordx4, no
//     registers) into a 72 KB ring and read by every wave with ds_read_b128: 16 B/clk per CU from L2 instead of 64.
//   * The weight gradients contract over the batch rows, i.e. over the LANES of the chain layout: the activations are
//     exchanged once through LDS as float32 "images" [column][64 rows] (rows contiguous: a lane's eight-row fragment is two
//     ds_read_b128 instead of eight ds_read_b32; XOR-swizzled, conflict-free for every access pattern below) and dW2 / dW1 /
//     dW3 run as in k_fused_train: dW2's 256 x 256 accumulators pinned in the 256 AGPRs of the four waves for the whole launch.
//     dW1 no longer holds 64 VGPRs for the launch: it is accumulated per tile and added to the workgroup's (L2-resident) slab.
//
// (Measured and not kept: a wave-PAIR ownership of the forward pass -- 32 rows x half the neurons per wave, every weight fragment
//  read from the ring feeding two B fragments, half the ring reads -- correct on the whole suite, 2 % slower: CHANGELOG.md, round 4;
//  the source is in git history, scratch/kernels_chain_pair.h at commit 72f49c6.)
// LDS (160 KB): h1 / dz1 image 64 KB | X image | constants | ring 12 KB + (h2 / dz2 image 64 KB: ring space while no image is live).
// Same slab format as k_fused_train: k_slab_reduce, the norm records and k_adam_pack are unchanged.
// Conditions: 256-wide tanh nets, heads <= 16 wide, observation rows padded to 16 / 32 / 64 columns (engine.hip fused_init).
#pragma once
#include <type_traits>
#include "kernels_fused.h"

namespace mobrob {

constexpr int CR = 64;                 // batch rows per tile (four waves x 16)
constexpr int CIMG = FH * CR;          // floats of a transposed activation image [256 columns][64 rows]
constexpr int CUNIT = 768;             // floats of a ring unit: one A fragment (16 neurons x 32 k slots) x 3 bf16 pieces = 3 KB
constexpr int CSEG = 8;                // units per ring segment (one workgroup barrier per segment)
constexpr int CNSEG = 3;               // segments in the ring: one being read, two in flight
constexpr int CSLOTS = CSEG * CNSEG;   // 24 units = 72 KB

// LDS carve-up (float offsets)
template <int DP>
struct CLay {
  static constexpr int H1 = 0;                  // h1, later dz1: [256][64] swizzled (img_addr)
  static constexpr int XI = H1 + CIMG;          // X: [DP][64] swizzled
  static constexpr int CST = XI + DP * CR;      // [3][32]: 1/var, log(sd)+log(sqrt(2pi)), head bias
  static constexpr int B1 = CST + 96;           // hidden biases x 2 log2(e) (the tanh epilogue's scale)
  static constexpr int B2 = B1 + FH;
  static constexpr int STAT = B2 + FH;          // [4 waves][4] loss sums (kernel end)
  static constexpr int GACC = STAT + 16;        // [4 waves][2][16] head-bias / log_std gradient sums (kernel end)
  static constexpr int RING = (GACC + 128 + 255) / 256 * 256;   // 1 KB aligned; the first 4 units are ring-only ...
  static constexpr int DO = RING;               // ... and hold the dout image [64 rows][16] between the two chain phases
  static constexpr int H2 = RING + 4 * CUNIT;   // h2, later dz2: [256][64] swizzled; units 4 .. 23 of the ring otherwise
  static constexpr int END = H2 + CIMG;
  static_assert(RING + CSLOTS * CUNIT <= END, "ring beyond the h2 image");
  static_assert(END * 4 <= 163840, "LDS");
};
inline size_t chain_lds_bytes(int Dp) { return Dp == 16 ? CLay<16>::END * 4 : Dp == 32 ? CLay<32>::END * 4 : CLay<64>::END * 4; }

// element (column m, batch row) of a transposed image: 16-byte chunk c of a column's 256 bytes sits at c ^ (m & 15)
__host__ __device__ __forceinline__ int img_addr(int m, int row) { return m * 64 + ((((row >> 2) ^ m) & 15) << 2) + (row & 3); }

#define MFMA16B(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), (c), 0, 0, 0)
// the six kept products of an x3 multiply-add, small terms first; W = weight fragment (A operand), X = activation fragment (B)
#define X3C_MFMA6(W_, X_, C_)                   \
  C_ = MFMA16B(W_.p[1], X_.p[1], C_);           \
  C_ = MFMA16B(W_.p[2], X_.p[0], C_);           \
  C_ = MFMA16B(W_.p[0], X_.p[2], C_);           \
  C_ = MFMA16B(W_.p[1], X_.p[0], C_);           \
  C_ = MFMA16B(W_.p[0], X_.p[1], C_);           \
  C_ = MFMA16B(W_.p[0], X_.p[0], C_);

// One ring unit: acc += W . X as six products, ordered so that the weight pieces die early -- piece 2 after the first MFMA, piece 1
// after the third -- and the NEXT unit's pieces are read into the freed registers right behind their last use (piece 0, needed
// to the end, alternates between two registers by unit parity).  Every read is then issued 3 .. 6 MFMAs (48 .. 96 cycles) before
// the MFMA that needs it: with all three pieces read in one burst late in the unit (the first version) every unit started with
// an LDS wait.  Magnitudes: (2,0) (1,1) ~2^-16, (1,0) ~2^-8, (0,2) ~2^-16, (0,1) ~2^-8, (0,0) ~1 of the product.
struct RingW { u32x4 p0[2], p1, p2; };
__device__ __forceinline__ u32x4 ring_read_piece(int ring_lane_f0, int slot, int pc) {
  return *reinterpret_cast<const u32x4*>(&lds[ring_lane_f0 + slot * CUNIT + 256 * pc]);
}
// (Scheduling barriers behind the second and third read -- the scheduler otherwise issues the three reads together behind the third
//  MFMA, 48 cycles before the next unit needs the first -- were measured twice (after every MFMA, and behind the two reads only): layer 2 19.0 k -> 19.8 k
//  cycles, launch +1.3 %.  The reads stay where the scheduler puts them.)
#define CHAIN_UNIT(Wr, par, X_, C_, have_next, nslot)                                        \
  if (have_next) Wr.p0[(par) ^ 1] = ring_read_piece(ringl, nslot, 0);                        \
  C_ = MFMA16B(Wr.p2, X_.p[0], C_);                                                          \
  if (have_next) Wr.p2 = ring_read_piece(ringl, nslot, 2);                                   \
  C_ = MFMA16B(Wr.p1, X_.p[1], C_);                                                          \
  C_ = MFMA16B(Wr.p1, X_.p[0], C_);                                                          \
  if (have_next) Wr.p1 = ring_read_piece(ringl, nslot, 1);                                   \
  C_ = MFMA16B(Wr.p0[par], X_.p[2], C_);                                                     \
  C_ = MFMA16B(Wr.p0[par], X_.p[1], C_);                                                     \
  C_ = MFMA16B(Wr.p0[par], X_.p[0], C_);

// LDS-DMA of 1 KB: lane l's 16 bytes at sbase + voff land at LDS byte address lds_byte + 16 l.  Issued as inline asm: the
// compiler knows nothing of it (no conservative vmcnt(0) in front of every ring read); the ring protocol below waits by hand.
// The six DMA instructions a wave owes a ring segment go out ONE PER UNIT behind the segment's barrier, not as a burst of six right
// behind it (an LDS-DMA costs its wave 60 - 180 cycles of issue inside a busy phase; spread, most of that falls into MFMA shadows):
// launch -1.9 .. -2.8 % on four of four boxes (profiles/r4/chain_l2.txt).  0: the burst form.
#ifndef MOBROB_CHAIN_SPREAD
#define MOBROB_CHAIN_SPREAD 1
#endif
#ifndef MOBROB_CHAIN_SKIP     // timing-only ablation builds (outputs wrong by construction; never in the product library)
#define MOBROB_CHAIN_SKIP 0
#endif
__device__ __forceinline__ void dma16(const void* sbase, unsigned voff, unsigned lds_byte) {
  if (MOBROB_CHAIN_SKIP & 1) return;
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_byte), "v"(voff), "s"(sbase) : "memory", "m0");
}
// one ring unit: the three pieces of one A fragment, 3 KB contiguous in the pack and in the ring.  The instruction offset of
// global_load_lds moves the global AND the LDS address (scratch/dma_offset_probe.hip, profiles/r4): one M0 write per unit.
__device__ __forceinline__ void dma_unit(const u32x4* src, int slot, unsigned lane16, int ring_f0) {
  if (MOBROB_CHAIN_SKIP & 1) return;
  const unsigned dst = (unsigned)(ring_f0 + slot * CUNIT) * 4u;
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %1, %2\n\t"
               "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
               "global_load_lds_dwordx4 %1, %2 offset:2048" ::"s"(dst), "v"(lane16), "s"(src) : "memory", "m0");
}
// one PIECE (1 KB) of a ring unit: for the spread issue (MOBROB_CHAIN_SPREAD), one DMA instruction per unit of the stream instead of six in
// a burst behind the segment barrier
__device__ __forceinline__ void dma_piece(const u32x4* src, int slot, int pc, unsigned lane16, int ring_f0) {
  if (MOBROB_CHAIN_SKIP & 1) return;
  const unsigned dst = (unsigned)(ring_f0 + slot * CUNIT + 256 * pc) * 4u;
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(lane16), "s"(src + 64 * pc) : "memory", "m0");
}
__device__ __forceinline__ X3Frag ring_read(int ring_lane_f0, int slot) {  // ring_lane_f0 = RING + 4 lane (opaque per-lane base)
  X3Frag f;
#pragma unroll
  for (int pc = 0; pc < 3; ++pc) f.p[pc] = *reinterpret_cast<const u32x4*>(&lds[ring_lane_f0 + slot * CUNIT + 256 * pc]);
  return f;
}
// A segment boundary of the weight ring, in front of the reads of segment Q (of NS segments of this stream): this wave's DMAs
// of segment Q have landed (the six of segment Q + 1 may still be in flight), its reads of segment Q - 1 are complete; after
// the barrier that holds for every wave, so segment Q may be read and segment Q - 1's slots refilled with segment Q + 2.
#if MOBROB_CHAIN_SKIP & 4
#define CHAIN_WAIT_DMA(more)
#elif defined(MOBROB_CHAIN_VMCNT0)   // validation builds: never rely on the count of DMAs in flight
#define CHAIN_WAIT_DMA(more) asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define CHAIN_WAIT_DMA(more) \
  if (more) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif
// (LDS_BARRIER: kernels_fused.h.  __syncthreads() here would wait for the next tile's gathers (HBM latency) and for the slab stores at
//  every barrier of the tile -- 2-3 k cycles each.)
#if MOBROB_CHAIN_SKIP & 2
#define CHAIN_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
#define CHAIN_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif

// wave-uniform values re-materialised per tile: without it LICM hoists the ~300 scalar DMA addresses of a tile (all functions
// of the wave index and the pack pointers) out of the tile loop and parks them in VGPR lanes
__device__ __forceinline__ int opaque_s(int x) {
  asm volatile("" : "+s"(x));
  return x;
}
template <class T>
__device__ __forceinline__ const T* opaque_sp(const T* p) {
  asm volatile("" : "+s"(p));
  return p;
}

// per-lane LDS base that is provably 16-byte aligned (x is a multiple of 4 floats): without the proof every f32x4 access becomes
// ds_read2_b32 pairs -- 4-way bank conflicts at a 16-byte lane stride (the first version of this kernel ran 4.5x off its MFMA bound)
__device__ __forceinline__ int opaque4(int x) { return 4 * opaque(x >> 2); }

// straight-line code for a compile-time range (a `#pragma unroll` loop of this size is refused by the unroller)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// eight float32 values of one lane -> B fragment (three pieces); element j = v[j]
__device__ __forceinline__ X3Frag x3_split8v(const float (&v)[8]) {
  X3Frag f;
  unsigned p1, p2, p3;
#pragma unroll
  for (int jp = 0; jp < 4; ++jp) {
    x3_split2(v[2 * jp], v[2 * jp + 1], p1, p2, p3);
    f.p[0][jp] = p1; f.p[1][jp] = p2; f.p[2][jp] = p3;
  }
  return f;
}

// ---- weight-gradient loops: ONE MFMA per statement, half a pair-split (5 / 6 VALU instructions) behind each ----
// One wave per SIMD hides at most ~5 single-issue instructions per v_mfma_f32_32x32x16_bf16 (24 of its 32 cycles; MI355X_MICROARCH.md,
// 'single-issue instructions HIDDEN per gap').  k_fused_train's x3 loops put a whole pair-split (11 instructions) behind every
// SECOND MFMA (two-MFMA statements): one gap empty, the next 20 cycles over budget -- its x3 phases ran at 55-65 % of the matrix
// rate.  Here every gap carries half a split.
struct SplitMid { unsigned p1; float ra, rb; };
template <bool NATIVE = false>   // NATIVE: the conversion as the compiler's own instruction (chain side work), else the asm statement (kernels_fused.h cvt_pk_bf16)
__device__ __forceinline__ void split_half1(float lo, float hi, SplitMid& m) {
  m.p1 = NATIVE ? cvt_pk_bf16_native(lo, hi) : cvt_pk_bf16(lo, hi);
  m.ra = lo - __uint_as_float(m.p1 << 16);
  m.rb = hi - __uint_as_float(m.p1 & 0xffff0000u);
}
template <bool NATIVE = false>
__device__ __forceinline__ void split_half2(const SplitMid& m, int jp, X3Frag& out) {
  const unsigned p2 = NATIVE ? cvt_pk_bf16_native(m.ra, m.rb) : cvt_pk_bf16(m.ra, m.rb);
  const float sa = m.ra - __uint_as_float(p2 << 16), sb = m.rb - __uint_as_float(p2 & 0xffff0000u);
  out.p[0][jp] = m.p1; out.p[1][jp] = p2; out.p[2][jp] = NATIVE ? cvt_pk_bf16_native(sa, sb) : cvt_pk_bf16(sa, sb);
}
#define MFMA_A1(acc, a_, b_) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a_), "v"(b_))
#define MFMA_V1(acc, a_, b_) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a_), "v"(b_))
#define PIN() __builtin_amdgcn_sched_barrier(0)
// the tile pair (neuron block 0 / 1) x input block of one 16-row k step: twelve MFMAs; the fragment `raw` (the next input block's,
// or nothing) is split behind the first eight
template <bool SPLIT>
__device__ __forceinline__ void dw2_block(f32x16& g0, f32x16& g1, const X3Frag& A0, const X3Frag& A1, const X3Frag& B, const ColFrag& raw,
                                          X3Frag& out) {
  SplitMid m;
  asm volatile("s_nop 1");  // VALU-written operand -> MFMA
  MFMA_A1(g0, A0.p[1], B.p[1]); if (SPLIT) split_half1(raw.v[0], raw.v[1], m); PIN();
  MFMA_A1(g1, A1.p[1], B.p[1]); if (SPLIT) split_half2(m, 0, out); PIN();
  MFMA_A1(g0, A0.p[0], B.p[2]); if (SPLIT) split_half1(raw.v[2], raw.v[3], m); PIN();
  MFMA_A1(g1, A1.p[0], B.p[2]); if (SPLIT) split_half2(m, 1, out); PIN();
  MFMA_A1(g0, A0.p[2], B.p[0]); if (SPLIT) split_half1(raw.v[4], raw.v[5], m); PIN();
  MFMA_A1(g1, A1.p[2], B.p[0]); if (SPLIT) split_half2(m, 2, out); PIN();
  MFMA_A1(g0, A0.p[0], B.p[1]); if (SPLIT) split_half1(raw.v[6], raw.v[7], m); PIN();
  MFMA_A1(g1, A1.p[0], B.p[1]); if (SPLIT) split_half2(m, 3, out); PIN();
  MFMA_A1(g0, A0.p[1], B.p[0]);
  MFMA_A1(g1, A1.p[1], B.p[0]);
  MFMA_A1(g0, A0.p[0], B.p[0]);
  MFMA_A1(g1, A1.p[0], B.p[0]);
}
// the same with TWO fragments split (the next k step's dz2 operands): a half behind each of the first eight MFMAs, two behind the last four
__device__ __forceinline__ void dw2_block2(f32x16& g0, f32x16& g1, const X3Frag& A0, const X3Frag& A1, const X3Frag& B,
                                           const ColFrag& r0, X3Frag& n0, const ColFrag& r1, X3Frag& n1) {
  SplitMid m, m2;
  asm volatile("s_nop 1");
  MFMA_A1(g0, A0.p[1], B.p[1]); split_half1(r0.v[0], r0.v[1], m); PIN();
  MFMA_A1(g1, A1.p[1], B.p[1]); split_half2(m, 0, n0); PIN();
  MFMA_A1(g0, A0.p[0], B.p[2]); split_half1(r0.v[2], r0.v[3], m); PIN();
  MFMA_A1(g1, A1.p[0], B.p[2]); split_half2(m, 1, n0); PIN();
  MFMA_A1(g0, A0.p[2], B.p[0]); split_half1(r0.v[4], r0.v[5], m); PIN();
  MFMA_A1(g1, A1.p[2], B.p[0]); split_half2(m, 2, n0); PIN();
  MFMA_A1(g0, A0.p[0], B.p[1]); split_half1(r0.v[6], r0.v[7], m); PIN();
  MFMA_A1(g1, A1.p[0], B.p[1]); split_half2(m, 3, n0); PIN();
  MFMA_A1(g0, A0.p[1], B.p[0]); split_half1(r1.v[0], r1.v[1], m); split_half1(r1.v[2], r1.v[3], m2); PIN();
  MFMA_A1(g1, A1.p[1], B.p[0]); split_half2(m, 0, n1); split_half2(m2, 1, n1); PIN();
  MFMA_A1(g0, A0.p[0], B.p[0]); split_half1(r1.v[4], r1.v[5], m); split_half1(r1.v[6], r1.v[7], m2); PIN();
  MFMA_A1(g1, A1.p[0], B.p[0]); split_half2(m, 2, n1); split_half2(m2, 3, n1);
}

// the same with the k-step XOR already applied to the base (a0 = base ^ kx) and a constant column-block offset on top: the offset
// folds into the DS instruction's immediate
__device__ __forceinline__ ColFrag img_frag_load_at(int a0, int a1, int off) {
  ColFrag f;
  const f32x4 lo = *reinterpret_cast<const f32x4*>(&lds[a0 + off]);
  const f32x4 hi = *reinterpret_cast<const f32x4*>(&lds[a1 + off]);
#pragma unroll
  for (int j = 0; j < 4; ++j) { f.v[j] = lo[j]; f.v[4 + j] = hi[j]; }
  return f;
}
// eight consecutive batch rows of one image column: two 16-byte chunks whose addresses differ in one XOR bit
__device__ __forceinline__ ColFrag img_frag_load(int a0) {
  ColFrag f;
  const f32x4 lo = *reinterpret_cast<const f32x4*>(&lds[a0]);
  const f32x4 hi = *reinterpret_cast<const f32x4*>(&lds[a0 ^ 4]);
#pragma unroll
  for (int j = 0; j < 4; ++j) { f.v[j] = lo[j]; f.v[4 + j] = hi[j]; }
  return f;
}

// dW1's running sums go back to the slab once per tile (64 KB per workgroup) and are read again one tile later.  Cache policy of
// that store (MOBROB_DW1_ST): 0 plain (the line stays in the XCD's L2: 32 workgroups x 64 KB = 2 MB of the 4 MB, next to 1.8 MB of
// weight packs), 1 sc1 / 3 sc0 sc1 (write-through, line dropped from L2: the re-read comes from the Infinity Cache), 2 nt.
#ifndef MOBROB_DW1_ST
#define MOBROB_DW1_ST 0
#endif
__device__ __forceinline__ void dw1_store(float* base, unsigned byte_off, const f32x4& v) {
#if MOBROB_DW1_ST == 1
  asm volatile("global_store_dwordx4 %0, %1, %2 sc1" ::"v"(byte_off), "v"(v), "s"(base) : "memory");
#elif MOBROB_DW1_ST == 2
  asm volatile("global_store_dwordx4 %0, %1, %2 nt" ::"v"(byte_off), "v"(v), "s"(base) : "memory");
#elif MOBROB_DW1_ST == 3
  asm volatile("global_store_dwordx4 %0, %1, %2 sc0 sc1" ::"v"(byte_off), "v"(v), "s"(base) : "memory");
#else
  stg16(base, byte_off, v);
#endif
}

template <int DP>
__global__ __launch_bounds__(FTHREADS, 1) void k_chain_train(FusedTrainArgs a) {
  using L = CLay<DP>;
  constexpr int K1 = (DP + 31) / 32;       // k steps of layer 1 (32 observation columns each; DP = 16 pads half a step)
  constexpr int NU1 = 16 * K1;             // ring units of layer 1, then 128 of layer 2
  constexpr int NUF = NU1 + 128, NSF = NUF / CSEG;   // the forward stream
  constexpr int NUB = 128, NSB = NUB / CSEG;          // the dh1 stream
  const int tid0 = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int nwg = gridDim.x >> 1;
  int net, wg;
  {  // blockIdx -> (network, tile sequence): as k_fused_train (policy / value workgroups of a tile sequence share an XCD and the rows
     // they gather).  One network per XCD (0.9 MB of weight packs next to the 2 MB of dW1 sums in each 4 MB L2) was measured: L2 misses
     // 1.9 M -> 1.35 M and fabric reads 78 -> 48 MB per launch, in-kernel cycles -4 %, wall time +1.5 % (profiles/r4/chain_l2.txt).
    const int b = blockIdx.x, g16 = b >> 4, o = b & 15;
    const int gsz = min(16, (int)gridDim.x - 16 * g16), half = gsz >> 1;
    net = o >= half ? 1 : 0;
    wg = 8 * g16 + (o - net * half);
  }
  const int slab_id = 2 * wg + net;
  const FusedNet W = a.net[net];
  const u32x4* W1c = reinterpret_cast<const u32x4*>(W.W1c);
  const u32x4* W2c = reinterpret_cast<const u32x4*>(W.W2c);
  const u32x4* W2bc = reinterpret_cast<const u32x4*>(W.W2bc);
  const int ntiles = (a.count + CR - 1) / CR;

  f32x16 gW2[16];   // dW2: this wave's 64 neurons x 256 inputs, pinned in the AGPRs for the whole launch ("+a" statements only)
#pragma unroll
  for (int t = 0; t < 16; ++t) gW2[t] = zero16();
  f32x4 gW3h0 = {0.f, 0.f, 0.f, 0.f}, gW3h1 = gW3h0, gW3h2 = gW3h0, gW3h3 = gW3h0;   // dW3: [16][this wave's 64 columns]
  float* slab = a.slabs + (size_t)slab_id * a.slab_floats;
  float* slab_w1 = slab + slab_off_w1();
  float* slab_w3 = slab + slab_off_w3(DP);
  float gb2 = 0.f, gb1 = 0.f;
  // (loss sums and head-bias / log_std gradient sums are folded per tile over the wave's sixteen rows and kept in LDS -- STAT, GACC:
  //  twelve registers that would otherwise live, unused, through every other phase of the tile)

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
  lds[L::B1 + tid0] = W.b1s[tid0];
  lds[L::B2 + tid0] = W.b2s[tid0];
  if (tid0 < 128 + 16) lds[L::STAT + tid0] = 0.f;   // STAT [4][4] and GACC [4][2][16] are adjacent

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
  // wave-uniform: kept as scalar registers (bit patterns), made vector values only where the loss stage uses them
  const int adv_mean_s = __builtin_amdgcn_readfirstlane(__float_as_int(adv_mean));
  const int adv_sd_s = __builtin_amdgcn_readfirstlane(__float_as_int(adv_sd + 1e-8f));

  // ---- operands of the tile about to be processed, fetched one tile ahead: this lane's batch row (16 wave + lane & 15), the
  //      eight observation columns 32 s + 8 g .. + 7 of each layer-1 k step, its four actions 4 g .. 4 g + 3 and the record tail
  f32x4 xr[2 * K1];
  f32x4 l_act = {0.f, 0.f, 0.f, 0.f}, l_tail = {0.f, 0.f, 0.f, 0.f};
  // Every wave issues the same 2 K1 + 2 vector loads whatever its rows are (dead rows read row 0 and are zeroed afterwards): the
  // weight-gradient phase counts on that number being in flight behind its own loads (s_waitcnt vmcnt(NGL)).
  auto gather_tile = [&](int src_or_neg, int g_) {
    const bool live = src_or_neg >= 0;
    const unsigned src = live ? (unsigned)src_or_neg : 0u;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < K1; ++s) {
      const unsigned col = (32 * s + 8 * g_ < DP) ? (unsigned)(32 * s + 8 * g_) : 0u;   // DP = 16: lane groups 2, 3 hold padding
      const f32x4 lo = ldg16(a.obs, src * (unsigned)(DP * 4) + col * 4u);
      const f32x4 hi = ldg16(a.obs, src * (unsigned)(DP * 4) + (col + 4u) * 4u);
      const bool on = live && 32 * s + 8 * g_ < DP;
      xr[2 * s] = on ? lo : z4;
      xr[2 * s + 1] = on ? hi : z4;
    }
    const f32x4 la = ldg16(a.rec, (src * (unsigned)a.RW + (unsigned)(4 * g_)) * 4u);
    const f32x4 lt = ldg16(a.rec, (src * (unsigned)a.RW + (unsigned)(a.RW - 4)) * 4u);   // old log-prob, advantage, return, old value
    l_act = (live && net == 0) ? la : z4;
    l_tail = live ? lt : z4;
  };
  {
    const int row = wg * CR + 16 * wave + (tid0 & 15);
    gather_tile((wg < ntiles && row < a.count) ? a.rows[row] : -1, (tid0 & 63) >> 4);
  }
  __syncthreads();   // constants and bias tables
#ifdef MOBROB_STAMPS
  unsigned long long* stamps_ = a.stamps;
#endif
  STAMP_INIT()

  bool first_tile = true, primed = false;
  for (int tile = wg; tile < ntiles; tile += nwg) {
    const int tid = opaque(tid0), lane = tid & 63;
    const int brow = lane & 15, g = lane >> 4;
    const int trow = 16 * wave + brow;                 // row of the tile
    const bool live = tile * CR + trow < a.count;
    const unsigned lane16 = opaque_u((unsigned)lane * 16u);
    const int ringl = opaque4(L::RING + 4 * lane);
    const int wv = opaque_s(wave);
    const u32x4* W1c_ = opaque_sp(W1c);
    const u32x4* W2c_ = opaque_sp(W2c);
    const u32x4* W2bc_ = opaque_sp(W2bc);
    const int nrow0 = (tile + nwg) * CR;
    const bool has_next = tile + nwg < ntiles;

    // ---- X: float32 image for dW1 (column-major, rows contiguous) and the layer-1 B fragments (split once) ----
    X3Frag xp[K1];
#pragma unroll
    for (int s = 0; s < K1; ++s) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[j] = xr[2 * s][j]; v[4 + j] = xr[2 * s + 1][j]; }
      if (32 * s + 8 * g < DP) {
#pragma unroll
        for (int j = 0; j < 8; ++j) lds[L::XI + img_addr(32 * s + 8 * g + j, trow)] = v[j];
      }
      xp[s] = x3_split8v(v);
    }

    STAMP(0)
    // ============================ forward: layer 1 and layer 2 as ONE stream of ring units ============================
    // unit u < NU1: layer 1, neuron tile u / K1, k step u % K1 (pack W1c, [tile][k step]: a tile is complete after K1 units, its
    // tanh goes to the h1 image under the next tile's MFMAs and its accumulator dies -- layer 1 holds four registers, not 64);
    // then layer 2, k step (u - NU1) / 16, neuron tile (u - NU1) % 16 (pack W2c), whose B fragments are this lane's own eight
    // elements of the h1 image, read back and split one k step ahead.
    // per-wave bases of the DMAs, re-materialised per tile: the unit a wave moves is base + a compile-time constant (source and
    // LDS destination alike), two scalar adds and one M0 write per unit instead of the multiply / shift chains of (8 q + 4 hh + wave)
    const u32x4* w1w = opaque_sp(W1c_ + (size_t)wv * 192);
    const u32x4* w2w = opaque_sp(W2c_ + (size_t)wv * 192);
    const u32x4* w2bw = opaque_sp(((MOBROB_CHAIN_SKIP & 64) ? W2c_ : W2bc_) + (size_t)wv * 192);
    const int ringw = opaque_s(L::RING + wv * CUNIT);
    auto fwd_issue = [&](int q) {   // DMAs of segment q: this wave moves units 8 q + wave and 8 q + 4 + wave
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int u0 = CSEG * q + 4 * hh;
        const u32x4* src = (CSEG * q < NU1) ? w1w + (size_t)u0 * 192 : w2w + (size_t)(u0 - NU1) * 192;
        dma_unit(src, u0 % CSLOTS, lane16, ringw);
      }
    };
    // this lane's elements of an image: column 16 t + 4 g + i, row trow -> hb[i] + 1024 t
    int hb1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) hb1[i] = opaque(L::H1 + img_addr(4 * g + i, trow));
    f32x4 acc2[16];
    f32x4 hw[8];           // rolling window over the float32 fragments of the head (forward: 16 tiles), then of dh2 (16 tiles): [tile][lane] x 16 bytes
#pragma unroll
    for (int t = 0; t < 16; ++t) acc2[t] = *reinterpret_cast<const f32x4*>(&lds[L::B2 + 16 * t + 4 * g]);
    if (!primed) {         // (the first tile; later tiles had their first two segments started under the previous tile's dW1 phase)
      fwd_issue(0);
      fwd_issue(1);
    }
    CHAIN_WAIT_DMA(true);
    CHAIN_BARRIER();       // also: the previous tile's last reads of the images are complete everywhere
    STAMP(18)
    fwd_issue(2);
    RingW Wr;
    Wr.p0[0] = ring_read_piece(ringl, 0, 0); Wr.p1 = ring_read_piece(ringl, 0, 1); Wr.p2 = ring_read_piece(ringl, 0, 2);
    X3Frag Bc, Bn;         // B fragment of the current / next layer-2 k step
    float hv[8];           // float32 elements of the B fragment being prepared
    SplitMid sm;           // a pair-split between its two halves
    f32x4 c1 = {0.f, 0.f, 0.f, 0.f}, pend = c1;   // layer 1: the tile being accumulated / the finished tile awaiting its tanh
    f32x4 cn = *reinterpret_cast<const f32x4*>(&lds[L::B1 + 4 * g]);   // the bias the next tile starts from, read one unit ahead
    static_for<0, NUF>([&](auto uc) {
      constexpr int u = decltype(uc)::value;
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (u == NU1) { STAMP(1) }
      if constexpr (u == NUF - 2 * CSEG) {   // the head's weight fragments (16 KB from L2): two ring segments ahead of their use
#pragma unroll
        for (int t = 0; t < 8; ++t) hw[t] = ldg16(W.W3c, lane16 + 1024u * t);   // uniform base + per-lane 32-bit offset: no 64-bit pointer registers
      }
      if constexpr ((u + 1) % CSEG == 0 && u + 1 < NUF) {
        constexpr int q = (u + 1) / CSEG;
        CHAIN_WAIT_DMA(q + 1 < NSF);
        CHAIN_BARRIER();
        if constexpr (!MOBROB_CHAIN_SPREAD && q + 2 < NSF) fwd_issue(q + 2);
      }
      if constexpr (MOBROB_CHAIN_SPREAD && u >= CSEG - 1) {   // piece i of segment q + 2 behind unit 8 q - 1 + i (i = 0 .. 5): after barrier q
        constexpr int q = (u + 1) / CSEG, i = (u + 1) % CSEG;
        if constexpr (i < 6 && q + 2 < NSF) {
          constexpr int u0 = CSEG * (q + 2) + 4 * (i / 3);
          const u32x4* src = (CSEG * (q + 2) < NU1) ? w1w + (size_t)u0 * 192 : w2w + (size_t)(u0 - NU1) * 192;
          dma_piece(src, u0 % CSLOTS, i % 3, lane16, ringw);
        }
      }
      if constexpr (u < NU1) {
        constexpr int t = u / K1, ks = u % K1;
        if constexpr (ks == 0) c1 = cn;
        if constexpr (ks == K1 - 1 && t + 1 < 16) cn = *reinterpret_cast<const f32x4*>(&lds[L::B1 + 16 * (t + 1) + 4 * g]);
        CHAIN_UNIT(Wr, u & 1, xp[ks], c1, u + 1 < NUF, (u + 1) % CSLOTS)
        if constexpr (t > 0 && ks == 0) {   // the previous tile: tanh -> h1 image
#pragma unroll
          for (int i = 0; i < 4; ++i) lds[hb1[i] + 1024 * (t - 1)] = fast_tanh_scaled(pend[i]);
        }
        if constexpr (ks == K1 - 1) pend = c1;
      } else {
        constexpr int v = u - NU1, t = v % 16, ks = v / 16;
        if constexpr (v == 0) {
#pragma unroll
          for (int i = 0; i < 4; ++i) lds[hb1[i] + 1024 * 15] = fast_tanh_scaled(pend[i]);
        }
        CHAIN_UNIT(Wr, u & 1, Bc, acc2[t], u + 1 < NUF, (u + 1) % CSLOTS)
      }
      // side work: the B fragment of layer-2 k step sn, prepared in the sixteen units in front of it (sn = 0: the last sixteen
      // units of layer 1, by which time tiles 0 and 1 of h1 are in the image): eight reads, then four pair-splits
      constexpr int w0 = u - (NU1 - 16);          // position in the windows of sixteen units
      if constexpr (w0 >= 0 && w0 / 16 < 8) {
        constexpr int sn = w0 / 16, c = w0 % 16;
        if constexpr (c >= 4 && c < 12) {
          constexpr int e = c - 4;
          hv[e] = lds[hb1[e & 3] + 1024 * (2 * sn + (e >> 2))];
        }
        // pair jp (elements 2 jp, 2 jp + 1, read in slices 4 + 2 jp, 5 + 2 jp) is split in two halves, slices 6 + 2 jp and 7 + 2 jp:
        // five or six VALU instructions per unit, not eleven in every fourth
        if constexpr (c >= 6 && c < 14) {
          constexpr int jp = (c - 6) >> 1;
          if constexpr (((c - 6) & 1) == 0) split_half1<true>(hv[2 * jp], hv[2 * jp + 1], sm);
          else split_half2<true>(sm, jp, Bn);
        }
        if constexpr (c == 15) Bc = Bn;
      }
    });
    // every wave is done with the ring: its space beyond the first four units becomes the h2 image
    CHAIN_BARRIER();
    STAMP(2)

    // ============================ h2 = tanh, head (float32 16x16x4), loss, dout ============================
    int hb2[4];            // the same elements of the h2 / dz2 image (derived here: four registers that need not live through the ring phases)
#pragma unroll
    for (int i = 0; i < 4; ++i) hb2[i] = opaque(hb1[i] + (L::H2 - L::H1));
    f32x4 mean = {0.f, 0.f, 0.f, 0.f};
    {
      f32x4 mean2 = mean;   // two accumulation chains (even / odd tiles): a dependent v_mfma_f32_16x16x4_f32 waits 40 cycles, an independent one 32
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const f32x4 wv = hw[t & 7];
        // the register the head's fragment just left takes the fragment needed eight tiles later: the head's own for t < 8, then dh2's
        hw[t & 7] = t < 8 ? ldg16(W.W3c, lane16 + 1024u * (t + 8)) : ldg16(W.W3bc, lane16 + 1024u * (t - 8));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float h = fast_tanh_scaled(acc2[t][i]);
          acc2[t][i] = h;
          lds[hb2[i] + 1024 * t] = h;
          if (t & 1) mean2 = MFMA16(wv[i], h, mean2);
          else mean = MFMA16(wv[i], h, mean);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) mean[i] += mean2[i];
    }
    f32x4 dout = {0.f, 0.f, 0.f, 0.f};
    {
      const f32x4 ivv = *reinterpret_cast<const f32x4*>(&lds[L::CST + 4 * g]);
      const f32x4 lcv = *reinterpret_cast<const f32x4*>(&lds[L::CST + 32 + 4 * g]);
      const f32x4 b3v = *reinterpret_cast<const f32x4*>(&lds[L::CST + 64 + 4 * g]);
      if (net == 0) {
        float lp = 0.f, d[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          d[i] = 0.f;
          if (4 * g + i < a.A && live) {
            d[i] = l_act[i] - (mean[i] + b3v[i]);
            lp += -(d[i] * d[i]) * (0.5f * ivv[i]) - lcv[i];
          }
        }
        lp += xor_lane(lp, 16);
        lp += xor_lane(lp, 32);
        float g_logp = 0.f, t_pl = 0.f, t_cf = 0.f, t_kl = 0.f;
        float am, asd;   // fresh vector copies of the two scalar registers (not hoistable: as loop-carried vector registers one was spilled)
        asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=v"(am), "=v"(asd) : "s"(adv_mean_s), "s"(adv_sd_s));
        if (live) {
          float adv = l_tail[1];
          if (a.normalize && adv_on)   // (scalar registers, re-read here: as loop-carried vector registers one of them was spilled)
            adv = (adv - am) / asd;
          const float log_ratio = lp - l_tail[0];
          const float ratio = expf(log_ratio);
          const float lo = 1.0f - a.clip, hi = 1.0f + a.clip;
          const float s1 = adv * ratio, s2 = adv * fminf(fmaxf(ratio, lo), hi);
          if (g == 0) {
            t_pl = fminf(s1, s2);
            t_cf = (fabsf(ratio - 1.0f) > a.clip) ? 1.f : 0.f;
            t_kl = (ratio - 1.0f) - log_ratio;
          }
          const float in_range = (ratio >= lo && ratio <= hi) ? 1.f : 0.f;
          const float w1 = (s1 < s2) ? 1.f : ((s1 > s2) ? 0.f : 0.5f);
          g_logp = -(w1 * adv + (1.0f - w1) * adv * in_range) * a.inv_bg * ratio;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          dout[i] = g_logp * d[i] * ivv[i];   // zero beyond the head (d = 0, 1/var = 0)
          float gm = dout[i], gl = (4 * g + i < a.A) ? g_logp * (d[i] * d[i] * ivv[i] - 1.0f) : 0.f;
#pragma unroll
          for (int o = 1; o < 16; o <<= 1) { gm += xor_lane(gm, o); gl += xor_lane(gl, o); }   // over the wave's sixteen rows (lanes of a group)
          if (brow == 0) {
            lds[L::GACC + wave * 32 + 4 * g + i] += gm;
            lds[L::GACC + wave * 32 + 16 + 4 * g + i] += gl;
          }
        }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { t_pl += xor_lane(t_pl, o); t_cf += xor_lane(t_cf, o); t_kl += xor_lane(t_kl, o); }
        if (lane == 0) {
          lds[L::STAT + wave * 4 + 0] += t_pl;
          lds[L::STAT + wave * 4 + 2] += t_kl;
          lds[L::STAT + wave * 4 + 3] += t_cf;
        }
      } else {
        float t_vl = 0.f;
        if (live && g == 0) {
          float sq, gv_;
          value_loss_terms(mean[0] + b3v[0], l_tail[2], a.clip_vf >= 0.f ? l_tail[3] : 0.f, a.clip_vf, sq, gv_);
          t_vl = sq;
          dout[0] = a.vf_coef * gv_ * a.inv_bg;
        }
        float gm = dout[0];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { gm += xor_lane(gm, o); t_vl += xor_lane(t_vl, o); }
        if (lane == 0) {
          lds[L::GACC + wave * 32] += gm;
          lds[L::STAT + wave * 4 + 1] += t_vl;
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) lds[L::DO + img_addr(4 * g + i, trow)] = dout[i];   // dout image [16 head rows][64 batch rows], swizzled like the others
    }
    // dh2 = W3^T dout (float32 16x16x4: k slot g of step i = head row 4 g + i), dz2 = dh2 (1 - h2^2): in registers
    {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const f32x4 wv = hw[t & 7];
        if (t < 8) hw[t] = ldg16(W.W3bc, lane16 + 1024u * (t + 8));
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) c = MFMA16(wv[i], dout[i], c);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc2[t][i] = c[i] * (1.0f - acc2[t][i] * acc2[t][i]);
      }
    }
    // (Touching the dh1 weight pack's lines here, 19 k cycles ahead of its stream -- it has usually left the XCD's L2 since the last
    //  tile -- was measured twice: by every workgroup for itself (dh1 loop 24.3 k -> 19.8 k cycles, 2.2 k of issue here, launch 332.7 ->
    //  341.5 us on the same box) and by ONE of the sixteen workgroups of an XCD and network per tile iteration (total 104.0 k -> 102.0 k
    //  cycles per tile, launch 338.2 -> 343.8 us).  Fewer cycles, longer launches, both times: not kept; profiles/r4/chain_l2.txt.)
    STAMP(3)
    LDS_BARRIER();   // h2 image and dout image complete
    STAMP(4)

    // ============================ dW3 += dout^T . h2 (float32 16x16x4; this wave's 64 columns) ============================
    {
      // k slot kk of MFMA step s carries batch row 16 kk + s: a lane's sixteen steps are sixteen CONSECUTIVE rows of its columns, i.e.
      // four 16-byte chunks each of the transposed images (the first version read row 4 s + kk: five ds_read_b32 per step, 2-way conflicts)
      const int i16 = lane & 15, kk = lane >> 4;
      const int ao = opaque4(L::DO + i16 * 64);                                   // A[a][k] = dout[row][a]: column a = i16 of the dout image
      const int bo = opaque4(L::H2 + (64 * wave + i16) * 64);                      // B[k][column 64 wave + 16 b + i16] = h2[row][column]
#pragma unroll
      for (int q = 0; q < 4; ++q) {                                               // rows 16 kk + 4 q .. + 3 = chunk 4 kk + q
        const int ch = (((4 * kk + q) ^ i16) & 15) << 2;
        const f32x4 xa = *reinterpret_cast<const f32x4*>(&lds[ao + ch]);
        const f32x4 y0 = *reinterpret_cast<const f32x4*>(&lds[bo + ch]);
        const f32x4 y1 = *reinterpret_cast<const f32x4*>(&lds[bo + 16 * 64 + ch]);
        const f32x4 y2 = *reinterpret_cast<const f32x4*>(&lds[bo + 32 * 64 + ch]);
        const f32x4 y3 = *reinterpret_cast<const f32x4*>(&lds[bo + 48 * 64 + ch]);
#pragma unroll
        for (int e = 0; e < 4; ++e) mfma16_x1y4(gW3h0, gW3h1, gW3h2, gW3h3, xa[e], y0[e], y1[e], y2[e], y3[e]);
      }
    }
    STAMP(5)
    LDS_BARRIER();   // the h2 image has been read: dz2 overwrites it
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) lds[hb2[i] + 1024 * t] = acc2[t][i];
    LDS_BARRIER();   // dz2 image complete
    STAMP(6)

    // ============================ dW2 += dz2^T . h1 (bf16 pipe, K = 64 rows; this wave: 64 neurons x 256 inputs) ============================
    int nsrc = -1;
    {
      const int r = lane & 31, h = lane >> 5;
      // a lane's fragment: rows 16 ks + 8 h .. + 7 of one column = chunks 4 ks + 2 h, + 1 (the second = the first ^ 1)
      const int ao = opaque4(L::H2 + (64 * wave + r) * 64 + ((((2 * h) ^ r) & 15) << 2));
      const int bo = opaque4(L::H1 + r * 64 + ((((2 * h) ^ r) & 15) << 2));
      const int co = opaque4(L::H2 + tid * 64);
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      nsrc = (has_next && nrow0 + trow < a.count) ? a.rows[nrow0 + trow] : -1;   // level 1 of the next tile's gathers
      // The two dz2 fragments of k step ks + 1 are read at the top of step ks and split between the MFMAs of its last two input
      // blocks (jb = 6, 7 carry no h1 split of their own: the fragment ring of h1 ends there) -- the first version split them in
      // front of every step, 600 cycles with the matrix pipe idle.
      X3Frag A0 = col_frag_split(img_frag_load(ao)), A1 = col_frag_split(img_frag_load(ao + 32 * 64));
#pragma unroll 1
      for (int ks = 0; ks < CR / 16; ++ks) {
        const int kx = ks << 4;   // chunk bits 2..3 of the address
        const int kn = (ks + 1 < CR / 16 ? ks + 1 : ks) << 4;
        const ColFrag rA0 = img_frag_load(ao ^ kn), rA1 = img_frag_load((ao + 32 * 64) ^ kn);
        X3Frag nA0, nA1;
        {  // db2: column `tid` of dz2 over these sixteen rows (four chunks), rows = 0..3 (mod 4) -> s0..s3
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(&lds[co + ((((4 * ks + c) ^ tid) & 15) << 2)]);
            s0 += v[0]; s1 += v[1]; s2 += v[2]; s3 += v[3];
          }
        }
        const int b0x = bo ^ kx, b1x = b0x ^ 4;   // (the k-step bits 4, 5 of the address and the column-block offsets, multiples of 2048, do not interact)
        X3Frag B = col_frag_split(img_frag_load_at(b0x, b1x, 0));
        ColFrag raw = img_frag_load_at(b0x, b1x, 32 * 64);
#pragma unroll
        for (int jb = 0; jb < 8; ++jb) {
          X3Frag Bn2;
          __builtin_amdgcn_sched_barrier(0);
          constexpr bool kSplitB = !(MOBROB_CHAIN_SKIP & 128);   // (128, timing only: the h1 fragments are not split -- what split-once planes could save at most)
          if (jb < 6) {
            const ColFrag raw2 = img_frag_load_at(b0x, b1x, 32 * 64 * (jb + 2));
            dw2_block<kSplitB>(gW2[jb], gW2[8 + jb], A0, A1, B, raw, Bn2);
            if (kSplitB) B = Bn2;
            raw = raw2;
          } else if (jb == 6) {
            dw2_block<kSplitB>(gW2[jb], gW2[8 + jb], A0, A1, B, raw, Bn2);   // splits the last h1 fragment of this step
            if (kSplitB) B = Bn2;
          } else {
            dw2_block2(gW2[jb], gW2[8 + jb], A0, A1, B, rA0, nA0, rA1, nA1);   // splits the next step's two dz2 fragments
          }
        }
        A0 = nA0; A1 = nA1;
      }
      gb2 += (s0 + s1) + (s2 + s3);
    }
    STAMP(7)
    LDS_BARRIER();   // the dz2 image has been read: its space is ring again
    STAMP(8)

    // ============================ dh1 = W2^T dz2 (chain; B fragments from the dz2 registers), dz1 = dh1 (1 - h1^2) ============================
    // Two passes over the k steps, eight neuron tiles of dh1 each (pack W2bc: [half][k step][tile of the half]): 32 accumulator
    // registers instead of 64 next to the 64 of dz2.  B fragment of k step s: dz2 tiles 2 s, 2 s + 1, split one step ahead.
    // dW1's accumulators START from the workgroup's running sums in its slab (no registers held across tiles, none for a copy):
    // loaded behind the last MFMA of this phase (sc1: past the vector L1 -- the lines were written by this lane one tile ago), due
    // at the first MFMA of the dW1 phase.
    constexpr bool two_ = DP > 32;
    constexpr int NGL = 2 * K1 + 2;   // vector loads of gather_tile, issued behind the slab loads and allowed to stay in flight
    const unsigned sb1 = (unsigned)(wave * 4 * 4 * 64 + lane) * 16u;
    f32x16 gW1a = zero16(), gW1b = zero16(), gW1c = zero16(), gW1d = zero16();   // [ib][jb] = 00, 10, 01, 11 -> slab tiles 0, 2, 1, 3
    auto slab_w1_load = [&](f32x16& acc, int tile_idx) {
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        f32x4 v;
        asm volatile("global_load_dwordx4 %0, %1, %2 sc1" : "=v"(v) : "v"(sb1 + (unsigned)(tile_idx * 4 + qd) * 1024u), "s"(slab_w1) : "memory");
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[4 * qd + e] = v[e];
      }
    };
    auto bwd_issue = [&](int q) {
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int u0 = CSEG * q + 4 * hh;
        dma_unit(w2bw + (size_t)u0 * 192, u0 % CSLOTS, lane16, ringw);
      }
    };
    {
      f32x4 acc4[8];
      X3Frag Bd, Bdn;        // B fragment of the current / next k step
      SplitMid smd;          // a pair-split between its two halves
      bwd_issue(0);
      bwd_issue(1);
      {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = acc2[j >> 2][j & 3];
        Bd = x3_split8v(v);
      }
      CHAIN_WAIT_DMA(true);
      CHAIN_BARRIER();
      STAMP(16)
      bwd_issue(2);
      Wr.p0[0] = ring_read_piece(ringl, 0, 0); Wr.p1 = ring_read_piece(ringl, 0, 1); Wr.p2 = ring_read_piece(ringl, 0, 2);
      static_for<0, NUB>([&](auto uc) {
        constexpr int u = decltype(uc)::value;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr ((u + 1) % CSEG == 0 && u + 1 < NUB) {
          constexpr int q = (u + 1) / CSEG;
          CHAIN_WAIT_DMA(q + 1 < NSB);
          CHAIN_BARRIER();
          if constexpr (!MOBROB_CHAIN_SPREAD && q + 2 < NSB) bwd_issue(q + 2);
        }
        if constexpr (MOBROB_CHAIN_SPREAD && u >= CSEG - 1) {
          constexpr int q = (u + 1) / CSEG, i = (u + 1) % CSEG;
          if constexpr (i < 6 && q + 2 < NSB) {
            constexpr int u0 = CSEG * (q + 2) + 4 * (i / 3);
            dma_piece(w2bw + (size_t)u0 * 192, u0 % CSLOTS, i % 3, lane16, ringw);
          }
        }
        constexpr int half = u / 64, ks = (u % 64) / 8, t8 = u % 8;
        if constexpr (ks == 0) acc4[t8] = f32x4{0.f, 0.f, 0.f, 0.f};
        CHAIN_UNIT(Wr, u & 1, Bd, acc4[t8], u + 1 < NUB, (u + 1) % CSLOTS)
        constexpr int sn = (ks + 1) % 8;           // the next k step (the second pass starts over at 0)
        if constexpr (u + 1 < NUB) {               // its B fragment: pair jp in units 2 jp (first half) and 2 jp + 1 (second half) of this step
          constexpr int jp = t8 >> 1, e0 = 2 * jp, e1 = 2 * jp + 1;
          if constexpr ((t8 & 1) == 0) split_half1<true>(acc2[2 * sn + (e0 >> 2)][e0 & 3], acc2[2 * sn + (e1 >> 2)][e1 & 3], smd);
          else split_half2<true>(smd, jp, Bdn);
          if constexpr (t8 == 7) Bd = Bdn;
        }
        if constexpr (u == NUB - 1) {
          STAMP(17)
          if (!first_tile && !(MOBROB_CHAIN_SKIP & 16)) {
            slab_w1_load(gW1a, 0);
            slab_w1_load(gW1b, 2);
            if constexpr (two_) { slab_w1_load(gW1c, 1); slab_w1_load(gW1d, 3); }
          }
        }
        if constexpr (u % 64 == 63) {
          // dz1 over h1, in place, for the eight tiles of this pass (every lane rewrites exactly the elements it wrote in the
          // forward pass; dW2's reads of the h1 image are complete everywhere: all waves have passed this phase's first barrier)
          // (all thirty-two h1 values are read first: a read-modify-write per element is a chain of dependent LDS round trips --
          //  every read waits behind the previous write, which may alias it as far as the compiler knows: 4 k cycles per pass)
          float h1v[32];
#pragma unroll
          for (int tt = 0; tt < 8; ++tt)
#pragma unroll
            for (int i = 0; i < 4; ++i) h1v[4 * tt + i] = lds[hb1[i] + 1024 * (8 * half + tt)];
#pragma unroll
          for (int tt = 0; tt < 8; ++tt)
#pragma unroll
            for (int i = 0; i < 4; ++i) lds[hb1[i] + 1024 * (8 * half + tt)] = acc4[tt][i] * (1.0f - h1v[4 * tt + i] * h1v[4 * tt + i]);
        }
      });
    }
    STAMP(9)
    // dW1's first X fragments: the X image was complete long ago, so they are read and split in front of the barrier -- while this wave
    // would otherwise wait for the slowest wave's dz1 epilogue
    X3Frag B0, B1;
    int b0o, b1o;
    {
      const int r = lane & 31, h = lane >> 5;
      const int c0 = (r < DP) ? r : 0;
      const int c1 = (32 + r < DP) ? 32 + r : c0;   // clamped columns are never read back
      b0o = opaque4(L::XI + c0 * 64 + ((((2 * h) ^ c0) & 15) << 2));
      b1o = opaque4(L::XI + c1 * 64 + ((((2 * h) ^ c1) & 15) << 2));
      B0 = col_frag_split(img_frag_load(b0o));
      B1 = B0;
      if (DP > 32) B1 = col_frag_split(img_frag_load(b1o));
    }
    // level 2 of the next tile's gathers: in flight under dW1 (LDS operands only) and into the next tile
    if (!(MOBROB_CHAIN_SKIP & 32)) gather_tile(nsrc, g);
    LDS_BARRIER();   // dz1 image complete (and every wave is done with the ring)
    STAMP(10)
    primed = has_next;
    if (primed) {      // the next tile's first two ring segments: their L2 latency runs under this tile's dW1 phase
      fwd_issue(0);
      fwd_issue(1);
    }

    // ============================ dW1 += dz1^T . X (bf16 pipe; this wave: 64 neurons x DP inputs), added to the slab ============================
    {
      constexpr bool two = DP > 32;
      const int r = lane & 31, h = lane >> 5;
      const int ao = opaque4(L::H1 + (64 * wave + r) * 64 + ((((2 * h) ^ r) & 15) << 2));
      const int co = opaque4(L::H1 + tid * 64);
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      // Software pipeline over the four k steps of sixteen rows: while the six MFMA statements of step ks run (four MFMAs each at
      // DP = 64), the fragments of step ks + 1 -- read from LDS at the top of the step -- are split pair by pair between them
      // (the first version split everything in front of the MFMAs: 12.5 k cycles per tile for 3 k of matrix time).
      X3Frag A0 = col_frag_split(img_frag_load(ao)), A1 = col_frag_split(img_frag_load(ao + 32 * 64));
      STAMP(13)
      // the slab loads have landed (the gathers behind them, and the twelve ring DMAs of a primed next tile, may still be in flight).
      // What this hand-counted wait rests on is CHECKED on every build (tests/test_chain_isa.py, from the disassembly of the shipped
      // code object): the slab loads write the accumulator registers themselves (the quads coalesced into the 16-register tuples:
      // nothing reads or copies a loaded register in front of this wait), exactly NGL gather loads and twelve ring DMAs -- and no
      // other vector-memory instruction -- lie between the slab loads and it.  Collecting the quads in standalone registers and
      // assembling the tuples behind the wait cost 64 transient registers: three spills at 64 observation columns.
      // -DMOBROB_CHAIN_VMCNT0 (validation build): vmcnt(0) here and at the ring's segment boundaries; its gradients must be
      // bit-equal to the default build's (tests/test_engine_gpu.py::test_chain_counted_waits_equal_full_waits).
#ifdef MOBROB_CHAIN_VMCNT0
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(gW1a), "+v"(gW1b), "+v"(gW1c), "+v"(gW1d) : : "memory");
#else
      if (primed) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(gW1a), "+v"(gW1b), "+v"(gW1c), "+v"(gW1d) : "n"(NGL + 12) : "memory");
      else asm volatile("s_waitcnt vmcnt(%4)" : "+v"(gW1a), "+v"(gW1b), "+v"(gW1c), "+v"(gW1d) : "n"(NGL) : "memory");
#endif
      STAMP(14)
#pragma unroll 1
      for (int ks = 0; ks < CR / 16; ++ks) {
        const int kn = (ks + 1 < CR / 16 ? ks + 1 : ks) << 4;   // the last step re-reads its own rows (unused)
        const ColFrag rA0 = img_frag_load(ao ^ kn), rA1 = img_frag_load((ao + 32 * 64) ^ kn);
        const ColFrag rB0 = img_frag_load(b0o ^ kn);
        ColFrag rB1 = rB0;
        if (two) rB1 = img_frag_load(b1o ^ kn);
        X3Frag nA0, nA1, nB0, nB1;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(&lds[co + ((((4 * ks + c) ^ tid) & 15) << 2)]);
          s0 += v[0]; s1 += v[1]; s2 += v[2]; s3 += v[3];
        }
        asm volatile("s_nop 1");
        // 24 (12) MFMAs, a half-split behind each: 16 (12) pair-splits = 32 (24) halves; the surplus of the wide case doubles up at the end
        SplitMid m, m2;
#define DW1_Q(ia, ib, S0, S1, S2, S3)                                              \
        MFMA_V1(gW1a, A0.p[ia], B0.p[ib]); S0; PIN();                              \
        MFMA_V1(gW1b, A1.p[ia], B0.p[ib]); S1; PIN();                              \
        if (two) { MFMA_V1(gW1c, A0.p[ia], B1.p[ib]); S2; PIN();                   \
                   MFMA_V1(gW1d, A1.p[ia], B1.p[ib]); S3; PIN(); }
        if (two) {
          DW1_Q(1, 1, split_half1(rA0.v[0], rA0.v[1], m), split_half2(m, 0, nA0), split_half1(rA0.v[2], rA0.v[3], m), split_half2(m, 1, nA0))
          DW1_Q(0, 2, split_half1(rA0.v[4], rA0.v[5], m), split_half2(m, 2, nA0), split_half1(rA0.v[6], rA0.v[7], m), split_half2(m, 3, nA0))
          DW1_Q(2, 0, split_half1(rA1.v[0], rA1.v[1], m), split_half2(m, 0, nA1), split_half1(rA1.v[2], rA1.v[3], m), split_half2(m, 1, nA1))
          DW1_Q(0, 1, split_half1(rA1.v[4], rA1.v[5], m), split_half2(m, 2, nA1), split_half1(rA1.v[6], rA1.v[7], m), split_half2(m, 3, nA1))
          DW1_Q(1, 0, (split_half1(rB0.v[0], rB0.v[1], m), split_half1(rB0.v[2], rB0.v[3], m2)), (split_half2(m, 0, nB0), split_half2(m2, 1, nB0)),
                (split_half1(rB0.v[4], rB0.v[5], m), split_half1(rB0.v[6], rB0.v[7], m2)), (split_half2(m, 2, nB0), split_half2(m2, 3, nB0)))
          DW1_Q(0, 0, (split_half1(rB1.v[0], rB1.v[1], m), split_half1(rB1.v[2], rB1.v[3], m2)), (split_half2(m, 0, nB1), split_half2(m2, 1, nB1)),
                (split_half1(rB1.v[4], rB1.v[5], m), split_half1(rB1.v[6], rB1.v[7], m2)), (split_half2(m, 2, nB1), split_half2(m2, 3, nB1)))
        } else {
          DW1_Q(1, 1, (split_half1(rA0.v[0], rA0.v[1], m), split_half1(rA0.v[2], rA0.v[3], m2)), (split_half2(m, 0, nA0), split_half2(m2, 1, nA0)), (void)0, (void)0)
          DW1_Q(0, 2, (split_half1(rA0.v[4], rA0.v[5], m), split_half1(rA0.v[6], rA0.v[7], m2)), (split_half2(m, 2, nA0), split_half2(m2, 3, nA0)), (void)0, (void)0)
          DW1_Q(2, 0, (split_half1(rA1.v[0], rA1.v[1], m), split_half1(rA1.v[2], rA1.v[3], m2)), (split_half2(m, 0, nA1), split_half2(m2, 1, nA1)), (void)0, (void)0)
          DW1_Q(0, 1, (split_half1(rA1.v[4], rA1.v[5], m), split_half1(rA1.v[6], rA1.v[7], m2)), (split_half2(m, 2, nA1), split_half2(m2, 3, nA1)), (void)0, (void)0)
          DW1_Q(1, 0, (split_half1(rB0.v[0], rB0.v[1], m), split_half1(rB0.v[2], rB0.v[3], m2)), (split_half2(m, 0, nB0), split_half2(m2, 1, nB0)), (void)0, (void)0)
          DW1_Q(0, 0, (split_half1(rB0.v[4], rB0.v[5], m), split_half1(rB0.v[6], rB0.v[7], m2)), (split_half2(m, 2, nB0), split_half2(m2, 3, nB0)), (void)0, (void)0)
          nB1 = nB0;
        }
#undef DW1_Q
        A0 = nA0; A1 = nA1; B0 = nB0; B1 = nB1;
      }
      asm volatile("s_nop 15\n\ts_nop 7" : "+v"(gW1a), "+v"(gW1b), "+v"(gW1c), "+v"(gW1d));   // opaque MFMA statements: XDL write -> VALU read
      STAMP(15)
      gb1 += (s0 + s1) + (s2 + s3);
      // the new running sums go back to the slab (fragment order [w][tile][quad][lane] x 16 B, as k_fused_train stores it)
      auto put = [&](const f32x16& gacc, int tile_idx) {
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const f32x4 v = {gacc[4 * qd], gacc[4 * qd + 1], gacc[4 * qd + 2], gacc[4 * qd + 3]};
          if (MOBROB_CHAIN_SKIP & 8) asm volatile("" :: "v"(v));
          else dw1_store(slab_w1, sb1 + (unsigned)(tile_idx * 4 + qd) * 1024u, v);
        }
      };
      put(gW1a, 0);
      put(gW1b, 2);
      if constexpr (two) { put(gW1c, 1); put(gW1d, 3); }
    }
    first_tile = false;
    STAMP(11)
    LDS_BARRIER();   // images are rewritten by the next tile
    STAMP(12)
  }
  STAMP_FLUSH()

  const int tid = tid0, lane = tid & 63;
  // ---- store this workgroup's partial gradients to its slab (k_fused_train's layout) ----
  asm volatile("s_nop 15\n\ts_nop 3");
  if (first_tile) {   // a workgroup without tiles: its dW1 region was never written
    const unsigned sb = (unsigned)(wave * 4 * 4 * 64 + lane) * 16u;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 16; ++k) stg16(slab_w1, sb + (unsigned)k * 1024u, z);
  }
  {
    const unsigned sh = (unsigned)(wave * 4 * 64 + lane) * 16u;
    stg16(slab_w3, sh, gW3h0);
    stg16(slab_w3, sh + 1024u, gW3h1);
    stg16(slab_w3, sh + 2048u, gW3h2);
    stg16(slab_w3, sh + 3072u, gW3h3);
  }
  float* w2base = slab + slab_off_w2();
  const unsigned s2 = (unsigned)(wave * 16 * 4 * 64 + lane) * 16u;
#pragma unroll
  for (int t = 0; t < 16; ++t) {
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = gW2[t][4 * qd + e];
      if (MOBROB_CHAIN_SKIP & 256) asm volatile("" :: "v"(v));   // (256, timing only: no dW2 slab stores -- what hiding them could gain at most)
      else stg16(w2base, s2 + (unsigned)(t * 4 + qd) * 1024u, v);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  slab[slab_off_b2(DP) + tid] = gb2;
  slab[slab_off_b1(DP) + tid] = gb1;
  {  // head-bias / log_std gradients and loss sums: the four waves' running sums (LDS), added in fixed order
    LDS_BARRIER();
    if (tid < 32) {
      float b3s = 0.f, lss = 0.f;
      if (tid < 16) {
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          b3s += lds[L::GACC + w * 32 + tid];
          lss += lds[L::GACC + w * 32 + 16 + tid];
        }
      }
      slab[slab_off_b3(DP) + tid] = b3s;
      slab[slab_off_ls(DP) + tid] = lss;
    }
    if (tid < 4)
      slab[slab_off_st(DP) + tid] = (lds[L::STAT + tid] + lds[L::STAT + 4 + tid]) + (lds[L::STAT + 8 + tid] + lds[L::STAT + 12 + tid]);
  }
}

// rebuild every chain pack of both networks from the canonical parameters (set_params; k_adam_pack keeps them current per step)
struct ChainPackArgs {
  const float* W1[2]; const float* W2[2]; const float* W3[2];
  unsigned short* w1c[2]; unsigned short* w2c[2]; unsigned short* w2bc[2]; float* w3c[2]; float* w3bc[2];
  int D, Dp, head[2];
};
__global__ __launch_bounds__(256) void k_pack_chain(ChainPackArgs a) {
  const int net = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int K1 = (a.Dp + 31) / 32;
  if (i < FH * FH) {
    const int n = i / FH, k = i - n * FH;
    const float w = a.W2[net][i];
    chain_store_w2(a.w2c[net], a.w2bc[net], n, k, kTanhScale * w, w);
  }
  if (i < FH * 32 * K1) {   // every k slot of the layer-1 pack, padding included
    const int n = i / (32 * K1), k = i - n * (32 * K1);
    chain_store_w1(a.w1c[net], n, k, K1, k < a.D ? kTanhScale * a.W1[net][n * a.D + k] : 0.f);
  }
  if (i < 16 * FH) {
    const int a_ = i / FH, k = i - a_ * FH;
    const float w = a_ < a.head[net] ? a.W3[net][a_ * FH + k] : 0.f;
    a.w3c[net][chain_head_fwd_idx(a_, k)] = w;
    a.w3bc[net][chain_head_bwd_idx(a_, k)] = w;
  }
}

}  // namespace mobrob
