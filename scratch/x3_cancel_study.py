"""Why the x3 products are ~2x worse than float32 BLAS on the planted-cancellation case (tests/test_x3_parity_gpu.py, kind "cancel";
VERDICT r4 #7), answered on the CPU in float64: the layer-2 pre-activations z2 = W2 . h1 of that case, evaluated
  (a) exactly (float64),
  (b) as the SIX kept bf16 piece products, every product and every sum exact (float64): what remains is the three dropped cross
      terms -- no accumulation order can recover them,
  (c) as eight products (the two 2^-24 cross terms p1 q2, p2 q1 added) and as all nine,
  (d) in float32 BLAS (sgemm) and as a sequential float32 fma chain in natural k order (what v_mfma_f32 does).
Errors are scaled by max |z2| like the test's.      python scratch/x3_cancel_study.py"""
import numpy as np

rng = np.random.default_rng(114)
H, B = 256, 2048


def bf16_rne(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def split3(x):
    p0 = bf16_rne(x)
    r1 = (x - p0).astype(np.float32)
    p1 = bf16_rne(r1)
    r2 = (r1 - p1).astype(np.float32)
    p2 = bf16_rne(r2)
    return [p.astype(np.float64) for p in (p0, p1, p2)]


h = np.tanh(rng.standard_normal((B, H // 2))).astype(np.float32)
h1 = np.empty((B, H), np.float32); h1[:, 0::2] = h; h1[:, 1::2] = h          # equal pairs
c = (rng.standard_normal((H, H // 2)) * 40).astype(np.float32)
w2 = np.empty((H, H), np.float32); w2[:, 0::2] = c; w2[:, 1::2] = -c * np.float32(1 + 2.0 ** -12)
exact = h1.astype(np.float64) @ w2.T.astype(np.float64)
scale = np.abs(exact).max()
err = lambda z: float(np.abs(z - exact).max() / scale)  # noqa: E731
a, b = split3(h1), split3(w2)
kept = [(0, 0), (0, 1), (1, 0), (0, 2), (2, 0), (1, 1)]
six = sum(a[i] @ b[j].T for i, j in kept)
eight = six + a[1] @ b[2].T + a[2] @ b[1].T
nine = eight + a[2] @ b[2].T
print(f"terms / result: {np.abs(h1.astype(np.float64)[:, None, :] * w2.astype(np.float64)[None, :8, :]).max() / scale:.0f}x")
print(f"six kept products, exact sums      {err(six):.2e}   <- the x3 scheme's floor on this data: the dropped cross terms")
print(f"eight products (+ p1q2, p2q1)      {err(eight):.2e}")
print(f"nine products                      {err(nine):.2e}")
print(f"float32 BLAS (sgemm)               {err((h1 @ w2.T).astype(np.float64)):.2e}")
acc = np.zeros((B, H), np.float32)
for k in range(H):                                   # sequential chain in natural k order (pairs are neighbours): float64 product, one float32 rounding per step = fma
    acc = (acc.astype(np.float64) + h1[:, k:k + 1].astype(np.float64) * w2[:, k][None, :].astype(np.float64)).astype(np.float32)
print(f"float32 fma chain, natural k order {err(acc.astype(np.float64)):.2e}")
# six products with ONE float32 rounding per 32-k MFMA and product, smallest terms first (the kernel's order of accumulation)
acc = np.zeros((B, H), np.float64)
for s in range(H // 32):
    sl = slice(32 * s, 32 * s + 32)
    for i, j in [(1, 1), (2, 0), (0, 2), (1, 0), (0, 1), (0, 0)]:
        acc = (acc + a[i][:, sl] @ b[j][:, sl].T).astype(np.float32).astype(np.float64)
print(f"six products, one f32 rounding per MFMA (kernel order)  {err(acc):.2e}")
acc = np.zeros((B, H), np.float64); lo = np.zeros((B, H), np.float64)
for s in range(H // 32):
    sl = slice(32 * s, 32 * s + 32)
    for i, j in [(1, 1), (2, 0), (0, 2)]:
        lo = (lo + a[i][:, sl] @ b[j][:, sl].T).astype(np.float32).astype(np.float64)
    for i, j in [(1, 0), (0, 1), (0, 0)]:
        acc = (acc + a[i][:, sl] @ b[j][:, sl].T).astype(np.float32).astype(np.float64)
print(f"six products, 2^-16 terms in a second accumulator added last  {err((acc + lo).astype(np.float32).astype(np.float64)):.2e}")

# ---- the same with the weights pre-scaled by 2 log2(e) and rounded to float32, as BOTH GPU pipes hold them (the tanh epilogue's
#      exp2 wants the scaled pre-activation; kernels_fused.h kTanhScale): the rounding of c s and of -c (1 + 2^-12) s breaks the
#      planted relation between the pair by 2^-24 of the terms -- 2^-12 of the result -- before any product is formed
s = np.float32(2.8853900817779268)
w2s = (w2 * s).astype(np.float32)
exact_s = h1.astype(np.float64) @ (w2.astype(np.float64) * float(s)).T
scale_s = np.abs(exact_s).max()
errs = lambda z: float(np.abs(z - exact_s).max() / scale_s)  # noqa: E731
bs = split3(w2s)
six_s = sum(a[i] @ bs[j].T for i, j in kept)
eight_s = six_s + a[1] @ bs[2].T + a[2] @ bs[1].T
print("with the weights scaled by 2 log2(e) and rounded to float32 first (both GPU pipes):")
print(f"  exact products of the rounded weights   {errs(h1.astype(np.float64) @ w2s.astype(np.float64).T):.2e}   <- shared by the f32 pipe and the x3 pipe")
print(f"  six kept products                       {errs(six_s):.2e}")
print(f"  eight products                          {errs(eight_s):.2e}")
acc = np.zeros((B, H), np.float32)
for kg in range(H // 8):                             # the f32 pipe's k order inside a fragment: k = 8 kg + 4 h + s, the two h halves on different lanes -> two chains added
    pass
acc = np.zeros((B, H), np.float32)
for k in range(H):
    acc = (acc.astype(np.float64) + h1[:, k:k + 1].astype(np.float64) * w2s[:, k][None, :].astype(np.float64)).astype(np.float32)
print(f"  float32 fma chain, natural k order      {errs(acc.astype(np.float64)):.2e}")
