"""Lane-level NumPy model of the register-chained x3 kernel's data movement (TEST INFRASTRUCTURE: csrc/kernels_chain.h).

`k_chain_train` computes every layer TRANSPOSED -- C[m][batch row] = sum_k W[m][k] * act[k][batch row] with the weights as the
MFMA A operand and the activations as the B operand -- so that a layer's accumulator tiles (output neuron in the registers /
lane groups, batch row on the lanes) ARE the next layer's B operand with no lane movement
(cdna_hip_programming.md, 'An accumulator tile as the next MFMA's operand').  The price is a permuted k order: k slot
(lane group g, element j) of k step s is input neuron n(s, g, j) = 32 s + 16 (j >> 2) + 4 g + (j & 3), and the weight packs
must be laid out in that order.  This module states the maps once, in plain NumPy, emulates `v_mfma_f32_16x16x32_bf16` /
`v_mfma_f32_16x16x4_f32` lane by lane, and lets tests check (a) the index algebra against a plain matrix product on the CPU
and (b) the device pack kernels against `pack_*` bit for bit on the GPU."""
import numpy as np

F32 = np.float32
LANES = 64


def bf16_rne(x):
    """float32 -> the float32 value of its bf16 rounding (round to nearest even; finite inputs)."""
    u = np.asarray(x, F32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return (r & 0xFFFFFFFF).astype(np.uint32).view(F32)


def split3(x):
    """x = p1 + p2 + p3 exactly, each the bf16 rounding of what the previous ones left (kernels_fused.h x3_split2)."""
    x = np.asarray(x, F32)
    p1 = bf16_rne(x)
    r1 = (x - p1).astype(F32)
    p2 = bf16_rne(r1)
    r2 = (r1 - p2).astype(F32)
    p3 = bf16_rne(r2)
    return p1, p2, p3


def kslot_neuron(s, g, j):
    """input neuron carried by k slot (lane group g, element j) of k step s of a chained layer."""
    return 32 * s + 16 * (j >> 2) + 4 * g + (j & 3)


def kslot_natural(s, g, j):
    """layer 1: the observation columns in their natural order."""
    return 32 * s + 8 * g + j


def pack_chain(W, ksteps, kmap, scale=1.0):
    """A operand pack of a chained layer: out[s][t][lane][j] = scale * W[16 t + (lane & 15)][kmap(s, lane >> 4, j)]
    (zero beyond the matrix).  W: [M][K] float32, M a multiple of 16."""
    M, K = W.shape
    out = np.zeros((ksteps, M // 16, LANES, 8), F32)
    for s in range(ksteps):
        for lane in range(LANES):
            for j in range(8):
                k = kmap(s, lane >> 4, j)
                if k < K:
                    out[s, :, lane, j] = (F32(scale) * W[(lane & 15)::16, k][: M // 16]).astype(F32)
    # W[(lane&15)::16] picks rows (lane&15), 16 + (lane&15), ...: tile t -> row 16 t + (lane & 15)
    return out


def pack_head_fwd(W3):
    """head forward (float32 16x16x4): out[t][lane][i] = W3[lane & 15][16 t + 4 (lane >> 4) + i], rows beyond the head = 0."""
    A, K = W3.shape
    out = np.zeros((K // 16, LANES, 4), F32)
    for t in range(K // 16):
        for lane in range(LANES):
            a = lane & 15
            if a < A:
                for i in range(4):
                    out[t, lane, i] = W3[a, 16 * t + 4 * (lane >> 4) + i]
    return out


def pack_head_bwd(W3):
    """dh2 (float32 16x16x4): out[t][lane][i] = W3[4 (lane >> 4) + i][16 t + (lane & 15)]."""
    A, K = W3.shape
    out = np.zeros((K // 16, LANES, 4), F32)
    for t in range(K // 16):
        for lane in range(LANES):
            for i in range(4):
                a = 4 * (lane >> 4) + i
                if a < A:
                    out[t, lane, i] = W3[a, 16 * t + (lane & 15)]
    return out


def mfma_16x16x32(a, b, c):
    """a, b: [64][8] (A[l & 15][8 (l >> 4) + j], B[8 (l >> 4) + j][l & 15]); c: [64][4] (C[4 (l >> 4) + r][l & 15])."""
    A = np.zeros((16, 32), np.float64)
    B = np.zeros((32, 16), np.float64)
    for l in range(LANES):
        A[l & 15, 8 * (l >> 4):8 * (l >> 4) + 8] = a[l]
        B[8 * (l >> 4):8 * (l >> 4) + 8, l & 15] = b[l]
    C = A @ B
    out = np.array(c, np.float64)
    for l in range(LANES):
        out[l] += C[4 * (l >> 4):4 * (l >> 4) + 4, l & 15]
    return out


def mfma_16x16x4(a, b, c):
    """a, b: [64] (A[l & 15][l >> 4], B[l >> 4][l & 15]); c: [64][4]."""
    A = np.zeros((16, 4), np.float64)
    B = np.zeros((4, 16), np.float64)
    for l in range(LANES):
        A[l & 15, l >> 4] = a[l]
        B[l >> 4, l & 15] = b[l]
    C = A @ B
    out = np.array(c, np.float64)
    for l in range(LANES):
        out[l] += C[4 * (l >> 4):4 * (l >> 4) + 4, l & 15]
    return out


def acc_to_matrix(acc):
    """accumulator tiles acc[t][lane][r] -> matrix [16 T][16 batch rows]."""
    T = acc.shape[0]
    out = np.zeros((16 * T, 16), np.float64)
    for t in range(T):
        for l in range(LANES):
            out[16 * t + 4 * (l >> 4):16 * t + 4 * (l >> 4) + 4, l & 15] = acc[t, l]
    return out


def b_frag_from_acc(acc, s):
    """B operand of k step s from the previous layer's accumulator tiles: element j of a lane = register j & 3 of tile 2 s + (j >> 2)."""
    b = np.zeros((LANES, 8), np.float64)
    for j in range(8):
        b[:, j] = acc[2 * s + (j >> 2), :, j & 3]
    return b


def b_frag_from_rows(X, s):
    """layer 1: lane (batch row l & 15, group g) holds X[row][32 s + 8 g .. + 7] (zero beyond the row)."""
    b = np.zeros((LANES, 8), np.float64)
    for l in range(LANES):
        for j in range(8):
            k = 32 * s + 8 * (l >> 4) + j
            if k < X.shape[1]:
                b[l, j] = X[l & 15, k]
    return b


def chained_layer(pack, bfrag):
    """acc[t] = sum_s A(s, t) . B(s) for all m tiles (float64: the map, not the rounding, is what this checks)."""
    S, T = pack.shape[:2]
    acc = np.zeros((T, LANES, 4), np.float64)
    for s in range(S):
        b = bfrag(s)
        for t in range(T):
            acc[t] = mfma_16x16x32(pack[s, t], b, acc[t])
    return acc


def image_addr(m, row):
    """float offset of element (column m, batch row) in a swizzled transposed image [256][64] (kernels_chain.h img_addr)."""
    return m * 64 + (((row >> 2) ^ (m & 15)) << 2) + (row & 3)
