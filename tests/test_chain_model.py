"""CPU: the index algebra of the register-chained kernel (tests/chain_model.py) against plain matrix products."""
import numpy as np
import pytest

from tests import chain_model as M


@pytest.mark.parametrize("D,A", [(58, 12), (14, 2), (26, 1)])
def test_chained_layers_equal_plain_matrix_products(D, A):
    rng = np.random.default_rng(D)
    H = 256
    Dp = 16 * ((D + 15) // 16)
    K1 = (Dp + 31) // 32
    X = rng.standard_normal((16, D))
    W1, W2, W3 = rng.standard_normal((H, D)), rng.standard_normal((H, H)), rng.standard_normal((A, H))
    # forward: layer 1 (natural k order), layer 2 (chained k order), head (float32 16x16x4, one register of a tile per step)
    acc1 = M.chained_layer(M.pack_chain(W1, K1, M.kslot_natural),
                           lambda s: M.b_frag_from_rows(X, s))
    z1 = W1 @ X.T
    assert np.allclose(M.acc_to_matrix(acc1), z1, atol=1e-4 * np.abs(z1).max())
    h1 = np.tanh(acc1)
    acc2 = M.chained_layer(M.pack_chain(W2, 8, M.kslot_neuron), lambda s: M.b_frag_from_acc(h1, s))
    z2 = W2 @ np.tanh(z1)
    assert np.allclose(M.acc_to_matrix(acc2), z2, atol=1e-4 * np.abs(z2).max())
    h2 = np.tanh(acc2)
    hp = M.pack_head_fwd(W3)
    mean = np.zeros((M.LANES, 4))
    for t in range(16):
        for i in range(4):
            mean = M.mfma_16x16x4(hp[t, :, i], h2[t, :, i], mean)
    ref = W3 @ np.tanh(z2)                                  # [A][16 rows]
    got = M.acc_to_matrix(mean[None])                       # [16][16]: head row a, batch row
    assert np.allclose(got[:A], ref, atol=1e-4 * np.abs(ref).max()) and not np.any(got[A:])
    # backward: dh2 = W3^T dout (k slot g of step i = head row 4 g + i), dh1 = W2^T dz2 (chained k order over the neurons of layer 2)
    dout = rng.standard_normal((A, 16))
    dacc = np.zeros((M.LANES, 4))
    for l in range(M.LANES):
        for i in range(4):
            a = 4 * (l >> 4) + i
            dacc[l, i] = dout[a, l & 15] if a < A else 0.0
    bp = M.pack_head_bwd(W3)
    dh2 = np.zeros((16, M.LANES, 4))
    for t in range(16):
        for i in range(4):
            dh2[t] = M.mfma_16x16x4(bp[t, :, i], dacc[:, i], dh2[t])
    ref = W3.T @ dout
    assert np.allclose(M.acc_to_matrix(dh2), ref, atol=1e-4 * np.abs(ref).max())
    dz2 = dh2 * (1 - h2 * h2)
    dh1 = M.chained_layer(M.pack_chain(np.ascontiguousarray(W2.T), 8, M.kslot_neuron), lambda s: M.b_frag_from_acc(dz2, s))
    ref = W2.T @ M.acc_to_matrix(dz2)
    assert np.allclose(M.acc_to_matrix(dh1), ref, atol=1e-4 * np.abs(ref).max())


def test_split3_is_exact_and_image_addresses_are_a_bijection():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(4096) * np.exp(rng.uniform(-20, 20, 4096))).astype(np.float32)
    p1, p2, p3 = M.split3(x)
    assert np.array_equal((p1.astype(np.float64) + p2 + p3).astype(np.float32), x)
    for p in (p1, p2, p3):
        assert not np.any(p.view(np.uint32) & 0xFFFF)          # every piece is a bf16 value
    addr = np.array([[M.image_addr(m, r) for r in range(64)] for m in range(256)])
    assert np.array_equal(np.sort(addr.ravel()), np.arange(256 * 64))
    # a lane's eight consecutive batch rows of one column are two aligned 16-byte chunks (what the weight-gradient loops read)
    for m in (0, 5, 77, 255):
        for r0 in range(0, 64, 8):
            a = addr[m, r0:r0 + 8]
            assert np.array_equal(a[:4], a[0] + np.arange(4)) and np.array_equal(a[4:], a[4] + np.arange(4)) and a[0] % 4 == 0 and a[4] % 4 == 0
