"""CPU checks (NumPy, fp64) of the algebra the Winograd / split-precision kernels implement
(ot-gan_amd/csrc/winograd.hip): transform identities, their adjoints, the polyphase decomposition of
the 5x5 stride-2 'SAME' convolution with its structural zeros, upsample folding, and the three-way
bf16 split.  No GPU, no reference needed: these pin the mathematics, the GPU tests pin the kernels."""
import numpy as np

Bt = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], float)
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], float)
At = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], float)


def corr_same(x, g):
    """3x3 'SAME' correlation of a 2-D array."""
    H, W = x.shape
    xp = np.pad(x, 1)
    return sum(g[i, j] * xp[i:i + H, j:j + W] for i in range(3) for j in range(3))


def test_f23_identity():
    rng = np.random.default_rng(0)
    d = rng.standard_normal((4, 4))
    g = rng.standard_normal((3, 3))
    Y = At @ ((G @ g @ G.T) * (Bt @ d @ Bt.T)) @ At.T
    ref = np.array([[np.sum(d[i:i + 3, j:j + 3] * g) for j in range(2)] for i in range(2)])
    assert np.allclose(Y, ref, atol=1e-13)


def test_transform_adjoints():
    """tf_output_adj / tf_filter_adj of winograd.hip are the transposes of tf_output / tf_filter."""
    rng = np.random.default_rng(1)
    M, dY = rng.standard_normal((4, 4)), rng.standard_normal((2, 2))
    assert np.isclose(np.sum((At @ M @ At.T) * dY), np.sum(M * (At.T @ dY @ At)))
    g, dU = rng.standard_normal((3, 3)), rng.standard_normal((4, 4))
    assert np.isclose(np.sum((G @ g @ G.T) * dU), np.sum(g * (G.T @ dU @ G)))


def s2_tap(parity, i):
    return 2 * i if parity else (-1 if i == 0 else 2 * i - 1)


def test_stride2_polyphase_and_structural_zeros():
    """y = 5x5 stride-2 SAME conv == sum over the four input-parity sub-images of a 3x3 SAME
    correlation with zero-padded 2-tap windows; G maps the zero tap to a vanishing frequency row."""
    rng = np.random.default_rng(2)
    H = 8
    x = rng.standard_normal((H, H))
    w = rng.standard_normal((5, 5))
    OH = H // 2
    xp = np.pad(x, ((1, 2), (1, 2)))                      # TF SAME: pad_before 1, pad_after 2
    direct = np.array([[np.sum(xp[2 * a:2 * a + 5, 2 * b:2 * b + 5] * w) for b in range(OH)] for a in range(OH)])
    total = np.zeros((OH, OH))
    for pi in (0, 1):
        for pj in (0, 1):
            sub = x[pi::2, pj::2]
            g = np.zeros((3, 3))
            for i in range(3):
                for j in range(3):
                    kh, kw = s2_tap(pi, i), s2_tap(pj, j)
                    if kh >= 0 and kw >= 0:
                        g[i, j] = w[kh, kw]
            total += corr_same(sub, g)
            U = G @ g @ G.T
            if pi == 0:
                assert np.all(U[0, :] == 0)               # forward orientation: index 0 vanishes
            if pj == 0:
                assert np.all(U[:, 0] == 0)
            Uf = G @ g[::-1, ::-1] @ G.T                   # flipped filters of dgrad: index 3 vanishes
            if pi == 0:
                assert np.all(Uf[3, :] == 0)
            if pj == 0:
                assert np.all(Uf[:, 3] == 0)
    assert np.allclose(total, direct, atol=1e-12)
    # 16 + 12 + 12 + 9 = 49 non-zero (class, frequency) blocks of 64
    present = sum((pi or fi != 0) and (pj or fj != 0) for pi in (0, 1) for pj in (0, 1) for fi in range(4) for fj in range(4))
    assert present == 49


def test_upsample_folding():
    """5x5 SAME conv of a 2x nearest-neighbour upsampled image == four 3x3 SAME correlations of the
    small image, one per output parity, with pre-summed taps."""
    rng = np.random.default_rng(3)
    H = 6
    x = rng.standard_normal((H, H))
    w = rng.standard_normal((5, 5))
    up = np.repeat(np.repeat(x, 2, 0), 2, 1)
    upp = np.pad(up, 2)
    direct = np.array([[np.sum(upp[a:a + 5, b:b + 5] * w) for b in range(2 * H)] for a in range(2 * H)])
    out = np.zeros_like(direct)
    for ph in (0, 1):
        for pw in (0, 1):
            g = np.zeros((3, 3))
            for kh in range(5):
                for kw in range(5):
                    dh = (ph + kh - 2) // 2 + 1            # small-image offset + 1
                    dw = (pw + kw - 2) // 2 + 1
                    g[dh, dw] += w[kh, kw]
            out[ph::2, pw::2] = corr_same(x, g)
    assert np.allclose(out, direct, atol=1e-12)


def bf16_rne(x):
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7fff + ((u >> 16) & 1)) >> 16 << 16
    return u.astype(np.uint32).view(np.float32)


def split3(x):
    x = np.asarray(x, np.float32)
    hi = bf16_rne(x)
    r1 = (x - hi).astype(np.float32)
    mid = bf16_rne(r1)
    r2 = (r1 - mid).astype(np.float32)
    return hi, mid, bf16_rne(r2)


def test_three_way_bf16_split():
    rng = np.random.default_rng(4)
    x = (rng.standard_normal(20000) * np.exp(rng.uniform(-8, 8, 20000))).astype(np.float32)
    hi, mid, lo = split3(x)
    rec = hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64)
    # three 8-bit pieces cover the 24-bit significand: the reconstruction is exact up to the last rounding
    assert np.max(np.abs(rec - x) / np.abs(x)) < 2.0 ** -24
    # six-term product sum against fp64: the dropped terms are O(2^-24) of the product
    K = 1024
    a = rng.standard_normal((8, K)).astype(np.float32)
    b = rng.standard_normal((8, K)).astype(np.float32)
    ah, am, al = [p.astype(np.float64) for p in split3(a)]
    bh, bm, bl = [p.astype(np.float64) for p in split3(b)]
    six = al @ bh.T + ah @ bl.T + am @ bm.T + am @ bh.T + ah @ bm.T + ah @ bh.T
    ref = a.astype(np.float64) @ b.astype(np.float64).T
    assert np.linalg.norm(six - ref) / np.linalg.norm(ref) < 2e-7
    three = am @ bh.T + ah @ bm.T + ah @ bh.T
    assert np.linalg.norm(three - ref) / np.linalg.norm(ref) > 1e-6      # why three terms are not enough
