"""CPU checks (NumPy, fp64) of the algebra the Winograd / split-precision kernels implement
(ot-gan_amd/csrc/winograd.hip): transform identities, their adjoints, the polyphase decomposition of
the 5x5 stride-2 'SAME' convolution with its structural zeros, upsample folding, F(4x4,3x3) and the three-way
bf16 split.  No GPU, no reference needed: these pin the mathematics, the GPU tests pin the kernels."""
import numpy as np

# F(4x4, 3x3) with the interpolation points {0, 1, -1, 1/2, -2, inf} (winograd.hip: bt1 / g1 / at1)
Bt = np.array([[1, -1.5, -2, 1.5, 1, 0], [0, -1, .5, 2.5, 1, 0], [0, 1, -2.5, .5, 1, 0],
               [0, -2, -1, 2, 1, 0], [0, .5, -1, -.5, 1, 0], [0, 1, -1.5, -2, 1.5, 1]], float)
G = np.array([[1, 0, 0], [1 / 3, 1 / 3, 1 / 3], [-1 / 3, 1 / 3, -1 / 3], [-16 / 15, -8 / 15, -4 / 15],
              [1 / 15, -2 / 15, 4 / 15], [0, 0, 1]], float)
At = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, .5, -2, 0], [0, 1, 1, .25, 4, 0], [0, 1, -1, .125, -8, 1]], float)
WA, WM = 6, 4


def corr_same(x, g):
    """3x3 'SAME' correlation of a 2-D array."""
    H, W = x.shape
    xp = np.pad(x, 1)
    return sum(g[i, j] * xp[i:i + H, j:j + W] for i in range(3) for j in range(3))


def test_f43_identity():
    rng = np.random.default_rng(0)
    d = rng.standard_normal((WA, WA))
    g = rng.standard_normal((3, 3))
    Y = At @ ((G @ g @ G.T) * (Bt @ d @ Bt.T)) @ At.T
    ref = np.array([[np.sum(d[i:i + 3, j:j + 3] * g) for j in range(WM)] for i in range(WM)])
    assert np.allclose(Y, ref, atol=1e-12)
    # the data and output transforms are dyadic: exact in fp32 (only the filter transform has thirds / fifteenths)
    for Mx in (Bt, At):
        assert np.all(Mx * 8 == np.round(Mx * 8))


def test_f43_point_set_accuracy():
    """Why {0, 1, -1, 1/2, -2, inf}: with exact products and fp32 accumulation (what the split-precision GEMM
    delivers) the relative L2 error against fp64 stays below 1e-6 -- under a plain fp32 MFMA chain of the direct
    convolution (1.3e-6 measured on the GPU), about 3x F(2x2,3x3) on the same GEMM."""
    rng = np.random.default_rng(5)
    K = 512
    f32 = np.float32
    errs = []
    for _ in range(3):
        d = rng.standard_normal((K, WA, WA)).astype(f32)
        g = (rng.standard_normal((K, 3, 3)) * 0.05).astype(f32)
        U = np.einsum("ij,kjl,ml->kim", G, g.astype(float), G).astype(f32)       # fp64 filter transform, stored fp32
        V = np.stack([((Bt.astype(f32) @ d[k]).astype(f32) @ Bt.T.astype(f32)).astype(f32) for k in range(K)])
        prod = U.astype(float) * V.astype(float)
        M = np.zeros((WA, WA), f32)
        for k0 in range(0, K, 16):
            M = (M.astype(float) + prod[k0:k0 + 16].sum(0)).astype(f32)
        Y = ((At.astype(f32) @ M).astype(f32) @ At.T.astype(f32)).astype(f32)
        ref = np.array([[np.sum(d[:, i:i + 3, j:j + 3].astype(float) * g.astype(float)) for j in range(WM)]
                        for i in range(WM)])
        errs.append(np.linalg.norm(Y - ref) / np.linalg.norm(ref))
    assert np.mean(errs) < 1.2e-6, errs


def test_transform_adjoints():
    """tf_output_adj / tf_filter_adj of winograd.hip are the transposes of tf_output / tf_filter."""
    rng = np.random.default_rng(1)
    M, dY = rng.standard_normal((WA, WA)), rng.standard_normal((WM, WM))
    assert np.isclose(np.sum((At @ M @ At.T) * dY), np.sum(M * (At.T @ dY @ At)))
    g, dU = rng.standard_normal((3, 3)), rng.standard_normal((WA, WA))
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
            Uf = G @ g[::-1, ::-1] @ G.T                   # flipped filters of dgrad: the LAST index vanishes
            if pi == 0:
                assert np.all(Uf[WA - 1, :] == 0)
            if pj == 0:
                assert np.all(Uf[:, WA - 1] == 0)
    assert np.allclose(total, direct, atol=1e-12)
    # 36 + 30 + 30 + 25 = 121 non-zero (class, frequency) blocks of 144
    present = sum((pi or fi != 0) and (pj or fj != 0) for pi in (0, 1) for pj in (0, 1) for fi in range(WA) for fj in range(WA))
    assert present == 121


def test_lpt_frequency_order_is_a_permutation_longest_first():
    """gemm_x3.h lpt_frequency(): the forward GEMMs of a strided layer dispatch their full-K frequencies first."""
    n = WA - 1
    order = []
    for z in range(WA * WA):
        if z < n * n:
            f = (1 + z // n) * WA + 1 + z % n
        elif z < n * n + n:
            f = 1 + (z - n * n)
        elif z < n * n + 2 * n:
            f = (1 + (z - n * n - n)) * WA
        else:
            f = 0
        order.append(f)
    assert sorted(order) == list(range(WA * WA))
    classes = [sum((pi or f // WA != 0) and (pj or f % WA != 0) for pi in (0, 1) for pj in (0, 1)) for f in order]
    assert classes == sorted(classes, reverse=True) and classes[0] == 4 and classes[-1] == 1


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
