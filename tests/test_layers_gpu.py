"""GPU parity tests of the layer kernels: HIP conv2d / dense (fwd, dgrad, wgrad through
torch.autograd), weight norm, GLU, tanh, feature head and the optimiser steps, against the
plain-PyTorch restatement of the reference layers (oracle/nets_torch.py) evaluated in fp64
on the CPU.  Tolerance: relative L2 error <= 2e-5 (fp32 MFMA fmaf chains vs fp64)."""
import numpy as np
import os

import pytest
import torch

from oracle import nets_torch as NT

pytestmark = pytest.mark.gpu
TOL = 2e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from otgan_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


def _rel(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


CONV_CASES = [
    # name, N, H, W, segs, Cout, k, stride, upsample, preact
    ("s1_plain", 2, 8, 8, (16,), 32, 5, 1, False, None),
    ("s2_crelu", 3, 16, 16, (16,), 48, 5, 2, False, "crelu"),
    ("up_plain", 2, 4, 4, (32,), 64, 5, 1, True, None),
    ("list_crelu_narrow", 2, 8, 8, (8, 4, 4), 16, 3, 1, False, "crelu"),
    ("list_crelu_s2", 2, 8, 8, (16, 16, 16), 24, 3, 2, False, "crelu"),
    ("up_crelu", 2, 4, 4, (16, 16), 16, 3, 1, True, "crelu"),
    ("rgb_in", 2, 16, 16, (3,), 128, 5, 1, False, None),
    ("rgb_out", 2, 16, 16, (128,), 3, 5, 1, False, None),
    ("celu", 2, 8, 8, (8, 8), 32, 3, 1, False, "celu"),
    ("elu", 2, 8, 8, (16,), 32, 3, 1, False, "elu"),
    ("relu_s2", 2, 8, 8, (16,), 32, 3, 2, False, "relu"),
    ("big_s2_crelu", 4, 16, 16, (64,), 256, 5, 2, False, "crelu"),
    ("big_wgrad_split", 8, 32, 32, (32,), 64, 3, 1, False, "crelu"),
    ("dcgan_like_up", 2, 8, 8, (64,), 128, 5, 1, True, None),
    ("ragged_cout", 2, 8, 8, (16,), 40, 3, 1, False, "crelu"),
    ("scalar_cin", 2, 8, 8, (6,), 20, 3, 1, False, "crelu"),
    # folded 5x5 upsampling layers without pre-activation: Winograd F(2x2,3x3) path
    ("wino_ragged", 3, 8, 8, (96,), 24, 5, 1, True, None),
    ("wino_wide", 5, 16, 16, (32,), 160, 5, 1, True, None),
    ("wino_split_k", 16, 16, 16, (64,), 32, 5, 1, True, None),
    # 5x5 stride-2 layers: four 3x3 sub-convolutions in Winograd form (single-tensor inputs)
    ("wino_s2_plain", 2, 8, 8, (32,), 24, 5, 2, False, None),
    ("wino_s2_elu", 2, 8, 8, (32,), 16, 5, 2, False, "elu"),
    ("wino_s2_celu", 2, 16, 16, (16,), 40, 5, 2, False, "celu"),
    ("wino_s2_relu", 3, 4, 4, (64,), 132, 5, 2, False, "relu"),
    ("wino_s2_split_k", 24, 16, 16, (32,), 32, 5, 2, False, "crelu"),
    ("list_s2_k5_generic", 2, 8, 8, (16, 16), 32, 5, 2, False, "crelu"),
    # weight gradient from forward-layout operands (t-leading GEMM): 32 tiles = two K stages (unpipelined kernel),
    # and a ragged last K split
    ("wino_s2_tl_short", 2, 16, 16, (32,), 32, 5, 2, False, None),
    ("wino_tl_short", 2, 8, 8, (32,), 32, 5, 1, True, None),
    ("wino_s2_tl_crelu", 10, 16, 16, (32,), 64, 5, 2, False, "crelu"),
    # DenseNet growth layers (3x3 -> 16 channels): the LDS-free dense16 kernels
    ("dense16_list", 2, 8, 8, (32, 16, 16), 16, 3, 1, False, "crelu"),
    ("dense16_tail", 3, 8, 8, (24, 16), 16, 3, 1, False, "crelu"),
    ("dense16_plain", 2, 16, 16, (40,), 16, 3, 1, False, None),
    ("dense16_celu", 2, 8, 8, (16, 8), 16, 3, 1, False, "celu"),
    ("dense16_relu", 1, 4, 4, (8,), 16, 3, 1, False, "relu"),
    # list elements that are not multiples of 4 channels wide: per-channel gathers
    ("odd_list_crelu", 2, 8, 8, (22, 8, 6), 32, 3, 1, False, "crelu"),
    ("odd_single_crelu", 2, 8, 8, (18,), 32, 3, 1, False, "crelu"),
    ("odd_list_dense16", 2, 8, 8, (22, 10), 16, 3, 1, False, "celu"),
    ("dense16_many_tiles", 130, 32, 32, (16,), 16, 3, 1, False, "crelu"),
    ("dense16_mid_tiles", 260, 16, 16, (24,), 16, 3, 1, False, "crelu"),
    # few-channel weight gradients, all taps per block (conv_outer2_kernel): several units per block, H < 8, 3x3
    ("rgb_in_k3", 2, 16, 16, (3,), 64, 3, 1, False, None),
    ("rgb_out_k3_elu", 3, 8, 8, (32,), 3, 3, 1, False, "elu"),
    ("rgb_out_many", 72, 32, 32, (64,), 3, 5, 1, False, None),
    ("rgb_in_many", 72, 32, 32, (3,), 32, 5, 1, False, None),
    ("few_out_small_crelu", 2, 4, 4, (16,), 2, 5, 1, False, "crelu"),
    ("few_out_list", 2, 8, 8, (16, 8), 3, 3, 1, False, "crelu"),
    # RGB-out input gradient on the streaming kernel (conv_fewout_dgrad_kernel): list input with interleaved (+c, -c)
    # order and a ragged last 64-channel chunk, single activations, 5x5 with a doubled ELU, rectangular image
    ("rgb_out_list_crelu", 3, 16, 16, (40, 16, 24), 3, 3, 1, False, "crelu"),
    ("rgb_out_elu_rect", 2, 16, 32, (72,), 3, 3, 1, False, "elu"),
    ("rgb_out_celu_k5", 2, 16, 16, (36,), 3, 5, 1, False, "celu"),
    ("rgb_out_relu", 2, 16, 16, (20,), 3, 3, 1, False, "relu"),
    # RGB-out forward / RGB-in input gradient on the fp32 matrix pipe (conv_fewout_mfma_kernel): channels a multiple of
    # 64, rows of 16 / 32 / 64 pixels; one and a half chunks of 128 channels, list input through the channel map
    ("rgb_out_mfma_two_chunks", 2, 16, 32, (192,), 3, 5, 1, False, None),
    ("rgb_out_mfma_list_crelu", 2, 16, 16, (32, 16, 8, 8), 3, 3, 1, False, "crelu"),
    ("rgb_out_mfma_celu_192", 2, 8, 16, (96,), 3, 3, 1, False, "celu"),
    ("rgb_out_mfma_w64_elu", 1, 8, 64, (64,), 3, 3, 1, False, "elu"),
    ("rgb_in_mfma_dgrad", 3, 16, 32, (3,), 64, 5, 1, False, None),
    ("two_out_mfma", 2, 8, 16, (64,), 2, 5, 1, False, "relu"),
    # DenseNet transition shapes: outputs just above 128 / 192 columns take one exact column tile (128x160, 128x224)
    ("wide160_s2_list", 128, 32, 32, (32, 16, 16), 144, 3, 2, False, "crelu"),
    ("wide224_up_list", 32, 16, 16, (32, 16, 16), 208, 3, 1, True, "crelu"),
    ("wide160_up_celu", 32, 16, 16, (32,), 144, 3, 1, True, "celu"),
    # Cout % 16 != 0 (DenseNet transitions 200, 228): float4 dgrad gathers with a zero-filled last K tile per tap
    ("cout200_s2_list", 4, 16, 16, (32, 16, 16), 200, 3, 2, False, "crelu"),
    ("cout228_s2", 3, 8, 8, (24,), 228, 3, 2, False, "crelu"),
    ("cout20_plain", 2, 8, 8, (16,), 20, 3, 1, False, None),
    # wide 3x3 stride-1 layers: Winograd F(4x4,3x3) with one class (the block-input convolution of a dense block)
    ("wino_plain3_crelu", 6, 16, 16, (48,), 128, 3, 1, False, "crelu"),
    ("wino_plain3_none", 3, 8, 8, (64,), 160, 3, 1, False, None),
    ("wino_plain3_celu_rect", 5, 8, 16, (32,), 256, 3, 1, False, "celu"),
    ("wino_plain3_list_generic", 2, 8, 8, (32, 16), 128, 3, 1, False, "crelu"),
    ("wino_plain3_k_pad", 4, 8, 8, (40,), 128, 3, 1, False, "crelu"),          # 80 effective channels: K padded to 96
    ("wino_plain3_k_pad_none", 2, 16, 16, (48,), 160, 3, 1, False, None),       # 48 -> 64
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv2d(dev, case):
    from otgan_amd import ops
    name, N, H, W, segs, Cout, k, stride, up, pre = case
    gen = torch.Generator().manual_seed(sum(map(ord, name)))
    C = sum(segs)
    mult = 2 if pre in ("crelu", "celu") else 1
    xs64 = [torch.randn(N, H, W, s, generator=gen, dtype=torch.float64) for s in segs]
    V64 = torch.randn(k, k, C * mult, Cout, generator=gen, dtype=torch.float64) * 0.05
    g64 = torch.rand(Cout, generator=gen, dtype=torch.float64) + 0.5
    b64 = torch.randn(Cout, generator=gen, dtype=torch.float64) * 0.1
    # the HIP path sees fp32 roundings of these; evaluate the oracle on the same values
    xs64 = [x.float().double().requires_grad_(True) for x in xs64]
    V64, g64, b64 = [t.float().double().requires_grad_(True) for t in (V64, g64, b64)]
    y_ref = NT.conv2d(xs64, {"V": V64, "g": g64, "b": b64}, pre, stride, up)
    dy64 = torch.randn(y_ref.shape, generator=gen, dtype=torch.float64).float().double()
    grads_ref = torch.autograd.grad(y_ref, xs64 + [V64, g64, b64], dy64)
    dx_ref = torch.cat(grads_ref[:len(segs)], 3)

    x = torch.cat([t.detach().float() for t in xs64], 3).to(dev).requires_grad_(True)
    V = V64.detach().float().to(dev).requires_grad_(True)
    g = g64.detach().float().to(dev).requires_grad_(True)
    b = b64.detach().float().to(dev).requires_grad_(True)
    y = ops.conv2d_op(x, V, g, b, stride=stride, upsample=up, preact=ops.ACT[pre], segs=segs)
    assert y.shape == y_ref.shape
    assert _rel(y, y_ref) < TOL, "forward"
    dx, dV, dg, db = torch.autograd.grad(y, [x, V, g, b], dy64.float().to(dev))
    assert _rel(dx, dx_ref) < TOL, "dgrad"
    assert _rel(dV, grads_ref[len(segs)]) < TOL, "wgrad/dV"
    assert _rel(dg, grads_ref[len(segs) + 1]) < TOL, "dg"
    assert _rel(db, grads_ref[len(segs) + 2]) < TOL, "db"


def test_conv2d_only_needed_grads(dev):
    # generator step: dgrad through the critic without touching its weights (train.py:112)
    from otgan_amd import ops
    x = torch.randn(2, 8, 8, 16, device=dev, requires_grad=True)
    V = torch.randn(3, 3, 32, 32, device=dev) * 0.05
    g = torch.ones(32, device=dev)
    b = torch.zeros(32, device=dev)
    y = ops.conv2d_op(x, V, g, b, preact=ops.ACT["crelu"])
    (dx,) = torch.autograd.grad(y, [x], torch.ones_like(y))
    assert dx.shape == x.shape and torch.isfinite(dx).all()


def test_prepared_filters_match_inline(dev):
    # Winograd-domain filters made once (otgan_conv2d_prepare_filters_f32) == derived inside the call, bit for bit;
    # layers without a Winograd path report no filters; a short buffer is refused
    import ctypes
    from otgan_amd import ops, _lib
    x = torch.randn(4, 16, 16, 32, device=dev)
    V2d = (torch.randn(5 * 5 * 64, 48, device=dev) * 0.05).contiguous()
    g = torch.ones(48, device=dev)
    w, wT, _ = ops.weightnorm_fwd(V2d, g)
    desc = ops.make_desc(x, 32, False, 5, 5, 2, 48, 48, 0, ops.ACT["crelu"])
    filt = ops.prepare_filters(desc, 0, wT)
    assert filt is not None
    ya, yb = torch.empty(4, 8, 8, 48, device=dev), torch.empty(4, 8, 8, 48, device=dev)
    ops.conv_fwd_raw(desc, x, None, wT, None, ya)
    ops.conv_fwd_raw(desc, x, None, wT, None, yb, filt)
    assert torch.equal(ya, yb)
    dy = torch.randn_like(ya)
    fb = ops.prepare_filters(desc, 1, w)
    da, db = torch.empty_like(x), torch.empty_like(x)
    ops.conv_dgrad_raw(desc, dy, w, x, None, da, 32, False)
    ops.conv_dgrad_raw(desc, dy, w, x, None, db, 32, False, fb)
    assert torch.equal(da, db)
    L = _lib.lib()
    rc = L.otgan_conv2d_prepare_filters_f32(ctypes.byref(desc), 0, wT.data_ptr(), filt.data_ptr(), 16, _lib.stream_ptr())
    assert rc == -2 and b"too small" in L.otgan_last_error()
    # folded (upsampling) layer: filters from the un-folded weights (which = 2 / 3) == from the folded ones (0 / 1)
    xu = torch.randn(2, 8, 8, 32, device=dev)
    Vu = (torch.randn(5 * 5 * 32, 64, device=dev) * 0.05).contiguous()
    wu, wuT, _ = ops.weightnorm_fwd(Vu, torch.ones(64, device=dev))
    du = ops.make_desc(xu, 32, True, 5, 5, 1, 64, 64, 0, 0)
    weff, weffT = ops.fold_weights(du, wu)
    # (the two carry different power-of-two scales -- the folded tensor's own largest magnitude against four times the
    # un-folded one's bound -- so the buffers differ; what they produce agrees to the rounding of the 22-bit pieces)
    yf, yu = torch.empty(2, 16, 16, 64, device=dev), torch.empty(2, 16, 16, 64, device=dev)
    ops.conv_fwd_raw(du, xu, None, weffT, None, yf, ops.prepare_filters(du, 0, weffT))
    ops.conv_fwd_raw(du, xu, None, wuT, None, yu, ops.prepare_filters(du, 2, wuT))
    assert _rel(yu, yf) < 1e-6
    dyu = torch.randn_like(yf)
    dxf, dxu = torch.empty_like(xu), torch.empty_like(xu)
    ops.conv_dgrad_raw(du, dyu, weff, xu, None, dxf, 32, False, ops.prepare_filters(du, 1, weff))
    ops.conv_dgrad_raw(du, dyu, wu, xu, None, dxu, 32, False, ops.prepare_filters(du, 3, wu))
    assert _rel(dxu, dxf) < 1e-6
    assert L.otgan_conv2d_filter_bytes(ctypes.byref(desc), 2) == 0      # strided layers have no folded form
    plain = ops.make_desc(torch.empty(2, 8, 8, 16, device=dev), 16, False, 3, 3, 1, 32, 32, 0, 0)
    assert ops.prepare_filters(plain, 0, wT) is None and L.otgan_conv2d_filter_bytes(ctypes.byref(plain), 2) == 0


@pytest.mark.parametrize("pre", [None, "crelu"])
def test_dense(dev, pre):
    from otgan_amd import ops
    gen = torch.Generator().manual_seed(3)
    N, Cin, Cout = 8, 100, 96
    mult = 2 if pre == "crelu" else 1
    x64 = (torch.rand(N, Cin, generator=gen, dtype=torch.float64) * 2 - 1).float().double().requires_grad_(True)
    V64 = (torch.randn(Cin * mult, Cout, generator=gen, dtype=torch.float64) * 0.05).float().double().requires_grad_(True)
    g64 = (torch.rand(Cout, generator=gen, dtype=torch.float64) + 0.5).float().double().requires_grad_(True)
    b64 = (torch.randn(Cout, generator=gen, dtype=torch.float64) * 0.1).float().double().requires_grad_(True)
    y_ref = NT.dense(x64, {"V": V64, "g": g64, "b": b64}, pre)
    dy = torch.randn(y_ref.shape, generator=gen, dtype=torch.float64).float().double()
    gref = torch.autograd.grad(y_ref, [x64, V64, g64, b64], dy)
    x, V, g, b = [t.detach().float().to(dev).requires_grad_(True) for t in (x64, V64, g64, b64)]
    y = ops.dense_op(x, V, g, b, preact=ops.ACT[pre])
    assert _rel(y, y_ref) < TOL
    got = torch.autograd.grad(y, [x, V, g, b], dy.float().to(dev))
    for a, r, n in zip(got, gref, "dx dV dg db".split()):
        assert _rel(a, r) < TOL, n


def test_weightnorm_epsilon_and_transpose(dev):
    from otgan_amd import ops
    V = torch.randn(75, 20, device=dev) * 0.05
    V[:, 3] = 0.0                                 # all-zero direction: epsilon under the max (nn.py:176)
    g = torch.rand(20, device=dev) + 0.5
    w, wT, inv = ops.weightnorm_fwd(V, g)
    ref = NT.weight_norm(V.double().cpu(), g.double().cpu())
    assert _rel(w, ref) < 1e-6
    assert torch.equal(wT, w.t().contiguous())
    assert torch.all(w[:, 3] == 0)


def test_glu_tanh_head(dev):
    from otgan_amd import ops
    gen = torch.Generator().manual_seed(5)
    x64 = torch.randn(3, 4, 4, 32, generator=gen, dtype=torch.float64).float().double().requires_grad_(True)
    y_ref = NT.glu(x64, 3)
    dy = torch.randn(y_ref.shape, generator=gen, dtype=torch.float64).float().double()
    (dx_ref,) = torch.autograd.grad(y_ref, [x64], dy)
    x = x64.detach().float().to(dev).requires_grad_(True)
    y = ops.glu(x)
    assert _rel(y, y_ref) < 1e-6
    (dx,) = torch.autograd.grad(y, [x], dy.float().to(dev))
    assert _rel(dx, dx_ref) < 1e-6
    # dense-style GLU: split along axis 1 of [B, 2C]
    x2 = torch.randn(4, 64, device=dev)
    assert _rel(ops.glu(x2), NT.glu(x2.double().cpu(), 1)) < 1e-6
    # tanh
    x64 = torch.randn(1000, generator=gen, dtype=torch.float64).float().double().requires_grad_(True)
    t_ref = torch.tanh(x64)
    (dt_ref,) = torch.autograd.grad(t_ref, [x64], torch.ones_like(t_ref))
    xt = x64.detach().float().to(dev).requires_grad_(True)
    t = ops.tanh(xt)
    (dt,) = torch.autograd.grad(t, [xt], torch.ones_like(t))
    assert _rel(t, t_ref) < 1e-6 and _rel(dt, dt_ref) < 1e-5
    # feature head
    x64 = torch.randn(5, 4, 4, 24, generator=gen, dtype=torch.float64).float().double().requires_grad_(True)
    f_ref = NT.feature_head(x64)
    df = torch.randn(f_ref.shape, generator=gen, dtype=torch.float64).float().double()
    (dxh_ref,) = torch.autograd.grad(f_ref, [x64], df)
    xh = x64.detach().float().to(dev).requires_grad_(True)
    f = ops.feature_head(xh)
    assert f.shape == (5, 4 * 4 * 48)
    assert _rel(f, f_ref) < 1e-6
    np.testing.assert_allclose(f.detach().norm(dim=1).cpu().numpy(), 1.0, atol=1e-6)
    (dxh,) = torch.autograd.grad(f, [xh], df.float().to(dev))
    assert _rel(dxh, dxh_ref) < 1e-5


def test_optimiser_steps(dev):
    from otgan_amd import ops
    gen = torch.Generator().manual_seed(9)
    n = 10007
    p0 = torch.randn(n, generator=gen, dtype=torch.float64)
    # Adam: three steps, shared t starting at 1 (nn.py:56,72), eps inside sqrt (nn.py:68)
    p_ref = p0.clone()
    st = {"t": 1.0, "v": torch.zeros(n, dtype=torch.float64), "mg": torch.zeros(n, dtype=torch.float64)}
    p = p0.float().to(dev)
    v = torch.zeros(n, device=dev)
    mg = torch.zeros(n, device=dev)
    for t in range(1, 4):
        gr = torch.randn(n, generator=gen, dtype=torch.float64).float().double()
        st["t"] = float(t)
        p_ref = NT.adam_update(p_ref, gr, st, -3e-4, 0.5, 0.999)     # critic: lr = -lr (train.py:143)
        ops.adam_step(p, gr.float().to(dev), v, mg, -3e-4, 0.5, 0.999, t)
    assert _rel(p, p_ref) < 1e-6
    assert _rel(v, st["v"]) < 1e-6 and _rel(mg, st["mg"]) < 1e-6
    # Adamax
    p_ref = p0.clone()
    st = {"v": torch.zeros(n, dtype=torch.float64), "mg": torch.zeros(n, dtype=torch.float64)}
    p = p0.float().to(dev); v.zero_(); mg.zero_()
    for _ in range(2):
        gr = torch.randn(n, generator=gen, dtype=torch.float64).float().double()
        p_ref = NT.adamax_update(p_ref, gr, st, 3e-4, 0.5, 0.999)
        ops.adamax_step(p, gr.float().to(dev), v, mg, 3e-4, 0.5, 0.999)
    assert _rel(p, p_ref) < 1e-6
    # Nesterov
    p_ref = p0.clone()
    st = {"v": torch.zeros(n, dtype=torch.float64)}
    p = p0.float().to(dev); v.zero_()
    for _ in range(2):
        gr = torch.randn(n, generator=gen, dtype=torch.float64).float().double()
        p_ref = NT.nesterov_update(p_ref, gr, st, 1e-2, 0.5)
        ops.nesterov_step(p, gr.float().to(dev), v, 1e-2, 0.5)
    assert _rel(p, p_ref) < 1e-6
    # EMA (train.py:63)
    sh = torch.zeros(n, device=dev)
    ops.ema_update(sh, p, 0.999)
    assert _rel(sh, 0.001 * p.double().cpu()) < 1e-6


def test_no_cpu_fallback_layers():
    from otgan_amd import _lib, ops
    with pytest.raises(_lib.OtganError):
        ops.glu(torch.zeros(2, 4))


FULL_SIZE = [  # name, H, C, Cout, k, stride, up, pre  -- the six heavy DCGAN layers at BASELINE batch 256
    ("D.conv1", 32, 128, 256, 5, 2, False, "crelu"),
    ("D.conv2", 16, 256, 512, 5, 2, False, "crelu"),
    ("D.conv3", 8, 512, 1024, 5, 2, False, "crelu"),
    ("G.conv0", 4, 1024, 1024, 5, 1, True, None),
    ("G.conv1", 8, 512, 512, 5, 1, True, None),
    ("G.conv2", 16, 256, 256, 5, 1, True, None),
]


@pytest.mark.parametrize("case", FULL_SIZE, ids=[c[0] for c in FULL_SIZE])
def test_full_size_adjoint_identities(dev, case):
    """Size-independent properties at the BASELINE problem size (no oracle can run there in
    seconds): a convolution is bilinear, so with y = conv(x, W) + b
        <y(x, W) - y(0-input), dy> = <x_eff, dgrad(dy)>-type identities hold; we check
        (1) <conv(x, W) - b, dy> == <W, wgrad(x, dy)>          (linearity in W)
        (2) for the layers without pre-activation also == <x, dgrad(dy)>   (linearity in x)
        (3) conv(x, W1 + W2) == conv(x, W1) + conv(x, W2)
    in fp32 with fp64 reductions; they tie the forward, dgrad and wgrad kernels (Winograd transforms,
    split-precision GEMMs, structural-zero skipping) to one another."""
    from otgan_amd import ops
    name, H, C, Cout, k, stride, up, pre = case
    gen = torch.Generator(device="cpu").manual_seed(sum(map(ord, name)))
    B = 256
    mult = 2 if pre == "crelu" else 1
    x = torch.randn(B, H, H, C, generator=gen).to(dev).requires_grad_(True)
    V = (torch.randn(k, k, C * mult, Cout, generator=gen) * 0.05).to(dev)
    g = torch.ones(Cout, device=dev)
    b = torch.zeros(Cout, device=dev)
    # weight norm makes W a nonlinear function of V: differentiate w.r.t. g-scaled direction instead by
    # feeding the normalised weight through V with g = ||V|| (then W == V and dW == dV + radial part).
    # Simpler and exact: use the raw launchers underneath the autograd function.
    V2d = V.view(-1, Cout).contiguous()
    w, wT, _ = ops.weightnorm_fwd(V2d, V2d.norm(dim=0))          # w == V up to rounding
    N_, H_, W_, _ = x.shape
    OH, OW = ops.out_hw(H_, W_, up, stride)
    desc = ops.make_desc(x, C, up, k, k, stride, Cout, Cout, 0, ops.ACT[pre])
    from otgan_amd import _lib
    L = _lib.lib()
    import ctypes
    folded = L.otgan_conv2d_folded_weight_elems(ctypes.byref(desc)) > 0

    def prepared(wmat, wmatT):
        if not folded:
            return wmat, wmatT
        return ops.fold_weights(desc, wmat)

    def fwd(wmat, wmatT):
        y = torch.empty((N_, OH, OW, Cout), device=dev)
        _, wt = prepared(wmat, wmatT)
        ops.conv_fwd_raw(desc, x.detach(), None, wt, b, y)
        return y

    y = fwd(w, wT)
    dy = torch.randn(y.shape, generator=gen).to(dev)
    dw = torch.empty_like(w)
    ops.conv_wgrad_raw(desc, x.detach(), None, dy, dw)
    lhs = float((y.double() * dy.double()).sum())
    rhs_w = float((w.double() * dw.double()).sum())
    assert abs(lhs - rhs_w) <= 2e-4 * max(abs(lhs), 1.0), ("wgrad", lhs, rhs_w)
    if pre is None:
        dx = torch.empty_like(x)
        wd, _ = prepared(w, wT)
        ops.conv_dgrad_raw(desc, dy, wd, x.detach(), None, dx, C, False)
        rhs_x = float((x.detach().double() * dx.double()).sum())
        assert abs(lhs - rhs_x) <= 2e-4 * max(abs(lhs), 1.0), ("dgrad", lhs, rhs_x)
    w2 = (torch.randn(w.shape, generator=gen) * 0.05).to(dev)
    w2T = w2.t().contiguous()
    ysum = fwd(w + w2, (w + w2).t().contiguous())
    y2 = fwd(w2, w2T)
    err = float((ysum - y - y2).norm() / ysum.norm())
    # three independent fp32 evaluations: each carries the F(4x4,3x3) rounding (~7e-7 .. 2e-6 relative at K ~ 10^4)
    assert err < 1e-5, ("additivity", err)


# ---- scaled two-piece fp16 operands of the Winograd GEMMs: the scale follows the data ----------------------
def _wino_layer(dev, x, V, dy, stride, up, pre):
    from otgan_amd import ops
    x = x.to(dev).requires_grad_(True)
    V = V.to(dev).requires_grad_(True)
    Cout = V.shape[-1]
    g = torch.ones(Cout, device=dev, requires_grad=True)
    b = torch.zeros(Cout, device=dev, requires_grad=True)
    y = ops.conv2d_op(x, V, g, b, stride=stride, upsample=up, preact=ops.ACT[pre])
    dx, dV = torch.autograd.grad(y, [x, V], dy.to(dev))
    return y, dx, dV


@pytest.mark.parametrize("kind", ["up", "s2_crelu"])
@pytest.mark.parametrize("mag", [1e-12, 1.0, 1e12])
def test_wino_scale_invariance(dev, kind, mag):
    """Magnitudes far outside the fp16 range on both sides (activations AND gradients scaled by `mag`): the
    per-tensor power-of-two scales make the result exactly covariant when `mag` is a power of two, and the
    accuracy against fp64 independent of it."""
    up, stride, pre = (True, 1, None) if kind == "up" else (False, 2, "crelu")
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(4, 8, 8, 64, generator=gen)
    V = torch.randn(5, 5, 64 * (2 if pre else 1), 32, generator=gen) * 0.05
    y0, dx0, dV0 = _wino_layer(dev, x, V, torch.randn(4, 16 if up else 4, 16 if up else 4, 32, generator=torch.Generator().manual_seed(3)), stride, up, pre)
    p2 = float(2.0 ** round(float(torch.log2(torch.tensor(mag)))))           # the power of two next to mag
    dy = torch.randn(4, 16 if up else 4, 16 if up else 4, 32, generator=torch.Generator().manual_seed(3))
    y1, dx1, dV1 = _wino_layer(dev, x * p2, V, dy * p2, stride, up, pre)
    assert torch.equal(y1, y0 * p2)                  # exact: every scale moved by the same power of two
    assert torch.equal(dx1, dx0 * p2)
    assert _rel(dV1, dV0 * (p2 * p2)) < 1e-6         # (weight-norm backward renormalises: not bitwise)
    assert torch.isfinite(y1).all() and torch.isfinite(dx1).all() and torch.isfinite(dV1).all()


def test_wino_outlier_zero_and_nan(dev):
    """One element 1e4 times the rest sets the scale of its tensor: the other outputs keep their accuracy; an
    all-zero input gives an exactly zero output (scale 1 for amax = 0); a NaN in the input makes the output NaN."""
    from otgan_amd import ops
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(2, 8, 8, 64, generator=gen)
    V = (torch.randn(5, 5, 64, 32, generator=gen) * 0.05)
    x_out = x.clone()
    x_out[0, 0, 0, 0] = 1.0e4
    g, b = torch.ones(32), torch.zeros(32)
    ref = NT.conv2d([x_out.double()], {"V": V.double(), "g": g.double(), "b": b.double()}, None, 1, True)
    y = ops.conv2d_op(x_out.to(dev), V.to(dev), g.to(dev), b.to(dev), stride=1, upsample=True, preact=0)
    assert _rel(y[1], ref[1]) < TOL                  # the image without the outlier: unaffected
    far = ref[0, 8:, 8:]                              # outputs of the outlier's image its 5x5 window does not reach
    assert _rel(y[0, 8:, 8:], far) < TOL
    assert _rel(y, ref) < TOL
    yz = ops.conv2d_op(torch.zeros_like(x).to(dev), V.to(dev), g.to(dev), b.to(dev), stride=1, upsample=True, preact=0)
    assert torch.count_nonzero(yz) == 0
    x_nan = x.clone()
    x_nan[1, 3, 3, 5] = float("nan")
    yn = ops.conv2d_op(x_nan.to(dev), V.to(dev), g.to(dev), b.to(dev), stride=1, upsample=True, preact=0)
    assert torch.isnan(yn).any()


@pytest.mark.parametrize("case", [
    ("crelu_list", 3, 16, 16, (32, 16), 8, "crelu"),
    ("crelu_single", 2, 8, 8, (64,), 9, "crelu"),
    ("celu_list", 2, 8, 16, (16, 8, 8), 8, "celu"),
    ("elu_single", 2, 8, 8, (64,), 8, "elu"),
    ("crelu_halves", 2, 8, 8, (32, 16), 16, "crelu"),     # 16 layers: the first 8 outputs enter the last 8 layers as one 128 -> 128 convolution
    ("elu_halves", 1, 8, 8, (64,), 16, "elu"),
    ("crelu_k_pad", 2, 8, 8, (24, 16), 16, "crelu"),       # 80 effective input channels (the 8x8 critic block has 400)
    # 20 layers = 18 own-chain layers, more than one otgan_dense16_prepare_* call takes (16) -- the block must
    # stay off the fp16 chain kernels instead of raising in the forward pass
    ("crelu_20_layers", 1, 8, 8, (32,), 20, "crelu"),
], ids=lambda c: c[0])
def test_dense_block_split_matches_chain(dev, case, monkeypatch):
    """A dense block computed as "block-input convolution (Winograd) + growth chain" (ops.DenseBlockFunction) against
    the fp64 oracle's plain chain of convolutions over the growing concatenation (reference nn.py:243-262), and
    against the same op with the split switched off."""
    from otgan_amd import ops
    name, N, H, W, segs0, L, pre = case
    F = 16
    gen = torch.Generator().manual_seed(sum(map(ord, name)))
    mult = 2 if pre in ("crelu", "celu") else 1
    C0 = sum(segs0)
    xs64 = [torch.randn(N, H, W, c, generator=gen, dtype=torch.float64).float().double().requires_grad_(True) for c in segs0]
    P64 = []
    for k in range(L):
        V = (torch.randn(3, 3, (C0 + k * F) * mult, F, generator=gen, dtype=torch.float64) * 0.05).float().double()
        g = (torch.rand(F, generator=gen, dtype=torch.float64) + 0.5).float().double()
        b = (torch.randn(F, generator=gen, dtype=torch.float64) * 0.1).float().double()
        P64.append([t.requires_grad_(True) for t in (V, g, b)])
    feats = list(xs64)
    for V, g, b in P64:
        feats.append(NT.conv2d(feats, {"V": V, "g": g, "b": b}, pre, 1, False))
    y_ref = torch.cat(feats, 3)
    dy64 = torch.randn(y_ref.shape, generator=gen, dtype=torch.float64).float().double()
    flat64 = [t for p in P64 for t in p]
    grads_ref = torch.autograd.grad(y_ref, xs64 + flat64, dy64)

    def run(split):
        monkeypatch.setattr(ops, "DENSE_SPLIT", bool(split))
        ops.bump_weights_epoch()
        x0 = torch.cat([t.detach().float() for t in xs64], 3).to(dev).requires_grad_(True)
        params = [[t.detach().float().to(dev).requires_grad_(True) for t in p] for p in P64]
        y = ops.dense_block_op(x0, segs0, params, 3, ops.ACT[pre])
        grads = torch.autograd.grad(y, [x0] + [t for p in params for t in p], dy64.float().to(dev))
        return y, grads

    y_s, g_s = run(True)
    y_c, g_c = run(False)
    assert _rel(y_s, y_ref) < TOL and _rel(y_c, y_ref) < TOL
    dx_ref = torch.cat(grads_ref[:len(segs0)], 3)
    assert _rel(g_s[0], dx_ref) < TOL and _rel(g_c[0], dx_ref) < TOL
    for i, (a, c, r) in enumerate(zip(g_s[1:], g_c[1:], grads_ref[len(segs0):])):
        assert _rel(a, r) < TOL, f"split: gradient of parameter {i}"
        assert _rel(c, r) < TOL, f"chain: gradient of parameter {i}"


def test_y_accumulate_contract(dev):
    """otgan_conv_desc::y_accumulate: honoured by the 16-output growth layers and the wide 3x3 layers on the Winograd
    path (y += conv + bias, bit-for-bit the separate sum up to one fp32 addition), rejected everywhere else."""
    import ctypes
    from otgan_amd import _lib, ops
    from otgan_amd._lib_layers import ConvDesc
    gen = torch.Generator().manual_seed(5)
    for C, Cout in ((32, 16), (64, 128)):
        x = torch.randn(4, 8, 8, C, generator=gen).to(dev)
        wT = (torch.randn(Cout, 9 * 2 * C, generator=gen) * 0.05).to(dev)
        b = torch.randn(Cout, generator=gen).to(dev)
        y0 = torch.randn(4, 8, 8, Cout, generator=gen).to(dev)
        desc = ConvDesc(4, 8, 8, C, C, 0, 3, 3, 1, Cout, Cout, 0, ops.ACT["crelu"], 1)
        plain = torch.empty_like(y0)
        ops.conv_fwd_raw(desc, x, None, wT, b, plain)
        acc = y0.clone()
        desc.y_accumulate = 1
        ops.conv_fwd_raw(desc, x, None, wT, b, acc)
        assert _rel(acc, (y0 + plain).double()) < 1e-6
    # a 5x5 layer, and a 3x3 layer too narrow for the Winograd path: rejected, output untouched
    for C, Cout, k in ((32, 64, 5), (16, 32, 3)):
        x = torch.randn(2, 8, 8, C, generator=gen).to(dev)
        wT = torch.randn(Cout, k * k * C, generator=gen).to(dev)
        y = torch.zeros(2, 8, 8, Cout, device=dev)
        desc = ConvDesc(2, 8, 8, C, C, 0, k, k, 1, Cout, Cout, 0, 0, 1)
        desc.y_accumulate = 1
        with pytest.raises(_lib.OtganError):
            ops.conv_fwd_raw(desc, x, None, wT, None, y)
        assert float(y.abs().max()) == 0.0


@pytest.mark.parametrize("case", [
    ("folded_up", 8, 8, 8, 64, 64, 5, 1, True, None),
    ("strided_crelu", 8, 16, 16, 32, 64, 5, 2, False, "crelu"),
    ("plain3_crelu", 8, 8, 8, 64, 128, 3, 1, False, "crelu"),
    ("up3_crelu", 8, 8, 8, 32, 32, 3, 1, True, "crelu"),
], ids=lambda c: c[0])
def test_shared_x_operand(dev, case):
    """otgan_conv_desc::x_operand: the weight gradient that reads the forward pass's transformed input back is
    bit-identical to the one that transforms x itself; the forward result does not depend on where the operand lives."""
    import ctypes
    from otgan_amd import _lib, ops
    from otgan_amd._lib_layers import ConvDesc
    name, N, H, W, C, Cout, k, stride, up, pre = case
    gen = torch.Generator().manual_seed(sum(map(ord, name)))
    mult = 2 if pre == "crelu" else 1
    x = torch.randn(N, H, W, C, generator=gen).to(dev)
    V = (torch.randn(k * k * C * mult, Cout, generator=gen) * 0.05).to(dev)
    g = (torch.rand(Cout, generator=gen) + 0.5).to(dev)
    b = torch.randn(Cout, generator=gen).to(dev)
    w, wT, _ = ops.weightnorm_fwd(V, g)
    OH, OW = ops.out_hw(H, W, up, stride)
    dy = torch.randn(N, OH, OW, Cout, generator=gen).to(dev)

    def run(shared):
        desc = ConvDesc(N, H, W, C, C, 1 if up else 0, k, k, stride, Cout, Cout, 0, ops.ACT[pre], 1)
        unfolded = _lib.lib().otgan_conv2d_filter_bytes(ctypes.byref(desc), 2) > 0
        wT_use = wT
        if up and not unfolded:
            _, wT_use = ops.fold_weights(desc, w)
        filt = ops.prepare_filters(desc, 2 if unfolded else 0, wT if unfolded else wT_use)
        buf = ops.shared_x_operand(desc, dev) if shared else None
        if shared and buf is None:
            # the fp32 engine (OTGAN_WINO_FP32=1) has different operand layouts in the two passes, the direct engine
            # (OTGAN_DISABLE_WINOGRAD=1) no transformed operand at all: nothing to share
            assert any(os.environ.get(k) for k in ("OTGAN_WINO_FP32", "OTGAN_DISABLE_WINOGRAD")), \
                "this layer's passes were expected to share their operand"
            pytest.skip("operand sharing is off in this engine mode")
        y = torch.empty(N, OH, OW, Cout, device=dev)
        ops.conv_fwd_raw(desc, x, None, wT_use, b, y, filt)
        dw = torch.empty_like(V)
        ops.conv_wgrad_raw(desc, x, None, dy, dw)
        return y, dw

    y0, dw0 = run(False)
    y1, dw1 = run(True)
    assert torch.equal(y0, y1) and torch.equal(dw0, dw1)
    # a list input (channel map) cannot take the Winograd path: with x_operand given the call is refused
    if pre == "crelu" and not up:
        desc = ConvDesc(N, H, W, C, C, 0, k, k, stride, Cout, Cout, 0, ops.ACT[pre], 1)
        buf = ops.shared_x_operand(desc, dev)
        cmap, _ = ops.channel_maps((C // 2, C // 2), ops.ACT[pre], dev)
        with pytest.raises(_lib.OtganError):
            ops.conv_fwd_raw(desc, x, cmap, wT, b, torch.empty(N, OH, OW, Cout, device=dev))


def test_adam_gathered_gradients_and_fused_ema(dev):
    """otgan_adam_step_gather_f32: the flat-buffer Adam step reading one gradient tensor per variable, with the EMA of
    the updated weights in the same launch, is BIT-IDENTICAL to concatenating the gradients, otgan_adam_step_f32 and
    otgan_ema_update_f32 (nn.py:50-73, train.py:63-64,223)."""
    from otgan_amd import ops
    g = torch.Generator().manual_seed(21)
    sizes = [(5, 5, 8, 16), (16,), (16,), (100, 64), (3,), (3,)]
    offs = [0]
    for sz in sizes:
        offs.append(offs[-1] + int(np.prod(sz)))
    n = offs[-1]
    p0 = torch.randn(n, generator=g).to(dev)
    grads = [torch.randn(sz, generator=g).to(dev) * 10 ** float(torch.randn((), generator=g)) for sz in sizes]
    for mom1 in (0.5, 0.0):
        pa, pb = p0.clone(), p0.clone()
        va = torch.zeros(n, device=dev) if mom1 > 0 else None
        vb = torch.zeros(n, device=dev) if mom1 > 0 else None
        mga, mgb = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        sha, shb = p0.clone(), p0.clone()
        for t in (1.0, 2.0, 3.0):
            ops.adam_step_gather(pa, grads, offs, va, mga, 3e-4, mom1, 0.999, t, sha, 0.999)
            ops.adam_step(pb, torch.cat([x.reshape(-1) for x in grads]), vb, mgb, 3e-4, mom1, 0.999, t)
            ops.ema_update(shb, pb, 0.999)
        assert torch.equal(mga, mgb), float((mga - mgb).abs().max())
        if mom1 > 0:
            assert torch.equal(va, vb), float((va - vb).abs().max())
        assert torch.equal(pa, pb), (float((pa - pb).abs().max()), int((pa != pb).sum()))
        assert torch.equal(sha, shb), (float((sha - shb).abs().max()), int((sha != shb).sum()))
    with pytest.raises(Exception):
        ops.adam_step_gather(pa, grads * 6, list(range(37)), va, mga, 3e-4, 0.5, 0.999, 1.0)     # > 32 segments


# ---- DenseNet at the BASELINE batch (configs[3]: 256 images per GPU) -------------------------------------------------
DENSE_FULL = [  # name, H, C0 (segments), preact: one dense block per resolution of the critic / generator (models/densenet.py:11-21,60-73)
    ("critic_32x32", 32, (32,), "crelu"),
    ("critic_16x16", 16, (144,), "crelu"),
    ("critic_8x8", 8, (200,), "crelu"),
    ("generator_16x16", 16, (144, 16), "crelu"),
    ("critic_32x32_celu", 32, (32,), "celu"),          # --nonlinearity celu: smooth, so the GRADIENTS can be held tight too
    ("critic_16x16_celu", 16, (144,), "celu"),
    ("generator_16x16_celu", 16, (144, 16), "celu"),
]


@pytest.mark.parametrize("case", DENSE_FULL, ids=[c[0] for c in DENSE_FULL])
def test_full_size_dense_block_split_matches_chain(dev, case, monkeypatch):
    """A 16-layer dense block at 256 images, where no oracle finishes in seconds: the block computed as wide Winograd
    convolutions of finished channel groups + short growth chains (the default; its GEMMs take the tile-count dependent
    branches of the bench -- 256 x 128 tiles, K padding, K splits of the weight gradients, batched weight norm) against
    the SAME block as a plain chain of 16-output convolutions on the dense16 kernels (ops.DENSE_SPLIT = False): two
    different algorithms, kernels and summation orders for every output, input gradient and weight gradient.  Each is
    pinned to the fp64 oracle at small sizes (test_dense_block_split_matches_chain)."""
    from otgan_amd import ops
    name, H, segs0, pre = case
    B, L, F = 256, 16, 16
    gen = torch.Generator().manual_seed(sum(map(ord, name)))
    mult = 2 if pre in ("crelu", "celu") else 1
    C0 = sum(segs0)
    x = torch.randn(B, H, H, C0, generator=gen)
    P = []
    for k in range(L):
        P.append(((torch.randn(3, 3, (C0 + k * F) * mult, F, generator=gen) * 0.05), torch.rand(F, generator=gen) + 0.5,
                  torch.randn(F, generator=gen) * 0.1))
    dy = torch.randn(B, H, H, C0 + L * F, generator=gen)

    def run(split):
        monkeypatch.setattr(ops, "DENSE_SPLIT", bool(split))
        ops.bump_weights_epoch()
        x0 = x.to(dev).requires_grad_(True)
        params = [[t.to(dev).requires_grad_(True) for t in p] for p in P]
        y = ops.dense_block_op(x0, segs0, params, 3, ops.ACT[pre])
        grads = torch.autograd.grad(y, [x0] + [t for p in params for t in p], dy.to(dev))
        return y.detach(), grads

    y_s, g_s = run(True)
    y_c, g_c = run(False)
    assert _rel(y_s, y_c) < 2e-5
    # CReLU: of the ~1e8 pre-activations of the block a few dozen sit within fp32 rounding of zero and land on either
    # side in the two evaluation orders; each flips one derivative mask (measured 9e-5 .. 5e-4 on the input gradient).
    # The smooth CELU cases hold the gradients as tight as the outputs.
    gtol = 5e-5 if pre == "celu" else 2e-3
    assert _rel(g_s[0], g_c[0]) < gtol, "input gradient"
    worst = max((_rel(a, c), i) for i, (a, c) in enumerate(zip(g_s[1:], g_c[1:])))
    assert worst[0] < gtol, f"parameter gradient {worst}"


TRANS_FULL = [  # name, H, C, Cout, stride, up, pre: the DenseNet transitions (models/densenet.py:17-21,67-73) at 256 images
    ("critic_t1", 32, 288, 144, 2, False, "crelu"),
    ("critic_t2", 16, 400, 200, 2, False, "crelu"),
    ("critic_t3", 8, 456, 228, 2, False, "crelu"),
    ("generator_up1", 8, 288, 144, 1, True, "crelu"),
    ("generator_up2", 16, 416, 208, 1, True, "crelu"),
]


@pytest.mark.parametrize("case", TRANS_FULL, ids=[c[0] for c in TRANS_FULL])
def test_full_size_transition_identities(dev, case):
    """The DenseNet transition layers at 256 images: y is linear in W whatever the pre-activation, so
    <conv(x, W) - b, dy> == <W, wgrad(x, dy)> and conv(x, W1 + W2) == conv(x, W1) + conv(x, W2) tie the forward and
    weight-gradient kernels of the shapes the bench runs (stride-2 implicit GEMMs with 128 x 160 / 128 x 224 tiles and K
    splits; 3x3 on the upsampled grid through the one-class Winograd passes) to one another; the input gradient is tied
    to the forward pass by Euler's identity for the degree-1 homogeneous CReLU, <x, dgrad(dy)> == <y, dy> (zero bias)."""
    from otgan_amd import ops
    name, H, C, Cout, stride, up, pre = case
    B, k = 256, 3
    gen = torch.Generator().manual_seed(sum(map(ord, name)))
    x = torch.randn(B, H, H, C, generator=gen).to(dev)
    V = (torch.randn(k, k, 2 * C, Cout, generator=gen) * 0.05).to(dev)
    V2 = (torch.randn(k, k, 2 * C, Cout, generator=gen) * 0.05).to(dev)
    ones, zeros = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)

    def conv(xx, VV, grad=False):
        # g = ||V|| per output column makes the normalised weight W == V (weight norm is then the identity map)
        gn = VV.reshape(-1, Cout).norm(dim=0)
        return ops.conv2d_op(xx, VV, gn, zeros, stride=stride, upsample=up, preact=ops.ACT[pre])

    xg = x.clone().requires_grad_(True)
    Vg = V.clone().requires_grad_(True)
    gn = V.reshape(-1, Cout).norm(dim=0).detach()
    y = ops.conv2d_op(xg, Vg, gn, zeros, stride=stride, upsample=up, preact=ops.ACT[pre])
    dy = torch.randn(y.shape, generator=gen).to(dev)
    dx, dV = torch.autograd.grad(y, [xg, Vg], dy)
    # dV is the gradient through weight norm: W = g V / ||V|| with g = ||V||  =>  <V, dV> = 0 (radial part removed) and
    # the tangential part equals that of dW.  Linearity in W: <y, dy> = <W, dW>; recover dW from the raw launcher instead.
    V2d = V.reshape(-1, Cout).contiguous()
    w, wT, _ = ops.weightnorm_fwd(V2d, gn)
    assert _rel(w, V2d) < 1e-6
    lhs = float((y.detach().double() * dy.double()).sum())
    # the inner products are net sums of ~1e8 terms of either sign: their fp32 rounding scales with sum |y dy|, not with the net
    scale = float((y.detach().double().abs() * dy.double().abs()).sum())
    # <W, dW> through the chain rule of weight norm: dV = (g/||V||) (dW - V <V, dW>/||V||^2) = dW - V <V,dW>/||V||^2 per column
    # => <V, dV> = 0 and <W, dW> = sum_c <V_c, dW_c> is not recoverable from dV alone; use dg instead: dg_c = <V_c, dW_c>/||V_c||
    gvar = gn.clone().requires_grad_(True)
    y2 = ops.conv2d_op(x, V, gvar, zeros, stride=stride, upsample=up, preact=ops.ACT[pre])
    (dg,) = torch.autograd.grad(y2, [gvar], dy)
    rhs = float((dg.double() * gn.double()).sum())           # sum_c g_c dg_c = <W, dW>
    assert abs(lhs - rhs) <= 2e-6 * scale, ("wgrad / weight norm", lhs, rhs, scale)
    # additivity in the weights (g = column norms of the SUM for all three, so that W is exactly V, V2, V + V2)
    with torch.no_grad():
        ya = conv(x, V)
        yb = conv(x, V2)
        ys = conv(x, V + V2)
    assert float((ys - ya - yb).norm() / ys.norm()) < 1e-5
    # input gradient: CReLU is positively homogeneous of degree 1, so is y(x) (zero bias): Euler's identity
    # <x, dy/dx . dy> == <y, dy> ties dgrad (activation derivative, class / tap bookkeeping) to the forward pass exactly
    rhs_x = float((x.double() * dx.double()).sum())
    assert abs(lhs - rhs_x) <= 2e-6 * scale, ("dgrad", lhs, rhs_x, scale)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(3, 4, 4, 64, 64), (2, 8, 8, 32, 48), (4, 16, 16, 64, 128)], ids=["4x4", "8x8", "16x16"])
def test_glu_in_output_transform(dev, shape):
    """The generator's 5x5 upsampling layers write the gated linear unit of their output themselves (conv2d_op
    glu_hint, otgan_conv_desc::glu_out; reference models/dcgan.py:35-36, 50): same pre-activation, gated product,
    amax record and gradients as the separate glu launch, bit for bit."""
    from otgan_amd import _lib, ops
    import ctypes
    N, H, W, C, Cout = shape
    gen = torch.Generator().manual_seed(11)
    x0 = torch.randn(N, H, W, C, generator=gen).to(dev)
    V0 = (torch.randn(5, 5, C, Cout, generator=gen) * 0.05).to(dev)
    g0, b0 = (1 + 0.1 * torch.randn(Cout, generator=gen)).to(dev), (0.1 * torch.randn(Cout, generator=gen)).to(dev)
    dz = torch.randn(N, 2 * H, 2 * W, Cout // 2, generator=gen).to(dev)
    desc = ops.make_desc(x0, C, True, 5, 5, 1, Cout, Cout, 0, 0)
    assert _lib.lib().otgan_conv2d_glu_fused(ctypes.byref(desc)) == 1
    res = []
    for hint in (False, True):
        x, V, g, b = (t.clone().requires_grad_(True) for t in (x0, V0, g0, b0))
        y = ops.conv2d_op(x, V, g, b, stride=1, upsample=True, preact=0, glu_hint=hint)
        assert hasattr(y, "_otgan_glu") == hint
        z = ops.glu(y)
        assert not hasattr(y, "_otgan_glu")
        rec = ops.amax_of(z)
        assert rec is not None
        z.backward(dz)
        res.append((y.detach(), z.detach(), rec.max().clone(), x.grad, V.grad, g.grad, b.grad))
    for a, c in zip(*res):
        assert torch.equal(a, c)
    assert float(res[1][2]) == float(res[1][1].abs().max())
    # a layer the fused form does not cover rejects glu_out instead of ignoring it
    d2 = ops.make_desc(x0, C, False, 3, 3, 1, Cout, Cout, 0, 0)
    assert _lib.lib().otgan_conv2d_glu_fused(ctypes.byref(d2)) == 0


@pytest.mark.gpu
def test_dense_block_gradient_buffer_in_place(dev, monkeypatch):
    """DenseBlockFunction.backward accumulates the earlier slices' gradients into the INCOMING gradient tensor when nothing
    else references it (an engine-made tensor), and into a copy otherwise: a caller's `gradient=` tensor stays intact, a
    gradient that torch's add hands to two blocks at once is not corrupted, and both paths give the same numbers."""
    from otgan_amd import ops
    gen = torch.Generator().manual_seed(21)
    N, H, W, C0, L, F = 2, 8, 8, 32, 8, 16
    x0 = torch.randn(N, H, W, C0, generator=gen).to(dev)
    mk = lambda: [[(torch.randn(3, 3, (C0 + k * F) * 2, F, generator=gen) * 0.05).to(dev), (torch.rand(F, generator=gen) + 0.5).to(dev),
                   (torch.randn(F, generator=gen) * 0.1).to(dev)] for k in range(L)]
    PA, PB = mk(), mk()
    w = torch.randn(N, H, W, C0 + L * F, generator=gen).to(dev)

    def run(inplace):
        monkeypatch.setattr(ops, "_GRAD_INPLACE", inplace)
        ops.bump_weights_epoch()
        xa, xb = x0.clone().requires_grad_(True), (0.5 * x0).requires_grad_(True)
        pa = [[t.clone().requires_grad_(True) for t in p] for p in PA]
        pb = [[t.clone().requires_grad_(True) for t in p] for p in PB]
        ya = ops.dense_block_op(xa, (C0,), pa, 3, ops.ACT["crelu"])
        yb = ops.dense_block_op(xb, (C0,), pb, 3, ops.ACT["crelu"])
        # torch's add passes ONE gradient tensor to both blocks; the multiply makes it an engine-made tensor
        loss = ((ya + yb) * w).sum() + (ya * ya).sum()
        leaves = [xa, xb] + [t for p in pa + pb for t in p]
        return torch.autograd.grad(loss, leaves)

    g_in, g_cp = run(True), run(False)
    for a, c in zip(g_in, g_cp):
        assert torch.equal(a, c)
    # a caller-owned gradient is never written to
    monkeypatch.setattr(ops, "_GRAD_INPLACE", True)
    xa = x0.clone().requires_grad_(True)
    pa = [[t.clone().requires_grad_(True) for t in p] for p in PA]
    ya = ops.dense_block_op(xa, (C0,), pa, 3, ops.ACT["crelu"])
    gout = w.clone()
    ya.backward(gradient=gout)
    assert torch.equal(gout, w)
    # a gradient that a tensor hook keeps alive has the same `_use_count()` as an un-retained one -- only the
    # Python reference count shows it.  The retained tensor must hold the block's INCOMING gradient after the pass
    # (2 * w * ya here), not the accumulated one.
    kept = []
    xa = x0.clone().requires_grad_(True)
    pa = [[t.clone().requires_grad_(True) for t in p] for p in PA]
    ya = ops.dense_block_op(xa, (C0,), pa, 3, ops.ACT["crelu"])
    ya.register_hook(lambda g: kept.append(g))
    expect = (2.0 * w * ya).detach().clone()
    torch.autograd.grad(((ya * w) * ya).sum(), [xa] + [t for p in pa for t in p])
    assert len(kept) == 1 and torch.allclose(kept[0], expect, rtol=1e-6, atol=1e-6)


@pytest.mark.gpu
def test_unfolded_weight_gradient_in_the_adjoint_filter_transform(tmp_path):
    """The adjoint filter transform of the 5x5 upsampling layers writes the un-folded weight gradient itself
    (wino_filter_adj_unfold5_kernel: the four parity classes in adjacent lanes, added in class order): the same sums as
    the adjoint transform followed by conv.hip's unfold_wgrad_kernel (OTGAN_WINO_UNFOLD_FUSED=0; read once per process)
    -- equal to 1 - 2 ulp, not bit for bit (the compiler contracts the transform's last multiply-adds differently in the
    two kernels), and bit-reproducible from run to run."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for i, v in enumerate(("1", "1", "0")):
        e = dict(os.environ)
        e["OTGAN_WINO_UNFOLD_FUSED"] = v
        f = str(tmp_path / f"g{i}.pt")
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "debug", "unfold_dbg.py"), f], cwd=root, env=e,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(torch.load(f))
    for a, b, c in zip(*outs):
        assert torch.equal(a, b)                                   # fused, twice
        assert float((a - c).abs().max()) <= 2e-6 * float(c.abs().max())   # fused against the two kernels it replaces


def test_wide_operand_gather_equals_cat_and_index_select(dev):
    """Round 6: the weights of a dense block's wide convolutions (rows of one channel group out of all later growth layers,
    side by side, re-ordered from the reference's per-list-element CReLU order -- utils/nn.py:198-200 -- to the single-tensor
    order) come from otgan_gather3d_batched_f32 instead of torch.cat / index_select: the same elements, bit for bit."""
    from otgan_amd import ops
    gen = torch.Generator().manual_seed(5)
    F, L, segs0 = 16, 6, (24, 8)
    C0 = sum(segs0)
    per_layer = []
    for k in range(L):
        ceff = 2 * (C0 + k * F)
        w = torch.randn(9 * ceff, F, generator=gen).to(dev)
        per_layer.append((w, w.view(9, ceff, F).permute(2, 0, 1).contiguous().view(F, 9 * ceff)))
    order = ops._input_row_order(segs0, ops.ACT["crelu"], dev)
    for layers, row0, nrows, od in ((range(0, L), 0, 2 * C0, order), (range(3, L), 2 * C0, 2 * 3 * F, None),
                                    (range(2, L), 2 * C0 + 8, 40, None)):
        layers = list(layers)
        w = torch.cat([per_layer[k][0].view(9, -1, F)[:, row0:row0 + nrows, :] for k in layers], dim=2)
        wT = torch.cat([per_layer[k][1].view(F, 9, -1)[:, :, row0:row0 + nrows] for k in layers], dim=0)
        if od is not None:
            w, wT = w.index_select(1, od), wT.index_select(2, od)
        n = len(layers) * F
        w, wT = w.contiguous().view(9 * nrows, n), wT.contiguous().view(n, 9 * nrows)
        real_prepare = ops.prepare_filters
        ops.prepare_filters = lambda desc, which, t: None        # (only the gathered weights are under test)
        try:
            got = ops._wide_operands(per_layer, layers, row0, nrows, od, F, None)
        finally:
            ops.prepare_filters = real_prepare
        assert torch.equal(got["w"], w) and torch.equal(got["wT"], wT)
