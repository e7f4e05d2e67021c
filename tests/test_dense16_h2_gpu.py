"""GPU: the growth-layer chain kernel on two scaled fp16 pieces (dense16_fwd_h2_kernel, round 4) through the C ABI
(otgan_dense16_prepare_filters_f32 + otgan_conv2d_fwd_pf_f32 with list_width / x_amax_count) against an fp64 convolution
of the CReLU-interleaved slices [x_0, -x_0, x_1, -x_1, ...] (reference utils/nn.py:198-200, models/densenet.py:11-16).
Tolerance: the layer-kernel bound of the suite, 2e-5 relative L2 (measured ~1e-6: 22-bit operands, fp32 accumulation)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from otgan_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


def _reference(xs, wT, y0):
    """xs [N,H,W,16 n] fp64 (the slices), wT [16, 9 * 32 n]; returns y0 + conv3x3_same(crelu-interleaved xs)."""
    n = xs.shape[-1] // 16
    parts = []
    for s in range(n):
        sl = xs[..., 16 * s:16 * s + 16]
        parts += [sl.clamp(min=0), (-sl).clamp(min=0)]
    eff = torch.cat(parts, -1).permute(0, 3, 1, 2)                      # [N, 32 n, H, W]
    w = wT.reshape(16, 9, 32 * n).permute(0, 2, 1).reshape(16, 32 * n, 3, 3)
    return y0 + torch.nn.functional.conv2d(eff, w, padding=1).permute(0, 2, 3, 1)


@pytest.mark.parametrize("N,H,n_own", [(8, 32, 1), (8, 32, 3), (128, 32, 7), (32, 16, 4), (256, 16, 2), (64, 8, 5), (512, 8, 3)])
def test_chain_kernel_vs_fp64(dev, N, H, n_own):
    from otgan_amd import _lib, ops
    from otgan_amd._lib_layers import ConvDesc
    L = _lib.lib()
    g = torch.Generator().manual_seed(N + H + n_own)
    C0, F = 32, 16
    Ctot = C0 + 16 * F
    buf = torch.randn(N, H, H, Ctot, generator=g).to(dev)
    buf[..., C0 + 2 * F:C0 + 3 * F] *= 37.0                                # slices of different magnitudes
    g0, k = 1, 1 + n_own                                                  # chain over growth slices [g0, k), output slice k
    wT = (torch.randn(F, 9 * 2 * F * n_own, generator=g) * 0.05).to(dev)
    desc = ConvDesc(N, H, H, n_own * F, Ctot, 0, 3, 3, 1, F, Ctot, C0 + k * F, ops.ACT["crelu"], 1)
    desc.y_accumulate, desc.list_width = 1, F
    assert L.otgan_dense16_h2_ok(ctypes.byref(desc)) == 1
    # records: the wide convolutions' record (here: nothing) then one per slice, as DenseBlockFunction lays them out
    R = torch.zeros((2 + n_own, ops.AMAX_RECORD_FLOATS), device=dev)
    for j in range(n_own):
        sl = buf[..., C0 + (g0 + j) * F:C0 + (g0 + j + 1) * F]
        R[1 + j, 32 * (j % 16)] = sl.abs().max()                          # any sub-slot
    fq = torch.empty(int(L.otgan_dense16_filter_bytes(n_own)), dtype=torch.uint8, device=dev)
    pw, pn, pf = (ctypes.c_void_p * 1)(wT.data_ptr()), (ctypes.c_int * 1)(n_own), (ctypes.c_void_p * 1)(fq.data_ptr())
    _lib.check(L.otgan_dense16_prepare_filters_f32(ctypes.cast(pw, ctypes.c_void_p), ctypes.cast(pn, ctypes.c_void_p),
                                                   ctypes.cast(pf, ctypes.c_void_p), 1, _lib.stream_ptr()), "prepare")
    desc.x_amax, desc.x_amax_count = R[0].data_ptr(), 1 + n_own
    desc.y_amax_out = R[1 + n_own].data_ptr()
    cmap, _inv = ops.channel_maps((F,) * n_own, ops.ACT["crelu"], dev)
    y0 = buf[..., C0 + k * F:C0 + (k + 1) * F].double().cpu()
    xs = buf[..., C0 + g0 * F:C0 + k * F].double().cpu()
    ops.conv_fwd_raw(desc, buf[..., C0 + g0 * F:], cmap, wT, None, buf, fq)
    got = buf[..., C0 + k * F:C0 + (k + 1) * F].double().cpu()
    want = _reference(xs, wT.double().cpu(), y0)
    err = float((got - want).norm() / want.norm())
    assert err < 2e-5, err
    # the record of the sums written
    assert float(R[1 + n_own].max()) == float(got.abs().max().float())
    # the same call without prepared filters runs the three-bf16-piece kernel: same result within the same bound
    buf2 = buf.clone()
    buf2[..., C0 + k * F:C0 + (k + 1) * F] = y0.float().to(dev)
    ops.conv_fwd_raw(desc, buf2[..., C0 + g0 * F:], cmap, wT, None, buf2, None)
    other = buf2[..., C0 + k * F:C0 + (k + 1) * F].double().cpu()
    assert float((other - want).norm() / want.norm()) < 2e-5
    # a NaN record (a NaN anywhere in the slices it bounds) must not vanish in the fp16 pieces
    R[1, 0] = float("nan")
    ops.conv_fwd_raw(desc, buf[..., C0 + g0 * F:], cmap, wT, None, buf, fq)
    assert bool(torch.isnan(buf[..., C0 + k * F:C0 + (k + 1) * F]).all())
