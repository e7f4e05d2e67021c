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


@pytest.mark.parametrize("N,H,nslices", [(256, 8, 8), (64, 8, 17), (256, 16, 8), (37, 16, 3), (8, 32, 4)],
                         ids=["8x8_block3", "8x8_longest", "16x16_block2", "16x16_odd_batch", "32x32_per_layer"])
def test_whole_chain_in_one_launch(dev, N, H, nslices):
    """Round 6: otgan_dense16_chain_fwd_f32 walks a group's chain -- layer j adds conv3x3(crelu(slices
    0 .. j-1)) onto slice j (reference models/densenet.py:11-16 through utils/nn.py:198-200,243-262) -- in ONE launch where a
    workgroup covers an image (8 x 8, 16 x 16; 32 x 32 keeps one launch per layer).  Every workgroup bounds the slices produced in
    the launch by what IT wrote (the global records are still filling), so the test is against fp64, layer by layer, at the
    layer tolerance of the suite; twice, bit for bit; and the global records of the produced slices must hold their amax."""
    from otgan_amd import _lib, ops
    L = _lib.lib()
    g = torch.Generator().manual_seed(N + H + nslices)
    F, C0 = 16, 32
    Ctot = C0 + nslices * F + 16
    buf0 = torch.randn(N, H, H, Ctot, generator=g)
    buf0[..., C0 + F:C0 + 2 * F] *= 11.0                                   # slices of different magnitudes
    buf0[3 % N] *= 7.0                                                    # ... and images
    wTs = [(torch.randn(F, 9 * 2 * F * j, generator=g) * (0.05 / j ** 0.5)) for j in range(1, nslices)]
    # fp64 reference: the chain layer by layer on the growing buffer
    ref = buf0.double().clone()
    grp = ref[..., C0:C0 + nslices * F]
    for j in range(1, nslices):
        grp[..., 16 * j:16 * j + 16] = _reference(grp[..., :16 * j].contiguous(), wTs[j - 1].double(), grp[..., 16 * j:16 * j + 16])
    nsl = list(range(1, nslices))
    fbytes = [int(L.otgan_dense16_filter_bytes(n)) for n in nsl]
    flat = torch.empty(sum(fbytes), dtype=torch.uint8, device=dev)
    wdev = [w.to(dev) for w in wTs]
    offs = [sum(fbytes[:i]) for i in range(len(nsl))]
    n = len(nsl)
    pw = (ctypes.c_void_p * n)(*[w.data_ptr() for w in wdev])
    pn = (ctypes.c_int * n)(*nsl)
    pf = (ctypes.c_void_p * n)(*[flat.data_ptr() + o for o in offs])
    _lib.check(L.otgan_dense16_prepare_filters_f32(ctypes.cast(pw, ctypes.c_void_p), ctypes.cast(pn, ctypes.c_void_p),
                                                   ctypes.cast(pf, ctypes.c_void_p), n, _lib.stream_ptr()), "prepare")
    outs = []
    for rep in range(2):
        buf = buf0.to(dev)
        R = torch.zeros((1 + nslices, ops.AMAX_RECORD_FLOATS), device=dev)
        R[0, 0] = buf[..., C0:C0 + nslices * F].abs().max()               # the wide convolutions' record bounds every initial sum
        R[1, 32] = buf[..., C0:C0 + F].abs().max()                        # slice 0 (final before the chain)
        _lib.check(L.otgan_dense16_chain_fwd_f32(N, H, H, nslices, buf.data_ptr() + 4 * C0, Ctot, ctypes.cast(pf, ctypes.c_void_p),
                                                 R.data_ptr(), _lib.stream_ptr()), "chain_fwd")
        torch.cuda.synchronize()
        outs.append((buf.cpu(), R.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    got, R = outs[0]
    assert torch.equal(got[..., :C0 + F], buf0[..., :C0 + F]) and torch.equal(got[..., C0 + nslices * F:], buf0[..., C0 + nslices * F:])
    for j in range(1, nslices):
        a, b = got[..., C0 + 16 * j:C0 + 16 * j + 16].double(), ref[..., C0 + 16 * j:C0 + 16 * j + 16]
        assert float((a - b).norm() / b.norm()) < 2e-5, j
        assert float(R[1 + j].max()) == float(got[..., C0 + 16 * j:C0 + 16 * j + 16].abs().max()), j


_BLOCK_WORKER = r"""
import sys, torch
sys.path.insert(0, sys.argv[2])
from otgan_amd import ops, _lib
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(17)
out = {}
_lib.lib()
_lib.prof_reset(); _lib.prof_enable(True)
for name, N, H, C0 in (("8x8", 64, 8, 200), ("16x16", 32, 16, 144)):
    L, F = 16, 16
    x0 = torch.randn(N, H, H, C0, generator=gen)
    x0[0] *= 64.0            # one image six binades above the rest: the per-batch scale of a slice is not the other images' per-image one
    x0 = x0.to(dev).requires_grad_(True)
    params = []
    for k in range(L):
        params.append([(torch.randn(3, 3, 2 * (C0 + k * F), F, generator=gen) * 0.05).to(dev).requires_grad_(True),
                       (torch.rand(F, generator=gen) + 0.5).to(dev).requires_grad_(True),
                       (torch.randn(F, generator=gen) * 0.1).to(dev).requires_grad_(True)])
    dy = torch.randn(N, H, H, C0 + L * F, generator=gen).to(dev)
    y = ops.dense_block_op(x0, (C0,), params, 3, ops.ACT["crelu"])
    grads = torch.autograd.grad(y, [x0] + [t for p in params for t in p], dy)
    ops.join_side_stream(grads)
    torch.cuda.synchronize()
    out[name + ".y"] = y.detach().cpu()
    for i, t in enumerate(grads):
        out[f"{name}.g{i}"] = t.detach().cpu()
pc = _lib.prof_collect(); _lib.prof_enable(False)
out["launches"] = torch.tensor([pc["conv_fwd"]["launches"], pc["conv_dgrad"]["launches"]])
torch.save(out, sys.argv[1])
"""


def test_one_launch_chains_agree_with_the_per_layer_kernels(dev, tmp_path):
    """The 16-layer dense blocks of the DenseNet critic at 8 x 8 and 16 x 16 (models/densenet.py:11-16), forward and backward,
    with the chains as ONE launch per group (default) and as one launch per layer / slice (OTGAN_DENSE16_CHAIN=0, read once per
    process).  The two differ only in the power-of-two scale of the fp16 operand pieces (per image against per batch), which
    commutes with the rounding of the pieces except at the edges of the fp16 exponent range: every output and gradient tensor
    within 1e-5 of each other (bit-identical in practice), each run deterministic, and 13 launches per group fewer."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "worker.py"
    script.write_text(_BLOCK_WORKER)
    res = {}
    for tag, env in (("one", {}), ("one_again", {}), ("per_layer", {"OTGAN_DENSE16_CHAIN": "0"})):
        path = tmp_path / (tag + ".pt")
        subprocess.run([sys.executable, str(script), str(path), root], check=True, env=dict(os.environ, **env), timeout=600)
        res[tag] = torch.load(path)
    for k in res["one"]:
        if k == "launches":
            continue
        a, b, c = res["one"][k], res["one_again"][k], res["per_layer"][k]
        assert torch.equal(a, b), k
        assert bool(torch.isfinite(a).all())
        err = float((a.double() - c.double()).norm() / c.double().norm().clamp_min(1e-30))
        assert err < 1e-5, (k, err)
    # two blocks x two groups of eight slices: 7 forward layers per group become one launch (8 x 8 and 16 x 16), 7 backward
    # slices per group too at 8 x 8 (at 16 x 16 the per-slice kernels on half images are faster and stay)
    one, per = res["one"]["launches"].tolist(), res["per_layer"]["launches"].tolist()
    assert per[0] - one[0] == 2 * 2 * 6 and per[1] - one[1] == 2 * 6, (one, per)
