"""GPU: amax records written by the kernels that PRODUCE a tensor (otgan_layers.h: y_amax_out / dx_amax_out,
otgan_glu_*_amax_f32, otgan_feature_head_bwd_amax_f32).  A Winograd layer scales its fp16 GEMM operands by the largest
magnitude of the tensor they are a transform of; the record must equal max |t| EXACTLY (max is order-free), so a layer
fed by a producer's record computes bit-identical results to one that reduces the tensor itself."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    from otgan_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


def _rec_value(t):
    from otgan_amd import ops
    rec = ops.amax_of(t)
    assert rec is not None, "the producer left no amax record"
    sub = rec.view(16, 32)[:, 0]                 # 16 sub-slots, one cache line each: the value is their maximum
    return float("nan") if bool(torch.isnan(sub).any()) else sub.max().item()


def test_glu_records_are_exact_and_tagged():
    from otgan_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(64, 8, 8, 512, generator=g) * 3).to(dev).requires_grad_(True)
    y = ops.glu(x)
    assert _rec_value(y) == y.detach().abs().max().item()
    dy = torch.randn(y.shape, generator=g).to(dev)
    (dx,) = torch.autograd.grad(y, x, dy)
    assert _rec_value(dx) == dx.abs().max().item()
    # reference values of the float4 kernels (the scalar kernels they replace computed the same expressions)
    a, l = x.detach()[..., :256].double(), x.detach()[..., 256:].double()
    s = torch.sigmoid(l)
    assert float((y.detach().double() - a * s).norm() / (a * s).norm()) < 1e-6
    ref_dx = torch.cat([dy.double() * s, dy.double() * a * s * (1 - s)], -1)
    assert float((dx.double() - ref_dx).norm() / ref_dx.norm()) < 1e-6     # (1 - s) cancels in fp32 for saturated gates
    # a view loses the tag, carry_amax keeps it; an in-place change invalidates it
    v = y.detach().view(64, -1)
    assert ops.amax_of(v) is None
    assert ops.amax_of(ops.carry_amax(v, y)) is not None
    y2 = ops.glu(x.detach())
    y2.mul_(2.0)
    assert ops.amax_of(y2) is None


def test_nan_and_inf_reach_the_record():
    from otgan_amd import ops
    dev = _dev()
    x = torch.randn(8, 4, 4, 64, device=dev)
    x[3, 1, 2, 5] = float("nan")
    assert np.isnan(_rec_value(ops.glu(x)))
    x[3, 1, 2, 5] = float("inf")
    x[3, 1, 2, 5 + 32] = 50.0           # sigmoid -> 1: inf * 1
    assert np.isinf(_rec_value(ops.glu(x)))
    z = torch.zeros(8, 4, 4, 64, device=dev)
    assert _rec_value(ops.glu(z)) == 0.0


def _critic_pair(dev, seed, strip):
    """RGB-in 5x5 -> CReLU 5x5 stride 2 (Winograd) -> CReLU 5x5 stride 2: outputs and gradients; `strip` removes the
    producers' records so that every layer reduces its tensors itself (the round-2 behaviour)."""
    from otgan_amd import ops
    g = torch.Generator().manual_seed(seed)
    x = (torch.rand(32, 32, 32, 3, generator=g) * 2 - 1).to(dev).requires_grad_(True)
    Vs = [(torch.randn(5, 5, 3, 128, generator=g) * 0.05), (torch.randn(5, 5, 256, 128, generator=g) * 0.05),
          (torch.randn(5, 5, 256, 256, generator=g) * 0.05)]
    Vs = [v.to(dev).requires_grad_(True) for v in Vs]
    gs = [torch.ones(v.shape[-1], device=dev, requires_grad=True) for v in Vs]
    bs = [torch.zeros(v.shape[-1], device=dev, requires_grad=True) for v in Vs]
    seen = []

    def maybe_strip(t):
        seen.append(ops.amax_of(t) is not None)
        if strip and hasattr(t, "_otgan_amax"):
            del t._otgan_amax
        return t

    y0 = maybe_strip(ops.conv2d_op(x, Vs[0], gs[0], bs[0], stride=1, preact=ops.ACT[None]))
    y1 = maybe_strip(ops.conv2d_op(y0, Vs[1], gs[1], bs[1], stride=2, preact=ops.ACT["crelu"]))
    y2 = ops.conv2d_op(y1, Vs[2], gs[2], bs[2], stride=2, preact=ops.ACT["crelu"])
    f = ops.feature_head(y2)
    df = torch.randn(f.shape, generator=g).to(dev)
    grads = torch.autograd.grad(f, [x] + Vs, df)
    return [y0, y1, y2] + list(grads), seen


def test_conv_chain_with_producer_records_is_bit_identical():
    from otgan_amd import ops
    dev = _dev()
    fused, seen = _critic_pair(dev, 11, strip=False)
    assert seen == [True, True]                      # RGB-in layer and the strided layer's output transform
    assert _rec_value(fused[0]) == fused[0].detach().abs().max().item()
    assert _rec_value(fused[1]) == fused[1].detach().abs().max().item()
    plain, _ = _critic_pair(dev, 11, strip=True)
    for i, (a, b) in enumerate(zip(fused, plain)):
        assert torch.equal(a.detach(), b.detach()), i


def test_backward_records(monkeypatch):
    """dy of every Winograd layer in the chain arrives with its producer's record: the feature head's backward and
    the input-gradient output transform of the strided layers; no absmax launch is left in the backward pass."""
    from otgan_amd import ops
    dev = _dev()
    calls = []
    real = ops.absmax_record
    monkeypatch.setattr(ops, "absmax_record", lambda t: (calls.append(tuple(t.shape)), real(t))[1])
    _critic_pair(dev, 5, strip=False)
    assert calls == [], calls
    _critic_pair(dev, 5, strip=True)
    assert len(calls) >= 2          # stripped forward records are reduced again (backward ones still arrive tagged)


def test_glu_backward_leaves_the_bias_gradient():
    """GluFunction.backward tags its result with its column sums (otgan_glu_bwd_colsum_f32): the bias gradient of the
    convolution in front of the GLU, which then skips its own reduction (ops.colsum_of); a ragged last 64-column block and
    more rows than one chunk."""
    from otgan_amd import ops
    dev = _dev()
    torch.manual_seed(3)
    for shape in ((4, 8, 8, 2 * 72), (256, 16, 16, 2 * 256), (2, 4, 4, 2 * 8)):
        x = torch.randn(shape, device=dev, requires_grad=True)
        y = ops.glu(x)
        dy = torch.randn_like(y)
        seen = {}

        class Probe(torch.autograd.Function):          # sits where the convolution's backward would: sees the GLU's dx
            @staticmethod
            def forward(ctx, t):
                return t.view_as(t)

            @staticmethod
            def backward(ctx, g):
                seen["colsum"] = ops.colsum_of(g)
                seen["g"] = g
                return g
        x2 = torch.randn(shape, device=dev, requires_grad=True)
        y2 = ops.glu(Probe.apply(x2))
        y2.backward(dy)
        cs, g = seen["colsum"], seen["g"]
        assert cs is not None and cs.shape == (shape[-1],)
        ref = g.double().reshape(-1, shape[-1]).sum(0)
        assert float((cs.double() - ref).abs().max()) <= 2e-6 * float(g.double().abs().reshape(-1, shape[-1]).sum(0).max()) + 1e-12
        # and the values of dx itself against autograd through the formula
        xr = x2.detach().double().requires_grad_(True)
        a, l = xr[..., :shape[-1] // 2], xr[..., shape[-1] // 2:]
        (a * torch.sigmoid(l)).backward(dy.double())
        assert float((g.double() - xr.grad).abs().max()) < 1e-5
        g.add_(1.0)                                     # modified since: the tag must not be trusted any more
        assert ops.colsum_of(g) is None


@pytest.mark.parametrize("case", ["igemm_s2_list", "igemm_1x1", "rgb_out", "up3"])
def test_input_gradient_records_of_the_remaining_producers(case):
    """Round 4: the input-gradient epilogues of the implicit-GEMM layers (stride-2 3x3 over a CReLU list input = a DenseNet
    transition; a dense 1x1), of the RGB-out layer and of the 3x3 upsampling layers (output transform, also the forward
    record of y) leave exact records of what they write -- the dense block in front of such a layer no longer reduces
    its whole gradient buffer."""
    from otgan_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(sum(map(ord, case)))
    N = 4
    if case == "igemm_s2_list":
        x = torch.randn(N, 16, 16, 80, generator=g).to(dev).requires_grad_(True)
        V = (torch.randn(3, 3, 160, 40, generator=g) * 0.05).to(dev)
        kw = dict(stride=2, preact=ops.ACT["crelu"], segs=[48, 16, 16])
    elif case == "igemm_1x1":
        x = torch.randn(N, 8, 8, 64, generator=g).to(dev).requires_grad_(True)
        V = (torch.randn(1, 1, 64, 96, generator=g) * 0.05).to(dev)
        kw = dict(stride=1, preact=ops.ACT[None])
    elif case == "rgb_out":
        x = torch.randn(N, 32, 32, 64, generator=g).to(dev).requires_grad_(True)
        V = (torch.randn(3, 3, 128, 3, generator=g) * 0.05).to(dev)
        kw = dict(stride=1, preact=ops.ACT["crelu"], segs=[32, 32])
    else:
        x = torch.randn(N, 8, 8, 64, generator=g).to(dev).requires_grad_(True)
        V = (torch.randn(3, 3, 128, 32, generator=g) * 0.05).to(dev)
        kw = dict(stride=1, upsample=True, preact=ops.ACT["crelu"])
    Co = V.shape[-1]
    gg, b = torch.ones(Co, device=dev), torch.zeros(Co, device=dev)
    y = ops.conv2d_op(x, V, gg, b, **kw)
    if case == "up3":
        assert _rec_value(y) == y.detach().abs().max().item()
    dy = torch.randn(y.shape, generator=g).to(dev)
    (dx,) = torch.autograd.grad(y, x, dy)
    assert _rec_value(dx) == dx.abs().max().item()
