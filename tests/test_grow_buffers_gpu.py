"""GPU: dense blocks that grow in the buffer of the layer in front of them (round 5; reference models/densenet.py:11-21,
60-73: `x.append(conv2d(x, F))` over a list the reference re-concatenates).  A convolution called with `grow=` allocates
its output as the channel prefix of a buffer with room for the block's outputs, the block writes behind it in place, and
on the way back the block hands the gradient of its input on as a VIEW of its gradient buffer, which the convolution's
backward kernels read through the ABI's channel stride (ldy).  None of that may change a value: every case compares with
the copying path BIT FOR BIT -- outputs, input gradients, every parameter gradient -- for the three kinds of layer that
feed a block in the DenseNet models (the RGB-in first convolution, a stride-2 CReLU transition over a list, an upsampling
transition followed by a [x, noise] list)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from otgan_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


def _params(gen, shapes, dev):
    out = []
    for shp in shapes:
        t = (torch.randn(shp, generator=gen, device=dev) * 0.05).requires_grad_(True)
        out.append(t)
    return out


def _block_params(gen, C0, L, F, dev):
    ps = []
    for k in range(L):
        V = (torch.randn((3, 3, 2 * (C0 + k * F), F), generator=gen, device=dev) * 0.05).requires_grad_(True)
        g = (1.0 + 0.1 * torch.randn(F, generator=gen, device=dev)).requires_grad_(True)
        b = (0.1 * torch.randn(F, generator=gen, device=dev)).requires_grad_(True)
        ps.append((V, g, b))
    return ps


def _run(dev, kind, grow_on):
    """conv (kind) -> [optional noise list] -> dense block (CReLU, L = 4, F = 16) -> weighted sum; returns every value."""
    from otgan_amd import ops
    gen = torch.Generator(device=dev).manual_seed(7)
    N, L, F = 4, 4, 16
    if kind == "rgb":
        x = torch.randn((N, 16, 16, 3), generator=gen, device=dev)
        V, g, b = _params(gen, [(3, 3, 3, 32), (32,), (32,)], dev)
        kw = dict(stride=1, upsample=False, preact=0, segs=None)
        Cout, others = 32, []
    elif kind == "stride2":
        x = torch.randn((N, 16, 16, 48), generator=gen, device=dev).requires_grad_(True)
        V, g, b = _params(gen, [(3, 3, 96, 24), (24,), (24,)], dev)
        kw = dict(stride=2, upsample=False, preact=ops.ACT["crelu"], segs=[32, 16])
        Cout, others = 24, []
    else:
        x = torch.randn((N, 8, 8, 48), generator=gen, device=dev).requires_grad_(True)
        V, g, b = _params(gen, [(3, 3, 96, 24), (24,), (24,)], dev)
        kw = dict(stride=1, upsample=True, preact=ops.ACT["crelu"], segs=None)
        Cout = 24
        others = [torch.rand((N, 16, 16, F), generator=gen, device=dev) * 2 - 1]
    C0 = Cout + sum(int(t.shape[-1]) for t in others)
    bp = _block_params(gen, C0, L, F, dev)
    grow = (C0 - Cout) + L * F if grow_on else 0
    y = ops.conv2d_op(x, V, g.abs() + 0.5, b, grow=grow, **kw)
    if others:
        y_rec = ops.amax_of(y)              # (read before the in-place extension bumps the shared version counter)
        x0 = ops.extend_channels([y] + others, C0 + L * F) if grow_on else None
        assert (x0 is not None) == grow_on
        if grow_on:
            # the producing convolution's amax record travels with the extended block input (the block would
            # otherwise reduce its input once more per step)
            assert y_rec is not None and ops.amax_of(x0) is not None
        if x0 is None:
            x0 = ops.concat_channels([y] + others)
        segs0 = [Cout] + [int(t.shape[-1]) for t in others]
    else:
        x0, segs0 = y, [Cout]
    if grow_on:
        assert ops.grown_buffer(x0, C0 + L * F) is not None          # the block finds the room ...
    buf = ops.dense_block_op(x0, segs0, bp, ksize=3, preact=ops.ACT["crelu"])
    if grow_on:
        assert buf.data_ptr() == y.data_ptr() and ops.grown_buffer(x0, C0 + L * F) is None     # ... takes it, once
    wsum = torch.randn(buf.shape, generator=gen, device=dev)
    leaves = [V, g, b] + [t for p in bp for t in p] + ([x] if x.requires_grad else [])
    grads = torch.autograd.grad((buf * wsum).sum(), leaves)
    torch.cuda.synchronize()
    return [buf.detach().clone()] + [t.detach().clone() for t in grads]


@pytest.mark.parametrize("kind", ["rgb", "stride2", "upsample_list"])
def test_grown_block_equals_copying_block(dev, kind):
    a = _run(dev, kind, False)
    b = _run(dev, kind, True)
    assert len(a) == len(b)
    for i, (u, v) in enumerate(zip(a, b)):
        assert u.shape == v.shape and torch.equal(u, v), (kind, i, float((u - v).abs().max()))


def test_strided_upstream_gradient_of_a_convolution(dev):
    """Conv2dFunction.backward on a channel-prefix view (read in place) against the same values made contiguous"""
    from otgan_amd import ops
    gen = torch.Generator(device=dev).manual_seed(3)
    x = torch.randn((4, 16, 16, 64), generator=gen, device=dev).requires_grad_(True)
    V, g, b = _params(gen, [(3, 3, 64, 32), (32,), (32,)], dev)
    big = torch.randn((4, 16, 16, 96), generator=gen, device=dev)
    res = []
    for dy in (big[..., :32], big[..., :32].contiguous()):
        y = ops.conv2d_op(x, V, g.abs() + 0.5, b)
        res.append(torch.autograd.grad(y, [x, V, g, b], dy))
    for u, v in zip(*res):
        assert torch.equal(u, v)
    assert ops.channel_prefix_stride(big[..., :32]) == 96 and ops.channel_prefix_stride(big[..., 2:34]) is None   # (unaligned)
