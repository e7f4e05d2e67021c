"""torch.autograd plumbing around the HIP layer kernels (include/otgan_layers.h).

Every Function launches hand-written gfx950 kernels through the C ABI on the current
PyTorch stream; PyTorch only owns the device memory and the backward graph.  There is no
fallback: CPU tensors raise OtganError.
"""
import ctypes
import os
import sys
import weakref
from collections import OrderedDict

import torch

from . import _lib
from ._lib_layers import ConvDesc, WnBwdLayer, WnFwdLayer

ACT = {None: 0, "none": 0, "crelu": 1, "celu": 2, "elu": 3, "relu": 4}
DOUBLED = (1, 2)

_ws = {}


class _NullCtx:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NULL_CTX = _NullCtx()


def workspace(nbytes, device):
    """Grow-only scratch buffer per (device, stream) -- the caller-provided workspace of the C ABI.
    Kernels of one stream run in order, so one buffer per stream is race-free; two trainers driving
    different streams get different buffers."""
    key = (device.index if device.index is not None else torch.cuda.current_device(),
           torch.cuda.current_stream(device).cuda_stream)
    buf = _ws.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(max(nbytes, 1 << 20)), dtype=torch.uint8, device=device)
        _ws[key] = buf
    return buf


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.OtganError("otgan_amd ops need CUDA (MI355X) tensors; there is no CPU fallback")


_map_cache = {}


def channel_maps(segs, preact, device):
    """Device tables for a *list* input (reference nn.py:198,200 interleaves per element:
    [x0,-x0,x1,-x1,...]).  Returns (cmap, inv) int32 tensors, or (None, None) when the default
    single-tensor ordering applies."""
    segs = tuple(int(s) for s in segs)
    if preact not in DOUBLED or len(segs) <= 1:
        return None, None
    key = (segs, device)
    hit = _map_cache.get(key)
    if hit is not None:
        return hit
    C = sum(segs)
    cmap, invp, invn = [], [0] * C, [0] * C
    off = 0
    for s in segs:
        base = len(cmap)
        cmap += [off + i for i in range(s)]
        cmap += [(off + i) | (1 << 31) for i in range(s)]
        for i in range(s):
            invp[off + i] = base + i
            invn[off + i] = base + s + i
        off += s
    # int32 with the sign bit as the "negate" flag
    cm = torch.tensor([c - (1 << 32) if c >= (1 << 31) else c for c in cmap], dtype=torch.int32,
                      device=device)
    inv = torch.tensor(invp + invn, dtype=torch.int32, device=device)
    _map_cache[key] = (cm, inv)
    return cm, inv


def make_desc(x, C, upsample, kh, kw, stride, cout, ldy, y_coff, preact, segs=None):
    N, H, W, ldx = x.shape
    quads = 1 if all(int(s) % 4 == 0 for s in (segs or (C,))) else 0
    return ConvDesc(N, H, W, C, ldx, 1 if upsample else 0, kh, kw, stride, cout, ldy, y_coff, preact, quads)


def out_hw(H, W, upsample, stride):
    Hin, Win = (H * 2, W * 2) if upsample else (H, W)
    return -(-Hin // stride), -(-Win // stride)


# ------------------------------------------------------------------------------- weight cache
# Normalised (and, for upsampling layers, folded) weights depend only on (V, g); the critic
# runs two or three forward passes per step on the same parameters.  Entries are keyed by the
# parameter object and validated by (global epoch, tensor versions): the optimiser / EMA kernels
# write parameters through raw pointers, so they bump `weights_epoch` explicitly.
weights_epoch = 0          # global: bumped when anything may have changed (checkpoint load, tests)
_storage_epoch = {}        # per storage: bumped by the optimiser / EMA kernels that write into it
_wcache = OrderedDict()    # id(V) -> (V, g, token, value), least recently used first
_WCACHE_MAX = 1024         # > 2 networks x (live + EMA) x ~160 layers (DenseNet); entries are evicted one by one


def bump_weights_epoch(t=None):
    """Invalidate cached normalised weights: of the storage `t` lives in, or (no argument) all."""
    global weights_epoch
    if t is None:
        weights_epoch += 1
    else:
        key = t.untyped_storage().data_ptr()
        _storage_epoch[key] = _storage_epoch.get(key, 0) + 1


def _epoch_of(t):
    return _storage_epoch.get(t.untyped_storage().data_ptr(), 0)


def cached_weights(V, g, compute):
    key = id(V)
    token = (weights_epoch, _epoch_of(V), _epoch_of(g), V._version, g._version)
    hit = _wcache.get(key)
    if hit is not None and hit[0] is V and hit[1] is g and hit[2] == token:
        _wcache.move_to_end(key)
        return hit[3]
    val = compute()
    _wcache[key] = (V, g, token, val)      # holding V keeps id(V) from being reused
    _wcache.move_to_end(key)
    while len(_wcache) > _WCACHE_MAX:      # evict the least recently used entry only
        _wcache.popitem(last=False)
    return val


# ------------------------------------------------------------------------------- raw launchers
def weightnorm_fwd(V2d, g):
    """V2d: [K, Cout] view of the HWIO direction tensor.  Returns (w, wT, inv_norm); w and wT carry the amax record of
    the normalised weights (the Winograd filter operands are scaled by it: prepare_filters)."""
    K, Cout = V2d.shape
    w = torch.empty_like(V2d)
    wT = torch.empty((Cout, K), dtype=V2d.dtype, device=V2d.device)
    inv = torch.empty(Cout, dtype=V2d.dtype, device=V2d.device)
    rec = amax_slot(V2d.device) if _FUSED_AMAX else None
    _lib.check(_lib.lib().otgan_weightnorm_fwd_amax_f32(V2d.data_ptr(), g.data_ptr(), K, Cout,
                                                        w.data_ptr(), wT.data_ptr(), inv.data_ptr(), _lib.ptr(rec),
                                                        _lib.stream_ptr()), "weightnorm_fwd")
    if rec is not None:
        tag_amax(w, rec)
        tag_amax(wT, rec)
    return w, wT, inv


def weightnorm_bwd(V2d, g, inv, dw):
    K, Cout = V2d.shape
    dV = torch.empty_like(V2d)
    dg = torch.empty_like(g)
    scratch = torch.empty(Cout, dtype=V2d.dtype, device=V2d.device)
    _lib.check(_lib.lib().otgan_weightnorm_bwd_f32(V2d.data_ptr(), g.data_ptr(), inv.data_ptr(),
                                                   dw.data_ptr(), K, Cout, dV.data_ptr(),
                                                   dg.data_ptr(), scratch.data_ptr(),
                                                   _lib.stream_ptr()), "weightnorm_bwd")
    return dV, dg


def colsum(a2d_ptr, rows, cols, lda, device):
    out = torch.empty(cols, dtype=torch.float32, device=device)
    scratch = torch.empty(256 * cols, dtype=torch.float32, device=device)
    _lib.check(_lib.lib().otgan_colsum_f32(a2d_ptr, rows, cols, lda, out.data_ptr(),
                                           scratch.data_ptr(), _lib.stream_ptr()), "colsum")
    return out


def conv_fwd_raw(desc, x, cmap, wT, bias, y, filters=None):
    L = _lib.lib()
    need = L.otgan_conv2d_workspace_bytes(ctypes.byref(desc), 0)
    ws = workspace(need, x.device)
    _lib.check(L.otgan_conv2d_fwd_pf_f32(ctypes.byref(desc), x.data_ptr(), _lib.ptr(cmap),
                                         wT.data_ptr(), _lib.ptr(filters), _lib.ptr(bias), y.data_ptr(),
                                         ws.data_ptr(), ws.numel(), _lib.stream_ptr()), "conv2d_fwd")


def conv_dgrad_raw(desc, dy, w, x, inv, dx, lddx, accumulate, filters=None):
    L = _lib.lib()
    need = L.otgan_conv2d_workspace_bytes(ctypes.byref(desc), 1)
    ws = workspace(need, dy.device)
    _lib.check(L.otgan_conv2d_dgrad_pf_f32(ctypes.byref(desc), dy.data_ptr(), w.data_ptr(), _lib.ptr(filters),
                                           _lib.ptr(x), _lib.ptr(inv), dx.data_ptr(), lddx,
                                           1 if accumulate else 0, ws.data_ptr(), ws.numel(),
                                           _lib.stream_ptr()), "conv2d_dgrad")


def absmax_record(t):
    """amax record (otgan_layers.h: otgan_absmax_f32) of a contiguous NHWC tensor: AMAX_RECORD_FLOATS floats on its device, value =
    max |t|.  The Winograd passes scale their two-piece fp16 operands by it; computing it here, once per tensor,
    lets forward + wgrad (x) and dgrad + wgrad (dy) share one reduction."""
    rec = torch.empty(AMAX_RECORD_FLOATS, dtype=torch.float32, device=t.device)
    C = t.shape[-1]
    _lib.check(_lib.lib().otgan_absmax_f32(t.data_ptr(), t.numel() // C, C, C, rec.data_ptr(), _lib.stream_ptr()),
               "absmax")
    return rec


# ---- amax records written by the kernel that PRODUCES a tensor (otgan_layers.h: y_amax_out / dx_amax_out) -------------
# A Winograd layer scales its fp16 GEMM operands by the largest magnitude of the tensor they are a transform of.  The
# record used to come from one more pass over the tensor (absmax_record: 20 launches, 0.33 ms of a DCGAN step); now the
# kernel that writes the tensor -- GLU forward / backward, the output transforms of the strided layers, the RGB-in
# layer, the feature head's backward -- leaves it in a zeroed slot, and the tensor carries the slot to its consumer
# as a Python attribute (checked against the tensor's version counter; a tensor that arrives without one, e.g.
# through a view or from outside, is reduced as before).
AMAX_RECORD_FLOATS = 512     # otgan_layers.h: OTGAN_AMAX_RECORD_FLOATS
_AMAX_SLOTS = 256
_amax_pool = {}       # device -> [zeroed [slots, AMAX_RECORD_FLOATS] tensor, next free slot]
# Constants since round 6 (they were OTGAN_* environment switches while each mechanism was being A/B-ed; the measurements are in
# DESIGN / docs/history).  Tests that compare a mechanism with its predecessor patch the attribute in-process.
_FUSED_AMAX = True        # producers leave the amax record of what they write
_GLU_COLSUM = True        # the gate's backward kernel also leaves the column sums (bias gradient) of what it writes
_GLU_FUSED = True         # the Winograd output transform writes the gated product beside y
_GRAD_INPLACE = True      # DenseBlockFunction.backward walks ONE gradient buffer
DENSE_SPLIT = True        # dense blocks: block input and finished halves through wide Winograd convolutions (_split_block_plan)
DENSE_GROUP = None        # growth outputs per wide convolution; None = half the block
DENSE_AMAX = True         # one record array per dense block, filled by the kernels that write its channels
WN_BATCHED = True         # one weight-norm launch per dense block


def colsum_of(t):
    """Column sums of `t` left by its producer (GluFunction.backward), or None."""
    tag = getattr(t, "_otgan_colsum", None)
    if tag is None or tag[1] != t._version:
        return None
    return tag[0]


def amax_slot(device):
    """A zeroed amax record (one launch zeroes 256 of them).  The pool is shared by the streams of a step (trainer.py runs two
    chains of a step on a second stream): a stream other than the one that zeroed the pool waits for that launch."""
    ent = _amax_pool.get(device)
    cur = torch.cuda.current_stream(device)
    if ent is None or ent[1] >= _AMAX_SLOTS:
        pool = torch.zeros((_AMAX_SLOTS, AMAX_RECORD_FLOATS), dtype=torch.float32, device=device)
        ent = [pool, 0, cur.cuda_stream, cur.record_event(), set()]
        _amax_pool[device] = ent
    elif cur.cuda_stream != ent[2] and cur.cuda_stream not in ent[4]:
        cur.wait_event(ent[3])
        ent[0].record_stream(cur)
        ent[4].add(cur.cuda_stream)
    rec = ent[0][ent[1]]
    ent[1] += 1
    return rec


def amax_slots(device, n):
    """n CONSECUTIVE zeroed records ([n, AMAX_RECORD_FLOATS]) from the pool: the record arrays of a dense block's passes
    (one row per layer) without a zeroing launch each."""
    if n > _AMAX_SLOTS:
        return torch.zeros((n, AMAX_RECORD_FLOATS), dtype=torch.float32, device=device)
    ent = _amax_pool.get(device)
    if ent is not None and ent[1] + n > _AMAX_SLOTS:
        ent[1] = _AMAX_SLOTS                      # not enough left in this pool: the next draw opens a new one
    first = amax_slot(device)                     # (opens / joins the pool; takes slot ent[1] - 1)
    ent = _amax_pool[device]
    i0 = ent[1] - 1
    ent[1] = i0 + n
    return ent[0][i0:i0 + n]


def reset_amax_pool():
    """Forget the current pool of zeroed records: the next amax_slot() allocates (and zeroes) a new one.  A step captured in
    a hipGraph calls this at the start of the capture, so that the pool's zeroing is part of the graph and every replay starts
    from zeroed records, and at its end, so that eager code never draws from a pool a replay re-zeroes."""
    _amax_pool.clear()


def tag_amax(t, rec):
    """`t` was just written by a kernel that max-accumulated |t| into rec[0]."""
    t._otgan_amax = (rec, t._version)
    return t


def amax_of(t):
    """The producer's amax record of `t`, or None (no record, or the tensor was modified since)."""
    tag = getattr(t, "_otgan_amax", None)
    if tag is None or tag[1] != t._version:
        return None
    return tag[0]


def carry_amax(dst, src):
    """`dst` holds the same values as `src` (a view / reshape): the record still describes it."""
    rec = amax_of(src)
    if rec is not None:
        tag_amax(dst, rec)
    return dst


def amax_fused(desc, which):
    return _FUSED_AMAX and bool(_lib.lib().otgan_conv2d_amax_fused(ctypes.byref(desc), which))


def shared_x_operand(desc, device):
    """Buffer for `otgan_conv_desc::x_operand` (the forward pass leaves its transformed input there, the weight
    gradient of the same x reads it back instead of transforming x again), or None when the layer's two passes do not
    share an operand.  Sets the descriptor field."""
    nbytes = _lib.lib().otgan_conv2d_operand_bytes(ctypes.byref(desc))
    if not nbytes:
        return None
    buf = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device)
    desc.x_operand = buf.data_ptr()
    return buf


def prepare_filters(desc, which, w):
    """Winograd-domain filters of a layer's forward (which=0, from wT) or dgrad (which=1, from w) pass, or None
    when the pass does not run as a Winograd GEMM (otgan_layers.h)."""
    L = _lib.lib()
    nbytes = L.otgan_conv2d_filter_bytes(ctypes.byref(desc), which)
    if not nbytes:
        return None
    buf = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=w.device)
    rec = amax_of(w)            # left by weightnorm_fwd on the un-folded w / wT (folded weights are another tensor)
    desc.w_amax = rec.data_ptr() if rec is not None else None
    _lib.check(L.otgan_conv2d_prepare_filters_f32(ctypes.byref(desc), which, w.data_ptr(), buf.data_ptr(),
                                                  buf.numel() * 4, _lib.stream_ptr()), "prepare_filters")
    desc.w_amax = None
    return buf


def conv_wgrad_raw(desc, x, cmap, dy, dw):
    L = _lib.lib()
    need = L.otgan_conv2d_workspace_bytes(ctypes.byref(desc), 2)
    ws = workspace(need, dy.device)
    _lib.check(L.otgan_conv2d_wgrad_f32(ctypes.byref(desc), x.data_ptr(), _lib.ptr(cmap),
                                        dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), ws.numel(),
                                        _lib.stream_ptr()), "conv2d_wgrad")


def fold_weights(desc, w):
    """(weff, weffT) of an upsampling layer that the library folds (otgan_layers.h); `w` is HWIO."""
    nfold = _lib.lib().otgan_conv2d_folded_weight_elems(ctypes.byref(desc))
    weff = torch.empty(nfold, dtype=w.dtype, device=w.device)
    weffT = torch.empty(nfold, dtype=w.dtype, device=w.device)
    _lib.check(_lib.lib().otgan_conv2d_fold_weights_f32(ctypes.byref(desc), w.data_ptr(), weff.data_ptr(),
                                                        weffT.data_ptr(), _lib.stream_ptr()), "fold_weights")
    return weff, weffT


# ------------------------------------------------------------------------------- second stream for weight gradients
# A layer's weight-gradient chain (adjoint transform of dy, t-leading GEMM, adjoint filter transform, weight-norm backward) feeds
# nothing but the optimiser, while its input-gradient chain is what the next layer's backward waits for.  With a side stream set
# (trainer.py: single-process runs), Conv2dFunction.backward issues the weight-gradient chain there: the two chains alternate
# HBM-bound transforms and matrix-bound GEMMs, and on one in-order stream every kernel also waits for the previous one's last
# workgroup -- on two streams each chain's kernels start in the other's tails.  Same kernels, same arguments, same results.
# The caller joins the side stream before it reads the gradients (ops.join_side_stream).
SIDE_STREAM = None
# Variables whose weight gradient this backward pass has already produced on the side stream.  A variable used by TWO nodes of one
# graph (weight sharing, the critic called twice under grad) has its gradients accumulated by the autograd engine on the MAIN
# stream as soon as the second node returns: that node therefore makes the main stream wait for the side stream before it hands
# its gradients back (today's models use every variable once per graph, so the wait never happens).
_SIDE_SEEN = set()


def join_side_stream(tensors=()):
    """Make the current stream wait for the side stream's work and tell the allocator that `tensors` (allocated there) are
    used here from now on."""
    _SIDE_SEEN.clear()
    if SIDE_STREAM is None:
        return
    cur = torch.cuda.current_stream()
    cur.wait_stream(SIDE_STREAM)
    for t in tensors:
        if t is not None and t.is_cuda:
            t.record_stream(cur)


# The gated product a convolution wrote for the glu() that follows it (glu_hint) rides on the convolution's output as an
# attribute until GluFunction.forward claims it.  A caller that passes the hint and then never applies glu to that very
# tensor would keep the product alive as long as the output: the next convolution's forward pass drops it.
_PENDING_GLU = [None]


def _drop_unclaimed_glu():
    ref = _PENDING_GLU[0]
    if ref is not None:
        t = ref()
        if t is not None and hasattr(t, "_otgan_glu"):
            del t._otgan_glu
        _PENDING_GLU[0] = None


# ------------------------------------------------------------------------------- conv2d / dense
# Test instrumentation (tests/test_dist_gpu.py): a list here makes every layer with a ReLU-type pre-activation and the feature
# head append the sign pattern of its input (a CPU bool tensor per call, in call order).  Two fp32 evaluations of one step
# agree to rounding EXCEPT where a pre-activation that is zero to rounding lands on different sides: the unit's gradient then
# switches branches (CReLU: [relu(x), relu(-x)], nn.py:198-200) -- counting those units separates arithmetic from flips.
SIGN_TRACE = None


def _trace_signs(x):
    if SIGN_TRACE is not None:
        SIGN_TRACE.append((x.detach() > 0).cpu())


class Conv2dFunction(torch.autograd.Function):
    """y = conv2d(preact(upsample(x)), g*V/||V||) + b     (reference nn.py:327-338).

    x: [N,H,W,C] NHWC (C may be the concatenation of a list, `segs` gives the element sizes);
    V: [KH,KW,Cin_eff,Cout]; g, b: [Cout]."""

    @staticmethod
    def forward(ctx, x, V, g, b, stride, upsample, preact, segs, glu_hint=False, grow=0):
        _need_cuda(x, V, g, b)
        _drop_unclaimed_glu()
        x = x.contiguous()
        if preact in (1, 4):
            _trace_signs(x)
        N, H, W, C = x.shape
        # a dense layer passes its [Cin_eff, Cout] variable as is (a 1x1 filter): the weight cache is keyed by
        # the parameter OBJECT, so a fresh .view() per call would miss every time and pin a new entry
        KH, KW, Cin_eff, Cout = V.shape if V.dim() == 4 else (1, 1) + tuple(V.shape)
        if Cin_eff != C * (2 if preact in DOUBLED else 1):
            raise ValueError(f"weight expects {Cin_eff} effective input channels, input gives "
                             f"{C * (2 if preact in DOUBLED else 1)}")
        V2d = V.contiguous().view(KH * KW * Cin_eff, Cout)
        OH, OW = out_hw(H, W, upsample, stride)
        grow = int(grow) if (grow and Cout % 4 == 0 and not glu_hint and _GROW_IN_PLACE) else 0
        if grow:
            # the caller appends `grow` channels next (a dense block, nn.dense_block): the output is the channel prefix of a
            # buffer with room for them (ldy of the conv ABI) -- the block grows there in place instead of copying its input
            full = torch.empty((N, OH, OW, Cout + grow), dtype=x.dtype, device=x.device)
            full._otgan_grow = True
            y = full[..., :Cout]
        else:
            y = torch.empty((N, OH, OW, Cout), dtype=x.dtype, device=x.device)
        desc = make_desc(x, C, upsample, KH, KW, stride, Cout, Cout + grow, 0, preact,
                         segs if (segs and not upsample) else None)
        # with upsample the reference concatenates the list BEFORE the pre-activation
        # (nn.py:235-237), so the doubled ordering is [x_all, -x_all], not per element
        cmap, inv = channel_maps(segs if (segs and not upsample) else (C,), preact, x.device)
        nfold = _lib.lib().otgan_conv2d_folded_weight_elems(ctypes.byref(desc))

        def compute():
            w, wT, inv_norm = weightnorm_fwd(V2d, g)
            wd = w                       # operand of dgrad
            # Winograd passes of a folded layer take filters made from the UN-folded weights (which = 2 forward,
            # 3 dgrad: no fold pass at all); a layer may have only the forward one (3x3 on an upsampled input)
            unf_f = bool(nfold) and _lib.lib().otgan_conv2d_filter_bytes(ctypes.byref(desc), 2) > 0
            unf_b = bool(nfold) and _lib.lib().otgan_conv2d_filter_bytes(ctypes.byref(desc), 3) > 0
            if nfold and not (unf_f and unf_b):
                # conv o upsample == four parity-class convs with pre-summed taps (otgan_layers.h)
                wd_f, wT_f = fold_weights(desc, w)
                if not unf_b:
                    wd = wd_f
                if not unf_f:
                    wT = wT_f
            # Winograd-domain filters live as long as the normalised weights (the critic's survive the five
            # generator steps between its updates); the dgrad ones are made by the first backward that needs them.
            return wd, wT, inv_norm, {"fwd": prepare_filters(desc, 2 if unf_f else 0, wT), "bwd": None,
                                      "bwd_done": False, "bwd_which": 3 if unf_b else 1}

        wd, wT, inv_norm, filt = cached_weights(V, g, compute)
        w_rec = amax_of(wT)              # left by the weight-norm kernel (same values in w and wT); the implicit-GEMM passes'
        ctx.w_rec = w_rec                # two-piece fp16 loop scales the weights by it (else it reduces them itself)
        desc.w_amax = w_rec.data_ptr() if w_rec is not None else None
        ctx.x_rec = None
        if filt["fwd"] is not None and x.is_contiguous() and C % 4 == 0:
            # Winograd passes: the record x's producer left (else one reduction of x), for the forward pass now and
            # the weight gradient later
            ctx.x_rec = amax_of(x)
            if ctx.x_rec is None:
                ctx.x_rec = absmax_record(x)
            desc.x_amax = ctx.x_rec.data_ptr()
        # 288 GB of HBM: keep the transformed input of the Winograd passes for the weight gradient (0.15-0.6 GB a layer)
        ctx.x_op = None
        if (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]) and cmap is None and filt["fwd"] is not None:
            ctx.x_op = shared_x_operand(desc, x.device)
        if ctx.x_rec is None:
            x_tag = amax_of(x)               # implicit-GEMM passes scale their two-piece operands by it (else they reduce x)
            desc.x_amax = x_tag.data_ptr() if x_tag is not None else None
        y_rec = None
        if amax_fused(desc, 0):
            y_rec = amax_slot(x.device)          # the kernel that writes y also leaves max |y| (for the next layer)
            desc.y_amax_out = y_rec.data_ptr()
        y_glu = None
        if glu_hint and _GLU_FUSED and cmap is None and _lib.lib().otgan_conv2d_glu_fused(ctypes.byref(desc)):
            # the caller applies a gated linear unit to y next (models/dcgan.py:50): the kernel that writes y leaves the
            # gated product (and its amax record) too; GluFunction.forward picks it up instead of reading y again
            y_glu = torch.empty((N, OH, OW, Cout // 2), dtype=x.dtype, device=x.device)
            glu_rec = amax_slot(x.device) if _FUSED_AMAX else None
            desc.glu_out, desc.glu_amax_out = y_glu.data_ptr(), _lib.ptr(glu_rec)
        conv_fwd_raw(desc, x, cmap, wT, b, y, filt["fwd"])
        if y_glu is not None:
            desc.glu_out = desc.glu_amax_out = None
            if glu_rec is not None:
                tag_amax(y_glu, glu_rec)
            y._otgan_glu = (y_glu, y._version)
            _PENDING_GLU[0] = weakref.ref(y)
        desc.y_amax_out = None
        if ctx.x_rec is None:
            desc.x_amax = None
        if y_rec is not None:
            tag_amax(y, y_rec)
        if (SIDE_STREAM is not None and ctx.needs_input_grad[0] and not filt["bwd_done"] and
                _lib.lib().otgan_conv2d_filter_bytes(ctypes.byref(desc), filt["bwd_which"]) > 0):
            # the input-gradient pass will want its Winograd-domain filters (a function of the weights alone): made now on the
            # side stream, under this forward pass, instead of on the backward pass's critical path
            cur = torch.cuda.current_stream()
            SIDE_STREAM.wait_stream(cur)                 # (the normalised weights were written on this stream)
            with torch.cuda.stream(SIDE_STREAM):
                filt["bwd"], filt["bwd_done"] = prepare_filters(desc, filt["bwd_which"], wd), True
                filt["bwd_event"] = SIDE_STREAM.record_event()
            wd.record_stream(SIDE_STREAM)
            desc.w_amax = w_rec.data_ptr() if w_rec is not None else None      # (prepare_filters cleared it)
        ctx.save_for_backward(x, V2d, g, wd, inv_norm)
        ctx.filt = filt
        ctx.desc, ctx.cmap, ctx.inv = desc, cmap, inv
        ctx.vshape = V.shape
        ctx.has_b = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, V2d, g, w, inv_norm = ctx.saved_tensors
        desc = ctx.desc
        # dy may be the channel prefix of a dense block's gradient buffer (DenseBlockFunction.backward hands it on without a
        # copy): read in place through the ABI's channel stride.  Anything else is made contiguous.
        ld_fwd = desc.ldy
        ld = channel_prefix_stride(dy)
        if ld is None:
            rec = amax_of(dy)
            dy = dy.contiguous()
            ld = dy.shape[-1]
            if rec is not None:
                tag_amax(dy, rec)
        desc.ldy = ld
        try:
            return Conv2dFunction._backward_impl(ctx, dy, ld, x, V2d, g, w, inv_norm)
        finally:
            desc.ldy = ld_fwd

    @staticmethod
    def _backward_impl(ctx, dy, ld, x, V2d, g, w, inv_norm):
        desc = ctx.desc
        dx = dV = dg = db = None
        dy_rec = None
        if ctx.x_rec is not None and dy.shape[-1] % 4 == 0:
            dy_rec = amax_of(dy)              # left by dy's producer, else reduced here; shared by dgrad and wgrad
            if dy_rec is None:
                dy_rec = (absmax_record(dy) if dy.is_contiguous() else
                          absmax_record_strided(dy.data_ptr(), dy.numel() // dy.shape[-1], dy.shape[-1], ld, dy.device))
            desc.dy_amax = dy_rec.data_ptr()
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            filt = ctx.filt
            if not filt["bwd_done"]:
                filt["bwd"], filt["bwd_done"] = prepare_filters(desc, filt["bwd_which"], w), True
            elif filt.get("bwd_event") is not None:
                # made on the side stream during a forward pass: this stream waits for that launch (once is enough, but the
                # wait is free when the event has long completed) and the buffer is in use here from now on
                torch.cuda.current_stream().wait_event(filt["bwd_event"])
                if filt["bwd"] is not None:
                    filt["bwd"].record_stream(torch.cuda.current_stream())
            dx_rec = None
            # (a list input keeps a layer off the Winograd passes; the implicit-GEMM epilogue writes every real channel
            # once whatever the channel map, round 4)
            if amax_fused(desc, 1) and (ctx.inv is None or filt["bwd"] is None):
                dx_rec = amax_slot(x.device)
                desc.dx_amax_out = dx_rec.data_ptr()
            desc.w_amax = ctx.w_rec.data_ptr() if ctx.w_rec is not None else None    # (prepare_filters cleared it)
            conv_dgrad_raw(desc, dy, w, x, ctx.inv, dx, x.shape[3], False, filt["bwd"])
            desc.dx_amax_out = None
            if dx_rec is not None:
                tag_amax(dx, dx_rec)
        need_w = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        need_b = ctx.has_b and ctx.needs_input_grad[3]
        side = SIDE_STREAM if (need_w or need_b) else None
        if side is not None:
            # fork: everything dy depends on is enqueued on the current stream
            side.wait_stream(torch.cuda.current_stream())
            for t in (dy, x, ctx.x_op, dy_rec, ctx.x_rec):
                if t is not None:
                    t.record_stream(side)        # (dy and the kept operand are freed when this node is: not before the side stream is done)
        with (torch.cuda.stream(side) if side is not None else _NULL_CTX):
            if need_w:
                dw = torch.empty_like(V2d)
                conv_wgrad_raw(desc, x, ctx.cmap, dy, dw)
                dV2d, dg = weightnorm_bwd(V2d, g, inv_norm, dw)
                dV = dV2d.view(ctx.vshape)
            if need_b:
                db = colsum_of(dy)
                if db is None:
                    rows = dy.numel() // dy.shape[-1]
                    db = colsum(dy.data_ptr(), rows, dy.shape[-1], ld, dy.device)
        if side is not None:
            key = V2d.untyped_storage().data_ptr() + V2d.storage_offset()
            if key in _SIDE_SEEN:       # second use of this variable in one pass: the engine will add the two gradients on the main stream
                torch.cuda.current_stream().wait_stream(side)
            _SIDE_SEEN.add(key)
        return dx, dV, dg, db, None, None, None, None, None, None


def conv2d_op(x, V, g, b, stride=1, upsample=False, preact=0, segs=None, glu_hint=False, grow=0):
    """glu_hint: the caller feeds the result to glu() next -- where the library can, the layer's output kernel also
    writes the gated product and glu() takes it from there (same values; the hint changes nothing else).
    grow: the caller appends that many channels next (nn.dense_block): the result is the channel prefix of a buffer with
    room for them, same values."""
    return Conv2dFunction.apply(x, V, g, b, int(stride), bool(upsample), int(preact),
                                tuple(segs) if segs else None, bool(glu_hint), int(grow))


def channel_prefix_stride(t):
    """ld when `t` [N,H,W,C] is a channel slice of a contiguous [N,H,W,ld] buffer that the kernels can read in place
    (16-byte aligned, ld % 4 == 0), else None (also for contiguous tensors: nothing to do)."""
    if t.dim() != 4 or t.is_contiguous():
        return None
    N, H, W, C = t.shape
    ld = t.stride(2)
    if (t.stride(3) != 1 or ld < C or ld % 4 or C % 4 or t.stride(1) != W * ld or t.stride(0) != H * W * ld
            or t.data_ptr() % 16):
        return None
    return int(ld)


def grown_buffer(x, Ctot, claim=False):
    """The [N,H,W,Ctot] buffer a convolution with `grow` allocated around its output `x` (its channel prefix), or None."""
    base = x._base if x.dim() == 4 else None
    if base is None or not getattr(base, "_otgan_grow", False) or not base.is_contiguous():
        return None
    N, H, W, C = x.shape
    if tuple(base.shape) != (N, H, W, Ctot) or x.data_ptr() != base.data_ptr() or channel_prefix_stride(x) != Ctot:
        return None
    if claim:
        base._otgan_grow = False       # one block grows in a buffer
    return base


def dense_op(x, V, g, b, preact=0, segs=None):
    """x: [N, Cin]; V: [Cin_eff, Cout]  (reference nn.py:314-325) -- a 1x1 conv on a 1x1 image."""
    N, C = x.shape
    y = Conv2dFunction.apply(x.view(N, 1, 1, C), V, g, b, 1, False,
                             int(preact), tuple(segs) if segs else None)
    return y.view(N, -1)


def absmax_record_strided(t_ptr, rows, C, ld, device):
    """amax record of a channel slice [rows][C] (row stride ld floats) of a larger NHWC buffer."""
    rec = torch.empty(AMAX_RECORD_FLOATS, dtype=torch.float32, device=device)
    _lib.check(_lib.lib().otgan_absmax_f32(t_ptr, rows, C, ld, rec.data_ptr(), _lib.stream_ptr()), "absmax")
    return rec


def _input_row_order(segs0, preact, device):
    """Effective-channel rows of the block input inside a growth layer's weight tensor, in the order the
    single-tensor convolution kernels use ([act(x_all), act(-x_all)]): the reference interleaves per list element
    ([x0, -x0, x1, -x1, ...], nn.py:198-200).  None = already in that order."""
    segs0 = tuple(int(c) for c in segs0)
    if preact not in DOUBLED or len(segs0) <= 1:
        return None
    key = ("rows", segs0, device)
    hit = _map_cache.get(key)
    if hit is None:
        pos, neg, off = [], [], 0
        for c in segs0:
            pos += [2 * off + i for i in range(c)]
            neg += [2 * off + c + i for i in range(c)]
            off += c
        hit = torch.tensor(pos + neg, dtype=torch.int64, device=device)
        _map_cache[key] = hit
    return hit


def _aligned16(*ts):
    return all(t is None or t.data_ptr() % 16 == 0 for t in ts)


def weightnorm_fwd_block(V2ds, gs):
    """weightnorm_fwd of every 16-output layer of a dense block in ONE launch (otgan_weightnorm_fwd_batched16_f32);
    the results are views of three block-wide buffers.  Returns [(w, wT, inv_norm)] per layer."""
    L = len(V2ds)
    Ks = [int(v.shape[0]) for v in V2ds]
    dev, dt = V2ds[0].device, V2ds[0].dtype
    w_all = torch.empty(sum(Ks) * 16, dtype=dt, device=dev)
    wT_all = torch.empty(sum(Ks) * 16, dtype=dt, device=dev)
    inv_all = torch.empty(L * 16, dtype=dt, device=dev)
    arr = (WnFwdLayer * L)()
    out, off = [], 0
    wp, tp, ip = w_all.data_ptr(), wT_all.data_ptr(), inv_all.data_ptr()
    for k in range(L):
        n = Ks[k] * 16
        a = arr[k]
        a.V, a.g, a.K = V2ds[k].data_ptr(), gs[k].data_ptr(), Ks[k]
        a.w, a.wT, a.inv = wp + 4 * off, tp + 4 * off, ip + 64 * k
        out.append((w_all[off:off + n].view(Ks[k], 16), wT_all[off:off + n].view(16, Ks[k]), inv_all[16 * k:16 * k + 16]))
        off += n
    _lib.check(_lib.lib().otgan_weightnorm_fwd_batched16_f32(arr, L, _lib.stream_ptr()), "weightnorm_fwd_batched")
    return out


def weightnorm_bwd_block(V2ds, gs, invs, parts):
    """weightnorm_bwd of every 16-output layer of a dense block in ONE launch.  parts[k]: up to three
    (tensor_or_pointer, perm, nrows, rstride) pieces of layer k's weight gradient (otgan_layers.h: otgan_wn_part).
    Returns ([dV2d_k], [dg_k])."""
    L = len(V2ds)
    Ks = [int(v.shape[0]) for v in V2ds]
    dev, dt = V2ds[0].device, V2ds[0].dtype
    dV_all = torch.empty(sum(Ks) * 16, dtype=dt, device=dev)
    dg_all = torch.empty(L * 16, dtype=dt, device=dev)
    arr = (WnBwdLayer * L)()
    dVs, dgs, off = [], [], 0
    vp, gp = dV_all.data_ptr(), dg_all.data_ptr()
    for k in range(L):
        n = Ks[k] * 16
        a = arr[k]
        a.V, a.g, a.inv = V2ds[k].data_ptr(), gs[k].data_ptr(), invs[k].data_ptr()
        a.dV, a.dg = vp + 4 * off, gp + 64 * k
        a.taps = 9
        a.Ceff = Ks[k] // 9
        for i, (ptr, perm, nrows, rstride) in enumerate(parts[k]):
            q = a.part[i]
            q.p, q.perm, q.nrows, q.rstride = ptr, (perm.data_ptr() if perm is not None else None), nrows, rstride
        dVs.append(dV_all[off:off + n].view(Ks[k], 16))
        dgs.append(dg_all[16 * k:16 * k + 16])
        off += n
    _lib.check(_lib.lib().otgan_weightnorm_bwd_batched16_f32(arr, L, _lib.stream_ptr()), "weightnorm_bwd_batched")
    return dVs, dgs


def _row_order_back(order):
    """int32 inverse of `_input_row_order`: position, in the single-tensor order, of every row of the reference's
    per-element order (None stays None)."""
    if order is None:
        return None
    key = ("back", order.data_ptr())
    hit = _map_cache.get(key)
    if hit is None:
        back = torch.empty_like(order)
        back[order] = torch.arange(order.numel(), device=order.device)
        hit = (order, back.to(torch.int32))      # holding `order` keeps its data_ptr from being reused
        _map_cache[key] = hit
    return hit[1]


_block_cache = OrderedDict()   # id(V of layer 0) -> (per-layer normalised weights it was made from, value)
DENSE16_MAX_BATCH = 16         # otgan_layers.h: OTGAN_DENSE16_MAX_BATCH (chain layers per otgan_dense16_prepare_* call)
DENSE16_MAX_CHAIN = 17         # slices per otgan_dense16_chain_{fwd,bwd}_f32 call (conv.hip)


def _wide_operands(per_layer, layers, row0, nrows, order, F, desc):
    """One wide 3x3 convolution gathered from a dense block: rows [row0, row0 + nrows) of the normalised weights of
    `layers` (the effective channels of one finished channel group), side by side.  w [9*nrows][len(layers)*F],
    wT [len(layers)*F][9*nrows], rows re-ordered to the single-tensor order when `order` is given."""
    layers = list(layers)
    nl = len(layers)
    n = nl * F
    ref = per_layer[layers[0]][0]
    w = torch.empty((9 * nrows, n), dtype=ref.dtype, device=ref.device)
    wT = torch.empty((n, 9 * nrows), dtype=ref.dtype, device=ref.device)
    map32 = None
    if order is not None:
        key = ("rows32", order.data_ptr())
        hit = _map_cache.get(key)
        if hit is None:
            hit = (order, order.to(torch.int32))      # holding `order` keeps its data_ptr from being reused
            _map_cache[key] = hit
        map32 = hit[1]
    # two launches (otgan_gather3d_batched_f32: one segment per layer) instead of a cat, an index_select and their copies
    # per operand -- 0.33 ms of framework kernels per DenseNet step before round 6; same values, bit for bit
    segs_w, segs_t = [], []
    for j, k in enumerate(layers):
        w_k, wT_k = per_layer[k][0], per_layer[k][1]
        assert w_k.is_contiguous() and wT_k.is_contiguous()
        ceff = w_k.shape[0] // 9
        assert row0 + nrows <= ceff
        # w_k [9][ceff][F] -> w [9][nrows][nl F] at column j F;  wT_k [F 9][ceff] -> wT [nl F 9][nrows] at row j F 9
        segs_w.append((w_k.data_ptr(), w.data_ptr() + 4 * j * F, 9, ceff * F, F, nrows * n, n))
        segs_t.append((wT_k.data_ptr(), wT.data_ptr() + 4 * j * F * 9 * nrows, F * 9, ceff, 1, nrows, 1))
    # the first gather also leaves the amax record of the gathered weights (w and wT hold the same elements): the filter
    # transforms of the wide convolution take it from there instead of reducing the weights again (15 launches per DenseNet step)
    rec = amax_slot(ref.device) if _FUSED_AMAX else None
    gather3d_batched(segs_w, nrows, F, row0, map32, rec)
    gather3d_batched(segs_t, nrows, 1, row0, map32)
    if rec is not None:
        tag_amax(w, rec)
        tag_amax(wT, rec)
    return {"w": w, "wT": wT, "fwd": prepare_filters(desc, 0, wT), "bwd": None, "bwd_done": False}


def _split_block_weights(Vs, per_layer, plan, F):
    """Operands of a dense block computed as "wide convolutions of finished channel groups + short growth chains"
    (DenseBlockFunction): per wide convolution the gathered weights and their Winograd-domain filters, per layer the
    rows of its own chain as contiguous tensors.  Cached for as long as the per-layer normalised weights are."""
    key = id(Vs[0])
    ws = [pl[0] for pl in per_layer]
    hit = _block_cache.get(key)
    if hit is not None and len(hit[0]) == len(ws) and all(a is b for a, b in zip(hit[0], ws)) and hit[2] == plan["key"]:
        _block_cache.move_to_end(key)
        return hit[1]
    L = len(per_layer)
    val = {"wide": [_wide_operands(per_layer, range(wd["d0"], L), wd["row0"], wd["nrows"], wd["order"], F, wd["desc"])
                    for wd in plan["wide"]],
           "w_g": [None] * L, "wT_g": [None] * L}
    # every layer's own-chain rows (rows r0 .. of each of the 9 taps) as contiguous tensors: two flat buffers carved
    # into the layers' slices, filled by ONE batched strided copy (28 slices per 16-layer block; as framework copies these
    # were 100 of the DenseNet step's launches)
    own = [k for k in range(L) if plan["own_len"][k]]
    if own:
        dev, dt = per_layer[own[0]][0].device, per_layer[own[0]][0].dtype
        sizes = [9 * (per_layer[k][0].shape[0] // 9 - plan["own_row0"][k]) * F for k in own]
        flat_w = torch.empty(sum(sizes), dtype=dt, device=dev)
        flat_wT = torch.empty(sum(sizes), dtype=dt, device=dev)
        segs, off = [], 0
        for k, sz in zip(own, sizes):
            r0 = plan["own_row0"][k]
            ceff = per_layer[k][0].shape[0] // 9
            n_own = ceff - r0
            w_k, wT_k = per_layer[k][0], per_layer[k][1]
            assert w_k.is_contiguous() and wT_k.is_contiguous()
            val["w_g"][k] = flat_w[off:off + sz].view(9 * n_own, F)
            val["wT_g"][k] = flat_wT[off:off + sz].view(F, 9 * n_own)
            # w [9][ceff][F] -> [9][n_own][F]: 9 rows of n_own * F floats;  wT [F * 9][ceff] -> [F * 9][n_own]
            segs.append((w_k.data_ptr() + 4 * r0 * F, val["w_g"][k].data_ptr(), 9, n_own * F, ceff * F, n_own * F))
            segs.append((wT_k.data_ptr() + 4 * r0, val["wT_g"][k].data_ptr(), 9 * F, n_own, ceff, n_own))
            off += sz
        copy2d_batched(segs)
        val["_own_flat"] = (flat_w, flat_wT)
        val["h2"] = [None] * L
        if plan.get("h2"):
            # the chain weights pre-split into two scaled fp16 pieces in MFMA fragment order: one launch per block and
            # weight update (otgan_dense16_prepare_filters_f32)
            lib = _lib.lib()
            nsl = [plan["own_len"][k] for k in own]
            fbytes = [int(lib.otgan_dense16_filter_bytes(n)) for n in nsl]
            flat_q = torch.empty(sum(fbytes), dtype=torch.uint8, device=dev)
            off = 0
            for k, nb in zip(own, fbytes):
                val["h2"][k] = flat_q[off:off + nb]
                off += nb
            n = len(own)
            pw = (ctypes.c_void_p * n)(*[val["wT_g"][k].data_ptr() for k in own])
            pn = (ctypes.c_int * n)(*nsl)
            pf = (ctypes.c_void_p * n)(*[val["h2"][k].data_ptr() for k in own])
            _lib.check(lib.otgan_dense16_prepare_filters_f32(ctypes.cast(pw, ctypes.c_void_p), ctypes.cast(pn, ctypes.c_void_p),
                                                              ctypes.cast(pf, ctypes.c_void_p), n, _lib.stream_ptr()),
                       "dense16_prepare_filters")
            val["_h2_flat"] = flat_q
    _block_cache[key] = (ws, val, plan["key"])
    while len(_block_cache) > 64:
        _block_cache.popitem(last=False)
    return val


def _split_block_plan(N, H, W, C0, L, F, segs0, preact, device):
    """How a dense block is cut (None: not at all).  The pre-activation of layer k is linear in act(every earlier
    channel), so the share of a FINISHED channel group in all later layers is one 3x3 convolution group -> (later
    layers) * F, wide enough for the Winograd F(4x4,3x3) passes:
      * the block input (C0 channels) into all L layers, before the chain starts;
      * every group of `DENSE_GROUP` (default L/2: the first half into the second half) consecutive growth outputs
        into all layers after the group, once its last layer is done -- as long as the library takes that convolution
        on its Winograd path.
    What stays on the 16-output growth kernels is each layer's chain inside its own group."""
    if not DENSE_SPLIT or F != 16 or L < 2 or any(int(c) % 4 for c in segs0):
        return None
    mult = 2 if preact in DOUBLED else 1
    Ctot = C0 + L * F
    lib = _lib.lib()
    desc_in = ConvDesc(N, H, W, C0, Ctot, 0, 3, 3, 1, L * F, Ctot, C0, preact, 1)
    if not lib.otgan_conv2d_filter_bytes(ctypes.byref(desc_in), 0):
        return None
    wide = [{"desc": desc_in, "x_off": 0, "C": C0, "d0": 0, "row0": 0, "nrows": C0 * mult,
             "order": _input_row_order(segs0, preact, device), "accumulate": 0, "after": -1}]
    # measured on the DenseNet step (L = 16): halves 51.9 ms, groups of four 52.9 (with 64-column convolutions allowed
    # 53.5), pairs 60.7, block input only 58.2 -- a wide convolution with K = 128 is bound by its transforms
    group = (L + 1) // 2 if DENSE_GROUP is None else int(DENSE_GROUP)
    g0 = [0] * L          # growth outputs [0, g0[k]) reach layer k through wide convolutions
    for s0 in range(0, L, group) if group > 0 else ():
        s1 = min(s0 + group, L)
        if s1 >= L:
            break
        n = s1 - s0
        desc = ConvDesc(N, H, W, n * F, Ctot, 0, 3, 3, 1, (L - s1) * F, Ctot, C0 + s1 * F, preact, 1)
        if not lib.otgan_conv2d_filter_bytes(ctypes.byref(desc), 0):
            break         # later groups feed fewer layers still
        wide.append({"desc": desc, "x_off": C0 + s0 * F, "C": n * F, "d0": s1, "row0": (C0 + s0 * F) * mult,
                     "nrows": n * F * mult, "order": _input_row_order((F,) * n, preact, device), "accumulate": 1,
                     "after": s1 - 1})
        for k in range(s1, L):
            g0[k] = s1
    for wd in wide:
        wd["back"] = _row_order_back(wd["order"])
    # the chains on the two-scaled-fp16-piece kernel (CReLU, 16-channel list elements; otgan_layers.h: list_width)
    probe = ConvDesc(N, H, W, F, Ctot, 0, 3, 3, 1, F, Ctot, C0, preact, 1)
    probe.y_accumulate, probe.list_width = 1, F
    h2 = bool(lib.otgan_dense16_h2_ok(ctypes.byref(probe))) and DENSE_AMAX and _FUSED_AMAX
    # the library prepares at most OTGAN_DENSE16_MAX_BATCH chain layers per call (forward and by-slice backward filters) and
    # a chain call takes at most 17 slices per group: longer blocks / groups (layers_per_block > 18 at the default grouping)
    # keep the fp32 growth kernels
    own_len = [k - g0[k] for k in range(L)]
    starts = sorted({wd["d0"] for wd in wide}) + [L]
    if sum(1 for n in own_len if n) > DENSE16_MAX_BATCH or max(b - a for a, b in zip(starts[:-1], starts[1:])) > DENSE16_MAX_CHAIN:
        h2 = False
    return {"wide": wide, "g0": g0, "own_len": own_len, "h2": h2,
            "own_row0": [(C0 + g0[k] * F) * mult for k in range(L)],
            "key": (N, H, W, C0, L, F, tuple(segs0), preact, tuple(g0), h2)}


def _calibrate_backward_arg_refs():
    """sys.getrefcount of the gradient argument inside a custom Function's backward when nothing but the engine and the
    call hold it (a CPU mini-graph; the count is a property of the interpreter / torch build, 6 on Python 3.10 + torch 2.10)."""
    seen = []

    class _Probe(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x * 2

        @staticmethod
        def backward(ctx, dy):
            seen.append(sys.getrefcount(dy))
            return dy

    with torch.enable_grad():
        x = torch.zeros(2, requires_grad=True)
        (_Probe.apply(x) * 3).sum().backward()
    return seen[0]


_BACKWARD_ARG_REFS = _calibrate_backward_arg_refs()
# False (tests): convolutions ignore `grow`, dense blocks copy their input into their own buffer as before round 5
_GROW_IN_PLACE = True


class DenseBlockFunction(torch.autograd.Function):
    """L weight-normalised convolutions that each read the concatenation of everything before
    them and append `F` channels (reference models/densenet.py:11-16: `x.append(conv2d(x, F))`).

    The block grows IN PLACE: one [N,H,W,C0+L*F] buffer, layer k reads channels [0, Ck) through
    the channel map of its list segmentation and writes channels [Ck, Ck+F) (ldy / y_coff of
    the conv ABI).  The backward pass walks the layers in reverse over one gradient buffer:
    layer k's dy is a channel slice of it and its dgrad accumulates into channels [0, Ck).
    No concatenation copies, no per-layer activation tensors.

    Finished channel groups are taken out of the chain (`_split_block_plan`): every layer's pre-activation is linear
    in act(earlier channels), so the block input's share of ALL L layers is one 3x3 convolution C0 -> L*F written
    straight into channels [C0, Ctot) of the buffer, and the finished first half of the growth outputs enters the second
    half of the layers through one more (L/2)*F -> (L/2)*F convolution added onto them -- GEMMs wide enough for the
    Winograd F(4x4,3x3) passes on the split-precision engine (2.25 instead of 9 products per output); layer k then
    only adds the convolution of the growth outputs of its own group (`y_accumulate`).  Backward mirrors it: the
    growth layers' dgrad / wgrad see their own group only, each wide convolution takes one dgrad and one wgrad
    against the finished gradient of the channels it wrote.  Same sums in a different order; chosen when the
    library routes the wide convolutions to its Winograd path (OTGAN_DENSE_SPLIT=0: never, OTGAN_DENSE_GROUP=n:
    groups of n, 0: the block input only).

    args: x0 [N,H,W,C0] (concatenated initial list), then V_k, g_k, b_k for every layer."""

    @staticmethod
    def forward(ctx, x0, segs0, ksize, preact, *params):
        _need_cuda(x0)
        L = len(params) // 3
        N, H, W, C0 = x0.shape
        F = params[0].shape[-1]
        Ctot = C0 + L * F
        buf = grown_buffer(x0, Ctot, claim=True)      # the producing convolution left room (conv2d_op grow): no copy
        if buf is None:
            buf = torch.empty((N, H, W, Ctot), dtype=x0.dtype, device=x0.device)
            buf[..., :C0].copy_(x0)
        mult = 2 if preact in DOUBLED else 1
        saved, descs, maps, per_layer = [], [], [], []
        segs = list(segs0)
        V2ds = []
        for k in range(L):
            V = params[3 * k]
            assert tuple(V.shape) == (ksize, ksize, (C0 + k * F) * mult, F), (V.shape, C0 + k * F, mult)
            V2ds.append(V.contiguous().view(-1, F))
        gs = list(params[1::3])
        ctx.batched = F == 16 and ksize == 3 and _aligned16(*V2ds, *gs) and WN_BATCHED
        batch = []

        def normalised(k):
            # all layers of the block in one launch the first time any of them misses the cache
            if ctx.batched:
                if not batch:
                    batch.extend(weightnorm_fwd_block(V2ds, gs))
                return batch[k]
            return weightnorm_fwd(V2ds[k], gs[k])

        for k in range(L):
            per_layer.append(cached_weights(params[3 * k], gs[k], lambda k=k: normalised(k)))
            saved += [V2ds[k], gs[k], per_layer[k][0], per_layer[k][2]]
        ctx.vshapes = [p.shape for p in params[0::3]]
        ctx.L, ctx.C0, ctx.F = L, C0, F

        plan = None
        if ksize == 3 and all(p is not None for p in params[2::3]):
            plan = _split_block_plan(N, H, W, C0, L, F, segs0, preact, x0.device)
        ctx.plan = plan
        if plan is not None:
            sw = _split_block_weights(params[0::3], per_layer, plan, F)
            # the L biases side by side, kept with the block's operands: rebuilt with them after a weight update (the optimiser
            # kernels bump the storage epoch that invalidates `sw`), or when a torch op touched a bias (version counters)
            bkey = tuple((id(b), b._version) for b in params[2::3])
            if sw.get("bias_key") != bkey:
                sw["bias_all"], sw["bias_key"] = torch.cat([b.detach() for b in params[2::3]]), bkey
            bias_all = sw["bias_all"]
            rows = N * H * W
            ctx.x_recs, ctx.x_ops = [], []
            # amax records without extra passes (DENSE_AMAX False: one reduction per slice, as in round 2): every kernel
            # that writes growth channels -- the wide convolutions (the second one adds onto the first) and the 16-output
            # kernels of the chains -- leaves the largest magnitude of the values it writes in a record (layout below);
            # the block's output carries max(record of x0, all of them) for the transition that reads the whole buffer.
            shared = DENSE_AMAX and _FUSED_AMAX
            rec_x0 = amax_of(x0) if shared else None
            # round 4: ONE zeroed array of records per block.  Group i (the layers between two wide convolutions) owns
            # rows [gbase(i), gbase(i + 1)): first the record of the sums the group's wide convolution leaves in the
            # growth channels it writes (W_i: bounds every slice a layer of the group finishes without a chain kernel,
            # and the partial sums of the others), then one record per layer of the group, written by the chain kernel
            # that finishes the layer's slice.  A chain kernel reads rows [gbase, gbase + 1 + j) -- consecutive records,
            # otgan_conv_desc::x_amax_count -- as the bound of the slices it multiplies (the two-scaled-fp16-piece
            # kernel needs it; records are only ever read by launches after their writers: deterministic).
            wides = plan["wide"]
            gbase = [wd["d0"] + i for i, wd in enumerate(wides)] + [L + len(wides)]
            gidx = {wd["d0"]: i for i, wd in enumerate(wides)}
            R = amax_slots(buf.device, L + len(wides)) if shared else None

            def wide_fwd(i):
                wd, ops_ = plan["wide"][i], sw["wide"][i]
                desc = wd["desc"]
                src = buf[..., wd["x_off"]:]       # channel slice: same rows, pointer advanced by x_off floats
                nrec = 1
                if shared and i == 0 and rec_x0 is not None:
                    rec = rec_x0
                elif shared and i > 0:
                    # the finished slices of the group that feeds this convolution: its rows of R as they are
                    # (otgan_conv_desc::x_amax_count consecutive records; until round 4 an amax over them: one launch)
                    rec, nrec = R[gbase[i - 1]:gbase[i]], gbase[i] - gbase[i - 1]
                else:
                    rec = absmax_record_strided(src.data_ptr(), rows, wd["C"], Ctot, buf.device)
                ctx.x_recs.append((rec, nrec))
                desc.x_amax, desc.x_amax_count = rec.data_ptr(), nrec
                if any(ctx.needs_input_grad[4:]):
                    ctx.x_ops.append(shared_x_operand(desc, buf.device))    # read back by this convolution's wgrad
                desc.y_accumulate = wd["accumulate"]
                desc.y_amax_out = R[gbase[i]].data_ptr() if shared else None
                conv_fwd_raw(desc, src, None, ops_["wT"], None if wd["accumulate"] else bias_all, buf, ops_["fwd"])
                desc.y_accumulate = 0
                desc.y_amax_out = None
                desc.x_amax_count = 0

            wide_fwd(0)
            chain_h2 = shared and plan.get("h2") and all(sw["h2"][k] is not None for k in range(L) if plan["own_len"][k])
            if chain_h2:
                # every group's chain in one library call (otgan_dense16_chain_fwd_f32: its layers' launches back to back)
                lib = _lib.lib()
                for gi, wd in enumerate(wides):
                    d0, d1 = wd["d0"], (wides[gi + 1]["d0"] if gi + 1 < len(wides) else L)
                    if gi > 0:
                        wide_fwd(gi)        # (the group that feeds this one is finished)
                    if d1 - d0 >= 2:
                        pf = (ctypes.c_void_p * (d1 - d0 - 1))(*[sw["h2"][k].data_ptr() for k in range(d0 + 1, d1)])
                        _lib.check(lib.otgan_dense16_chain_fwd_f32(N, H, W, d1 - d0, buf.data_ptr() + 4 * (C0 + d0 * F), Ctot,
                                                                   ctypes.cast(pf, ctypes.c_void_p), R[gbase[gi]].data_ptr(),
                                                                   _lib.stream_ptr()), "dense16_chain_fwd")
                for k in range(L):      # descriptors / maps of the chain layers (their weight gradients use them)
                    n_own = plan["own_len"][k]
                    if n_own:
                        desc = ConvDesc(N, H, W, n_own * F, Ctot, 0, ksize, ksize, 1, F, Ctot, C0 + k * F, preact, 1)
                        desc.list_width = F
                        descs.append(desc)
                        maps.append(channel_maps((F,) * n_own, preact, x0.device))
                    else:
                        descs.append(None)
                        maps.append((None, None))
            for k in range(L) if not chain_h2 else ():
                n_own = plan["own_len"][k]
                desc = cmap = inv = None
                if n_own:
                    g0 = plan["g0"][k]
                    desc = ConvDesc(N, H, W, n_own * F, Ctot, 0, ksize, ksize, 1, F, Ctot, C0 + k * F, preact, 1)
                    desc.y_accumulate = 1
                    desc.list_width = F
                    if shared:
                        gi = gidx[g0]
                        desc.x_amax = R[gbase[gi]].data_ptr()
                        desc.x_amax_count = 1 + n_own
                        desc.y_amax_out = R[gbase[gi] + 1 + n_own].data_ptr()
                    cmap, inv = channel_maps((F,) * n_own, preact, x0.device)
                    conv_fwd_raw(desc, buf[..., C0 + g0 * F:], cmap, sw["wT_g"][k], None, buf, sw["h2"][k] if shared else None)
                    desc.y_accumulate = 0
                    desc.y_amax_out = None
                    desc.x_amax = None
                    desc.x_amax_count = 0
                descs.append(desc)
                maps.append((cmap, inv))
                for i, wd in enumerate(plan["wide"]):
                    if wd["after"] == k:       # the group that ends with layer k is finished
                        wide_fwd(i)
            ctx.sw = sw
            ctx.save_for_backward(buf, *saved)
            ctx.descs, ctx.maps = descs, maps
            ctx.fwd_recs = (R, gbase, gidx) if shared else None      # read again by the chains' weight gradients
            if shared:
                x0_rec = ctx.x_recs[0][0]     # the producer's record of x0, or the reduction wide_fwd(0) made
                tag_amax(buf, torch.maximum(x0_rec, R.amax(0)))
            return buf

        for k in range(L):
            b = params[3 * k + 2]
            Ck = C0 + k * F
            desc = ConvDesc(N, H, W, Ck, Ctot, 0, ksize, ksize, 1, F, Ctot, Ck, preact,
                            1 if all(s % 4 == 0 for s in segs) else 0)
            cmap, inv = channel_maps(tuple(segs), preact, x0.device)
            conv_fwd_raw(desc, buf, cmap, per_layer[k][1], b, buf)
            descs.append(desc)
            maps.append((cmap, inv))
            segs.append(F)
        ctx.save_for_backward(buf, *saved)
        ctx.descs, ctx.maps = descs, maps
        return buf

    @staticmethod
    def backward(ctx, dbuf):
        # gradient w.r.t. the whole concatenation; the earlier slices' gradients are accumulated into it.  The incoming
        # tensor itself is used when nothing else can see it -- contiguous and referenced only by the engine and this call
        # (a gradient that a torch op hands to two nodes at once, e.g. of `block_a + block_b`, has more references) --
        # instead of a copy of the whole buffer per block and pass (0.5 ms of a DenseNet step).
        # `_use_count` counts references to the C++ tensor, not Python references to its wrapper: a gradient that a tensor
        # hook (register_hook) or an upstream custom Function has kept alive shows up in sys.getrefcount only.
        # Both are checked, the second against the count an un-retained gradient has inside a custom backward in this
        # interpreter / torch build (calibrated at import): a retained gradient is copied, never overwritten.
        if (_GRAD_INPLACE and dbuf.is_contiguous() and hasattr(dbuf, "_use_count") and dbuf._use_count() <= 2 and
                sys.getrefcount(dbuf) <= _BACKWARD_ARG_REFS):
            G = dbuf
        else:
            G = dbuf.contiguous().clone()
        try:
            return DenseBlockFunction._backward_impl(ctx, dbuf, G)
        finally:
            # the raw kernels wrote into G without bumping its version counter: a record (or column sums) tagged onto the
            # incoming tensor would still pass the version check
            if G is dbuf:
                for tag in ("_otgan_amax", "_otgan_colsum"):
                    if hasattr(dbuf, tag):
                        delattr(dbuf, tag)

    @staticmethod
    def _backward_impl(ctx, dbuf, G):
        buf, *saved = ctx.saved_tensors
        L, C0, F = ctx.L, ctx.C0, ctx.F
        N, H, W, Ctot = buf.shape
        need_w = any(ctx.needs_input_grad[4:])
        grads = [None] * (3 * L)
        rows = N * H * W
        plan = ctx.plan
        if plan is not None and plan.get("h2") and ctx.batched and len(plan["wide"]) <= 2:
            grads = DenseBlockFunction._backward_by_slice(ctx, buf, saved, G, dbuf, need_w)
            dx0 = _block_input_grad(G, C0) if ctx.needs_input_grad[0] else None
            if dx0 is not None and _FUSED_AMAX and C0 % 4 == 0:
                tag_amax(dx0, ctx.dx0_rec)
            return (dx0, None, None, None, *grads)
        if plan is not None:
            sw = ctx.sw
            dw_g = [None] * L
            dw_wide = [None] * len(plan["wide"])

            def wide_bwd(i, need_dx):
                # channels [C0 + d0*F, Ctot) of G are final here: the group sees them through its wide convolution
                wd, ops_ = plan["wide"][i], sw["wide"][i]
                desc = wd["desc"]
                src, gsrc = buf[..., wd["x_off"]:], G[..., wd["x_off"]:]
                n_out = (L - wd["d0"]) * F
                dy_rec = absmax_record_strided(G.data_ptr() + 4 * (C0 + wd["d0"] * F), rows, n_out, Ctot, G.device)
                desc.x_amax, desc.x_amax_count = ctx.x_recs[i][0].data_ptr(), ctx.x_recs[i][1]
                desc.dy_amax = dy_rec.data_ptr()
                if need_w:
                    dw = torch.empty_like(ops_["w"])
                    conv_wgrad_raw(desc, src, None, G, dw)
                    dw_wide[i] = dw          # [9][nrows][layers d0 ..][F], rows in the convolution's own order
                if need_dx:
                    if not ops_["bwd_done"]:
                        ops_["bwd"], ops_["bwd_done"] = prepare_filters(desc, 1, ops_["w"]), True
                    conv_dgrad_raw(desc, G, ops_["w"], src, None, gsrc, Ctot, True, ops_["bwd"])
                desc.dy_amax = None
                desc.x_amax_count = 0

            for k in reversed(range(L)):
                for i in reversed(range(1, len(plan["wide"]))):
                    if plan["wide"][i]["after"] == k:
                        wide_bwd(i, True)
                if plan["own_len"][k]:
                    desc = ctx.descs[k]
                    cmap, inv = ctx.maps[k]
                    off = C0 + plan["g0"][k] * F
                    if need_w:
                        dw_g[k] = torch.empty_like(sw["w_g"][k])
                        conv_wgrad_raw(desc, buf[..., off:], cmap, G, dw_g[k])
                    # d/d(growth outputs of the layer's own half) accumulates into their channels of G
                    conv_dgrad_raw(desc, G, sw["w_g"][k], buf[..., off:], inv, G[..., off:], Ctot, True)
            wide_bwd(0, ctx.needs_input_grad[0])
            if need_w and ctx.batched and len(plan["wide"]) <= 2:
                # the pieces of every layer's weight gradient where the passes left them (a column slice of each wide
                # convolution's dw, in that convolution's row order, then the layer's own chain): one launch
                parts = []
                for k in range(L):
                    pk = []
                    for i, wd in enumerate(plan["wide"]):
                        if wd["d0"] <= k:
                            nl = L - wd["d0"]
                            pk.append((dw_wide[i].data_ptr() + 4 * (k - wd["d0"]) * F, wd["back"], wd["nrows"], nl * F))
                    if dw_g[k] is not None:
                        pk.append((dw_g[k].data_ptr(), None, dw_g[k].shape[0] // 9, F))
                    parts.append(pk)
                dVs, dgs = weightnorm_bwd_block(saved[0::4], saved[1::4], saved[3::4], parts)
                for k in range(L):
                    grads[3 * k:3 * k + 2] = [dVs[k].view(ctx.vshapes[k]), dgs[k]]
            elif need_w:
                for k in range(L):
                    V2d, g, w, inv_norm = saved[4 * k:4 * k + 4]
                    parts = []
                    for i, wd in enumerate(plan["wide"]):
                        if wd["d0"] <= k:
                            dwi = dw_wide[i].view(9, wd["nrows"], L - wd["d0"], F)
                            if wd["back"] is not None:
                                dwi = dwi.index_select(1, wd["back"].long())
                            parts.append(dwi[:, :, k - wd["d0"], :])
                    if dw_g[k] is not None:
                        parts.append(dw_g[k].view(9, -1, F))
                    dw = torch.cat(parts, dim=1) if len(parts) > 1 else parts[0].contiguous()
                    dV2d, dg = weightnorm_bwd(V2d, g, inv_norm, dw.view(-1, F))
                    grads[3 * k:3 * k + 2] = [dV2d.view(ctx.vshapes[k]), dg]
        else:
            dws = [None] * L
            for k in reversed(range(L)):
                V2d, g, w, inv_norm = saved[4 * k:4 * k + 4]
                desc = ctx.descs[k]
                cmap, inv = ctx.maps[k]
                if need_w:
                    dws[k] = torch.empty_like(V2d)
                    conv_wgrad_raw(desc, buf, cmap, G, dws[k])
                    if not ctx.batched:
                        dV2d, dg = weightnorm_bwd(V2d, g, inv_norm, dws[k])
                        grads[3 * k:3 * k + 2] = [dV2d.view(ctx.vshapes[k]), dg]
                # d/d(inputs of layer k) accumulates into the first Ck channels of G
                conv_dgrad_raw(desc, G, w, buf, inv, G, Ctot, True)
            if need_w and ctx.batched:
                parts = [[(dws[k].data_ptr(), None, dws[k].shape[0] // 9, F)] for k in range(L)]
                dVs, dgs = weightnorm_bwd_block(saved[0::4], saved[1::4], saved[3::4], parts)
                for k in range(L):
                    grads[3 * k:3 * k + 2] = [dVs[k].view(ctx.vshapes[k]), dgs[k]]
        if need_w:
            # Layer k's output gradient G[..., Ck:Ck+F] is final once the layers after it have been
            # processed, and no earlier layer writes there: all L bias gradients are the column sums
            # of the finished G -- one reduction instead of L.
            db_all = colsum(G.data_ptr() + 4 * C0, rows, L * F, Ctot, G.device)
            for k in range(L):
                grads[3 * k + 2] = db_all[k * F:(k + 1) * F]
        dx0 = _block_input_grad(G, C0) if ctx.needs_input_grad[0] else None
        return (dx0, None, None, None, *grads)


def _block_input_grad(G, C0):
    """Gradient of a dense block's input = channels [0, C0) of its finished gradient buffer: handed on as a VIEW when the
    kernels of the layer in front can read it through a channel stride (Conv2dFunction.backward: channel_prefix_stride),
    else as a copy (round 5: one copy of the block input's gradient per block and pass less)."""
    v = G[..., :C0]
    return v if channel_prefix_stride(v) is not None else v.contiguous()


def _dense16_bwd_filters(sw, plan, L, F, device):
    """Prepared weights of the by-slice input gradient (otgan_dense16_prepare_bwd_filters_f32): per output slice c one
    buffer holding the pairs (c, k), k = the later layers of c's group.  Cached with the block's forward operands."""
    hit = sw.get("h2_bwd")
    if hit is not None:
        return hit
    from ._lib_layers import Dense16BwdPair
    lib = _lib.lib()
    starts = [wd["d0"] for wd in plan["wide"]] + [L]
    gend = {}
    for a, b in zip(starts[:-1], starts[1:]):
        for c in range(a, b):
            gend[c] = b
    nsl = {c: gend[c] - 1 - c for c in range(L) if gend[c] - 1 - c >= 1}
    sizes = {c: int(lib.otgan_dense16_bwd_filter_bytes(n)) for c, n in nsl.items()}
    flat = torch.empty(sum(sizes.values()), dtype=torch.uint8, device=device)
    out, off, pairs = {}, 0, []
    for c in sorted(nsl):
        out[c] = flat[off:off + sizes[c]]
        off += sizes[c]
        for j in range(nsl[c]):
            k = c + 1 + j
            pairs.append(Dense16BwdPair(sw["w_g"][k].data_ptr(), sw["h2"][k].data_ptr(), out[c].data_ptr(),
                                        plan["own_len"][k], c - plan["g0"][k], j))
    allf = [sw["h2"][k] for k in range(L) if sw["h2"][k] is not None]
    if pairs:
        arr = (Dense16BwdPair * len(pairs))(*pairs)
        pf = (ctypes.c_void_p * len(allf))(*[t.data_ptr() for t in allf])
        _lib.check(lib.otgan_dense16_prepare_bwd_filters_f32(ctypes.cast(arr, ctypes.c_void_p), len(pairs),
                                                              ctypes.cast(pf, ctypes.c_void_p), len(allf), _lib.stream_ptr()),
                   "dense16_prepare_bwd_filters")
    res = {"flat": flat, "slice": out, "nsl": nsl}
    sw["h2_bwd"] = res
    return res


def _backward_by_slice(ctx, buf, saved, G, dbuf, need_w):
    """Backward of a split dense block whose chains run on the two-scaled-fp16-piece kernels (plan["h2"]): the input
    gradient of the chains is gathered per SLICE, last slice first (otgan_dense16_bwd_slice_f32: one read-modify-write of
    each slice instead of one per (layer, earlier slice) pair), and every kernel that adds into the gradient buffer leaves
    the largest magnitude it wrote in a record -- no reduction passes over slices of the buffer (round 3: two per block).
    Returns the list of parameter gradients (grads[3 k + 2], the biases, included)."""
    L, C0, F = ctx.L, ctx.C0, ctx.F
    N, H, W, Ctot = buf.shape
    plan, sw = ctx.plan, ctx.sw
    rows = N * H * W
    lib = _lib.lib()
    grads = [None] * (3 * L)
    dw_g = [None] * L
    wides = plan["wide"]
    dw_wide = [None] * len(wides)
    bw = _dense16_bwd_filters(sw, plan, L, F, buf.device)
    # records: Rc bounds the incoming gradient and everything the wide convolutions' input gradients add; RS[c] what the
    # slice kernel of slice c leaves.  A reader takes the maximum of Rc and the rows of the slices it reads.
    # One array: rows [0, L) = RS, row L = Rc -- a reader of the slices from d0 on passes rows [d0, L] as they are
    # (otgan_conv_desc::dy_amax_count consecutive records; until round 4 an amax over the rows and a maximum with Rc: two launches).
    tag = amax_of(dbuf)
    RR = amax_slots(buf.device, L + 1)
    RS, Rc = RR[:L], RR[L]
    Rc.copy_(tag if tag is not None else absmax_record(G))
    R0 = amax_slot(buf.device)
    ctx.dx0_rec = R0
    gptr, bptr = G.data_ptr(), buf.data_ptr()
    # weight-gradient work on the side stream (SIDE_STREAM, see Conv2dFunction.backward): every weight gradient of the block
    # reads slices of G that are FINAL when it is issued and writes its own tensors, so it can run under the input-gradient
    # kernels that follow on this stream (they add into OTHER channels of G)
    side = SIDE_STREAM if need_w else None
    if side is not None:
        for t in [G, buf, RR, R0] + [t for t in ctx.x_ops if t is not None] + ([ctx.fwd_recs[0]] if ctx.fwd_recs is not None else []):
            t.record_stream(side)

    def wgrad_stream():
        if side is None:
            return _NULL_CTX
        side.wait_stream(torch.cuda.current_stream())      # (what the weight gradient reads is enqueued on this stream)
        return torch.cuda.stream(side)

    def wide_bwd(i, need_dx):
        wd, ops_ = wides[i], sw["wide"][i]
        desc = wd["desc"]
        src, gsrc = buf[..., wd["x_off"]:], G[..., wd["x_off"]:]
        desc.x_amax, desc.x_amax_count = ctx.x_recs[i][0].data_ptr(), ctx.x_recs[i][1]
        desc.dy_amax, desc.dy_amax_count = RR[wd["d0"]].data_ptr(), L + 1 - wd["d0"]
        if need_w:
            with wgrad_stream():
                dw = torch.empty_like(ops_["w"])
                conv_wgrad_raw(desc, src, None, G, dw)
            dw_wide[i] = dw
        if need_dx:
            if not ops_["bwd_done"]:
                ops_["bwd"], ops_["bwd_done"] = prepare_filters(desc, 1, ops_["w"]), True
            # i = 0 writes the block input's gradient (sums onto the incoming one): its own record, handed to the layer in
            # front of the block with dx0 (round 4: that layer reduced the tensor itself)
            desc.dx_amax_out = Rc.data_ptr() if i > 0 else R0.data_ptr()
            conv_dgrad_raw(desc, G, ops_["w"], src, None, gsrc, Ctot, True, ops_["bwd"])
            desc.dx_amax_out = None
        desc.dy_amax = None
        desc.dy_amax_count = desc.x_amax_count = 0

    starts = [wd["d0"] for wd in wides] + [L]
    for gi in reversed(range(len(wides))):
        d0, d1 = starts[gi], starts[gi + 1]
        if gi + 1 < len(wides):
            wide_bwd(gi + 1, True)       # the later groups are final: their share in this group's slices
        if d1 - d0 >= 2:
            # slices d1 - 2 .. d0, last first, in one library call (otgan_dense16_chain_bwd_f32)
            pf = (ctypes.c_void_p * (d1 - d0 - 1))(*[bw["slice"][c].data_ptr() for c in range(d0, d1 - 1)])
            off0 = 4 * (C0 + d0 * F)
            _lib.check(lib.otgan_dense16_chain_bwd_f32(N, H, W, d1 - d0, gptr + off0, Ctot, bptr + off0, Ctot,
                                                       ctypes.cast(pf, ctypes.c_void_p), Rc.data_ptr(), RS[d0].data_ptr(),
                                                       _lib.stream_ptr()), "dense16_chain_bwd")
        # the gradients of the group's slices are final: the chain weight gradients of its layers
        for c in reversed(range(d0, d1)):
            if need_w and plan["own_len"][c]:
                desc = ctx.descs[c]
                cmap, _inv = ctx.maps[c]
                off = C0 + plan["g0"][c] * F
                with wgrad_stream():
                    dw_g[c] = torch.empty_like(sw["w_g"][c])
                if ctx.fwd_recs is not None:
                    # records of the slices the layer read (the rows its forward kernel read) and of its output gradient (the
                    # slices' rows from c on and the incoming bound): the kernel runs on the fp16 matrix pipe (round 4)
                    R, gbase, gidx = ctx.fwd_recs
                    desc.x_amax = R[gbase[gidx[plan["g0"][c]]]].data_ptr()
                    desc.x_amax_count = 1 + plan["own_len"][c]
                    desc.dy_amax, desc.dy_amax_count = RR[c].data_ptr(), L + 1 - c
                with wgrad_stream():
                    conv_wgrad_raw(desc, buf[..., off:], cmap, G, dw_g[c])
                desc.x_amax = desc.dy_amax = None
                desc.x_amax_count = desc.dy_amax_count = 0
    wide_bwd(0, ctx.needs_input_grad[0])
    if need_w:
        assert ctx.batched and len(wides) <= 2
        parts = []
        for k in range(L):
            pk = []
            for i, wd in enumerate(wides):
                if wd["d0"] <= k:
                    nl = L - wd["d0"]
                    pk.append((dw_wide[i].data_ptr() + 4 * (k - wd["d0"]) * F, wd["back"], wd["nrows"], nl * F))
            if dw_g[k] is not None:
                pk.append((dw_g[k].data_ptr(), None, dw_g[k].shape[0] // 9, F))
            parts.append(pk)
        with wgrad_stream():
            dVs, dgs = weightnorm_bwd_block(saved[0::4], saved[1::4], saved[3::4], parts)
            for k in range(L):
                grads[3 * k:3 * k + 2] = [dVs[k].view(ctx.vshapes[k]), dgs[k]]
            db_all = colsum(G.data_ptr() + 4 * C0, rows, L * F, Ctot, G.device)
            for k in range(L):
                grads[3 * k + 2] = db_all[k * F:(k + 1) * F]
    return grads


DenseBlockFunction._backward_by_slice = staticmethod(_backward_by_slice)


def dense_block_op(x0, segs0, params, ksize=3, preact=1):
    flat = [t for p in params for t in p]
    return DenseBlockFunction.apply(x0, tuple(segs0), int(ksize), int(preact), *flat)


# ------------------------------------------------------------------------------- pointwise
class GluFunction(torch.autograd.Function):
    """x: [..., 2C] -> x[..., :C] * sigmoid(x[..., C:])   (models/dcgan.py:35-36)."""

    @staticmethod
    def forward(ctx, x):
        _need_cuda(x)
        x = x.contiguous()
        C2 = x.shape[-1]
        rows = x.numel() // C2
        pre = getattr(x, "_otgan_glu", None)
        if pre is not None and pre[1] == x._version:
            # the layer that produced x already wrote the gated product (Conv2dFunction, glu_hint)
            del x._otgan_glu
            _PENDING_GLU[0] = None
            ctx.save_for_backward(x)
            return pre[0]
        y = torch.empty(x.shape[:-1] + (C2 // 2,), dtype=x.dtype, device=x.device)
        rec = amax_slot(x.device) if (_FUSED_AMAX and (C2 // 2) % 4 == 0) else None
        _lib.check(_lib.lib().otgan_glu_fwd_amax_f32(x.data_ptr(), rows, C2 // 2, y.data_ptr(), _lib.ptr(rec),
                                                     _lib.stream_ptr()), "glu_fwd")
        if rec is not None:
            tag_amax(y, rec)
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = dy.contiguous()
        C2 = x.shape[-1]
        dx = torch.empty_like(x)
        rec = amax_slot(x.device) if (_FUSED_AMAX and (C2 // 2) % 4 == 0) else None
        if _GLU_COLSUM and (C2 // 2) % 4 == 0:
            # dx is the output gradient of the convolution in front of the GLU: leave its column sums (that layer's
            # bias gradient) with it instead of reading dx once more there
            cs = torch.empty(C2, dtype=x.dtype, device=x.device)
            scratch = torch.empty(256 * C2, dtype=x.dtype, device=x.device)
            _lib.check(_lib.lib().otgan_glu_bwd_colsum_f32(x.data_ptr(), dy.data_ptr(), x.numel() // C2, C2 // 2,
                                                           dx.data_ptr(), _lib.ptr(rec), cs.data_ptr(), scratch.data_ptr(),
                                                           _lib.stream_ptr()), "glu_bwd_colsum")
            dx._otgan_colsum = (cs, dx._version)
        else:
            _lib.check(_lib.lib().otgan_glu_bwd_amax_f32(x.data_ptr(), dy.data_ptr(), x.numel() // C2,
                                                         C2 // 2, dx.data_ptr(), _lib.ptr(rec), _lib.stream_ptr()), "glu_bwd")
        if rec is not None:
            tag_amax(dx, rec)
        return dx


class TanhFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _need_cuda(x)
        x = x.contiguous()
        y = torch.empty_like(x)
        _lib.check(_lib.lib().otgan_tanh_fwd_f32(x.data_ptr(), x.numel(), y.data_ptr(),
                                                 _lib.stream_ptr()), "tanh_fwd")
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(y)
        _lib.check(_lib.lib().otgan_tanh_bwd_f32(y.data_ptr(), dy.data_ptr(), y.numel(),
                                                 dx.data_ptr(), _lib.stream_ptr()), "tanh_bwd")
        return dx


class FeatureHeadFunction(torch.autograd.Function):
    """[N,H,W,C] -> [N, H*W*2C]: concat([relu(x), relu(-x)], channel), flatten, L2-normalise
    (models/dcgan.py:16-19, models/densenet.py:37-42; no epsilon)."""

    @staticmethod
    def forward(ctx, x):
        _need_cuda(x)
        x = x.contiguous()
        _trace_signs(x)
        N, H, W, C = x.shape
        f = torch.empty((N, H * W * 2 * C), dtype=x.dtype, device=x.device)
        norm = torch.empty(N, dtype=x.dtype, device=x.device)
        _lib.check(_lib.lib().otgan_feature_head_fwd_f32(x.data_ptr(), N, H * W, C, f.data_ptr(),
                                                         norm.data_ptr(), _lib.stream_ptr()), "head_fwd")
        ctx.save_for_backward(x, f, norm)
        return f

    @staticmethod
    def backward(ctx, df):
        x, f, norm = ctx.saved_tensors
        df = df.contiguous()
        N, H, W, C = x.shape
        dx = torch.empty_like(x)
        rec = amax_slot(x.device) if (_FUSED_AMAX and C % 4 == 0) else None
        _lib.check(_lib.lib().otgan_feature_head_bwd_amax_f32(x.data_ptr(), f.data_ptr(), norm.data_ptr(),
                                                              df.data_ptr(), N, H * W, C, dx.data_ptr(), _lib.ptr(rec),
                                                              _lib.stream_ptr()), "head_bwd")
        if rec is not None:
            tag_amax(dx, rec)
        return dx


class ConcatChannelsFunction(torch.autograd.Function):
    """torch.cat(xs, dim = 3) of NHWC tensors that carries amax records both ways (round 4): the result is tagged with the
    maximum of its elements' records (elements without one -- the 16-channel noise inputs of the DenseNet generator,
    models/densenet.py:60-73 -- are reduced, they are small), and every piece of the gradient with the gradient's own
    record: a bound of a channel subset is all a consumer's power-of-two scale needs.  Without this the dense block
    behind the cat reduced its block input, and the layer in front of it its output gradient, once per step each."""

    @staticmethod
    def forward(ctx, *xs):
        ctx.widths = [int(t.shape[-1]) for t in xs]
        y = torch.cat(xs, 3)
        if _FUSED_AMAX and all(w % 4 == 0 for w in ctx.widths) and any(amax_of(t) is not None for t in xs):
            recs = [amax_of(t) if amax_of(t) is not None else absmax_record(t.contiguous()) for t in xs]
            rec = recs[0]
            for r in recs[1:]:
                rec = torch.maximum(rec, r)
            tag_amax(y, rec)
        return y

    @staticmethod
    def backward(ctx, dy):
        rec = amax_of(dy)
        outs = []
        for piece in dy.split(ctx.widths, 3):
            piece = piece.contiguous()
            if rec is not None:
                tag_amax(piece, rec)
            outs.append(piece)
        return tuple(outs)


def concat_channels(xs):
    return ConcatChannelsFunction.apply(*xs)


class ExtendChannelsFunction(torch.autograd.Function):
    """[x, *others] concatenated along the channels IN the buffer a convolution with `grow` allocated around x (round 5):
    the other elements (the 16-channel noise inputs of the DenseNet generator, models/densenet.py:60-73) are copied behind
    x, the result is the channel prefix [0, C0) of that buffer -- x itself is not copied, and its gradient is handed back
    as a view of the incoming one.  Records as in ConcatChannelsFunction."""

    @staticmethod
    def forward(ctx, Ctot, x, *others):
        base = grown_buffer(x, Ctot)
        assert base is not None
        ctx.widths = [int(t.shape[-1]) for t in (x,) + others]
        # the records are read BEFORE the copies: x is a view of `base` and shares its version counter, so the in-place
        # copies below make x's tag look stale (the record was never carried and the block reduced its input again)
        rec = amax_of(x) if (_FUSED_AMAX and all(w % 4 == 0 for w in ctx.widths)) else None
        if rec is not None:
            for t in others:
                r = amax_of(t)
                rec = torch.maximum(rec, r if r is not None else absmax_record(t.contiguous()))
        off = ctx.widths[0]
        for t in others:
            base[..., off:off + t.shape[-1]].copy_(t)
            off += t.shape[-1]
        y = base[..., :off]
        if rec is not None:
            tag_amax(y, rec)
        return y

    @staticmethod
    def backward(ctx, dy):
        rec = amax_of(dy)
        outs, off = [], 0
        for i, w in enumerate(ctx.widths):
            piece = dy[..., off:off + w]
            off += w
            if not ctx.needs_input_grad[1 + i]:
                outs.append(None)
                continue
            if i > 0 or channel_prefix_stride(piece) is None:
                piece = piece.contiguous()
            if rec is not None:
                tag_amax(piece, rec)
            outs.append(piece)
        return (None, *outs)


def extend_channels(xs, Ctot):
    """concat_channels(xs) inside the grown buffer of xs[0] when there is one with Ctot channels, else None."""
    if grown_buffer(xs[0], Ctot) is None or any(t.shape[:3] != xs[0].shape[:3] for t in xs[1:]):
        return None
    return ExtendChannelsFunction.apply(int(Ctot), *xs)


glu = GluFunction.apply
tanh = TanhFunction.apply
feature_head = FeatureHeadFunction.apply


# ------------------------------------------------------------------------------- optimiser steps
def adam_step(p, grad, v, mg, lr, mom1, mom2, t, coef=None):
    bump_weights_epoch(p)
    _lib.check(_lib.lib().otgan_adam_step_coef_f32(p.data_ptr(), grad.data_ptr(), _lib.ptr(v), mg.data_ptr(),
                                                   p.numel(), float(lr), float(mom1), float(mom2),
                                                   float(t), _lib.ptr(coef), _lib.stream_ptr()), "adam_step")


ADAM_MAX_SEGMENTS = 32      # otgan_layers.h: OTGAN_ADAM_MAX_SEGMENTS


def adam_coefficients(mom1, mom2, t):
    """(1 - mom1^t, 1 - mom2^t) as the library's Adam entries evaluate them (fp32; host only, no launch)."""
    out = (ctypes.c_float * 2)()
    _lib.lib().otgan_adam_coefficients(float(mom1), float(mom2), float(t), ctypes.cast(out, ctypes.c_void_p))
    return float(out[0]), float(out[1])


def adam_step_gather(p_flat, grads, offsets, v, mg, lr, mom1, mom2, t, ema_shadow=None, ema_decay=0.0, coef=None):
    """Adam on a flat parameter buffer from per-variable gradient tensors (no concatenation); optionally the EMA of the
    updated parameters in the same launch (otgan_adam_step_gather_f32).  coef: a device tensor [2] holding the step's
    bias corrections (adam_coefficients) -- read by the kernel instead of being derived from `t` on the host, so that a
    captured launch can be replayed for later steps (trainer.GraphedSteps)."""
    n = len(grads)
    bump_weights_epoch(p_flat)
    if ema_shadow is not None:
        bump_weights_epoch(ema_shadow)
    gp = (ctypes.c_void_p * n)(*[g.data_ptr() for g in grads])
    off = (ctypes.c_long * (n + 1))(*offsets)
    _lib.check(_lib.lib().otgan_adam_step_gather_coef_f32(p_flat.data_ptr(), ctypes.cast(gp, ctypes.c_void_p),
                                                          ctypes.cast(off, ctypes.c_void_p), n, _lib.ptr(v), mg.data_ptr(),
                                                          float(lr), float(mom1), float(mom2), float(t), _lib.ptr(coef),
                                                          _lib.ptr(ema_shadow), float(ema_decay), _lib.stream_ptr()),
               "adam_step_gather")


COPY2D_MAX_SEGMENTS = 64   # include/otgan_layers.h


def copy2d_batched(segs):
    """Strided 2-D copies in one launch (otgan_copy2d_batched_f32): segs = [(src_ptr, dst_ptr, rows, cols, src_ld, dst_ld)],
    element units, at most COPY2D_MAX_SEGMENTS per launch (longer lists take several)."""
    for i0 in range(0, len(segs), COPY2D_MAX_SEGMENTS):
        part = segs[i0:i0 + COPY2D_MAX_SEGMENTS]
        n = len(part)
        src = (ctypes.c_void_p * n)(*[q[0] for q in part])
        dst = (ctypes.c_void_p * n)(*[q[1] for q in part])
        rows = (ctypes.c_int * n)(*[q[2] for q in part])
        cols = (ctypes.c_int * n)(*[q[3] for q in part])
        sld = (ctypes.c_long * n)(*[q[4] for q in part])
        dld = (ctypes.c_long * n)(*[q[5] for q in part])
        cast = lambda a: ctypes.cast(a, ctypes.c_void_p)
        _lib.check(_lib.lib().otgan_copy2d_batched_f32(cast(src), cast(dst), cast(rows), cast(cols), cast(sld), cast(dld), n,
                                                       _lib.stream_ptr()), "copy2d_batched")


GATHER3D_MAX_SEGMENTS = 32   # include/otgan_layers.h


def gather3d_batched(segs, n1, n2, base1=0, map1=None, amax_out=None):
    """Strided 3-D copies with a gathered middle index in one launch (otgan_gather3d_batched_f32): segs = [(src_ptr, dst_ptr, n0,
    src_stride0, src_stride1, dst_stride0, dst_stride1)], element units; map1: int32 device tensor of n1 indices or None."""
    assert map1 is None or (map1.dtype == torch.int32 and map1.numel() == n1 and map1.is_contiguous())
    assert amax_out is None or len(segs) <= GATHER3D_MAX_SEGMENTS
    for i0 in range(0, len(segs), GATHER3D_MAX_SEGMENTS):
        part = segs[i0:i0 + GATHER3D_MAX_SEGMENTS]
        n = len(part)
        cast = lambda a: ctypes.cast(a, ctypes.c_void_p)
        arr = lambda ty, i: cast((ty * n)(*[q[i] for q in part]))
        _lib.check(_lib.lib().otgan_gather3d_batched_f32(arr(ctypes.c_void_p, 0), arr(ctypes.c_void_p, 1), arr(ctypes.c_int, 2),
                                                         int(n1), int(n2), arr(ctypes.c_long, 3), arr(ctypes.c_long, 4),
                                                         arr(ctypes.c_long, 5), arr(ctypes.c_long, 6), int(base1),
                                                         _lib.ptr(map1), _lib.ptr(amax_out), n, _lib.stream_ptr()),
                   "gather3d_batched")


def adamax_step(p, grad, v, mg, lr, mom1, mom2):
    bump_weights_epoch(p)
    _lib.check(_lib.lib().otgan_adamax_step_f32(p.data_ptr(), grad.data_ptr(), _lib.ptr(v),
                                                mg.data_ptr(), p.numel(), float(lr), float(mom1),
                                                float(mom2), _lib.stream_ptr()), "adamax_step")


def nesterov_step(p, grad, v, lr, mom1):
    bump_weights_epoch(p)
    _lib.check(_lib.lib().otgan_nesterov_step_f32(p.data_ptr(), grad.data_ptr(), v.data_ptr(),
                                                  p.numel(), float(lr), float(mom1),
                                                  _lib.stream_ptr()), "nesterov_step")


def ema_update(shadow, p, decay):
    bump_weights_epoch(shadow)
    _lib.check(_lib.lib().otgan_ema_update_f32(shadow.data_ptr(), p.data_ptr(), p.numel(),
                                               float(decay), _lib.stream_ptr()), "ema_update")
