"""Mini-batch Sinkhorn energy distance on MI355X -- drop-in for the reference's
`utils/matching.py` (same function names, argument meaning, return structure).

    get_matched_features(features_a, features_b, sinkhorn_lambda, nr_sinkhorn_iter)
        reference utils/matching.py:11-85   -> otgan_matching_two_batch_f32
    get_matched_features_single_batch(...)
        reference utils/matching.py:88-136  -> otgan_matching_single_batch_f32
    get_matched_features_random(features_a, features_b)
        reference utils/matching.py:3-9     (list rotation only; no kernel)
    calc_distance(features_a, features_b, matched_features)
        reference utils/matching.py:139-153 -> otgan_calc_distance_f32
    matched_feature_grads(fa, fb, ...)      (added) the differences the training step injects as grad_ys
        reference train.py:111,125-126      -> otgan_matching_two_batch_grad_f32

`features_a` / `features_b` are lists of S equally-shaped `[B, D]` float32 CUDA tensors
(`a` = generated, `b` = data; shards [0,S/2) form mini-batch 1, [S/2,S) mini-batch 2).
The return value is `(features_a_a, features_b_b, features_a_b, features_b_a, entropy)`
with four lists of S `[B, D]` tensors and a 0-d tensor.  Gradients do not flow through
the matching (reference train.py:111-128 injects the matched differences as `grad_ys`),
so everything here is computed under no_grad and returned detached.

All arithmetic runs in the HIP library; there is no fallback.
"""
import torch

from .. import _lib

COST_COSINE = 0
COST_SQEUCLID_MEAN = 1
_MODE_TWO, _MODE_SINGLE = 0, 1

_ws_cache = {}


def _workspace(nbytes, device):
    # one grow-only buffer per (device, stream): calls on one stream are ordered, calls on different
    # streams (two trainers in one process) must not share scratch
    key = (device.index if device.index is not None else torch.cuda.current_device(),
           torch.cuda.current_stream(device).cuda_stream)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def _check_lists(features_a, features_b):
    if len(features_a) != len(features_b) or len(features_a) == 0:
        raise ValueError("features_a and features_b must be non-empty lists of equal length")
    ref = features_a[0]
    for t in list(features_a) + list(features_b):
        if not t.is_cuda:
            raise _lib.OtganError("matching needs CUDA (MI355X) tensors; there is no CPU fallback")
        if t.dtype != torch.float32 or t.dim() != 2 or t.shape != ref.shape:
            raise ValueError("all shards must be float32 [B, D] tensors of one shape")


def _stack(shards):
    with torch.no_grad():
        return torch.cat([s.detach() for s in shards], 0).contiguous()


class MatchedFeatures(tuple):
    """The reference's 5-tuple, carrying a few extras as attributes (flat outputs, the
    device-side fp64 distance and per-problem statistics) so that calc_distance and the
    training step need not recompute them."""
    flat = None
    distance = None
    stats = None


def _run(mode, fa, fb, S, lam, iters, cost_kind):
    L = _lib.lib()
    rows_total, D = fa.shape
    rows = rows_total // 2 if mode == _MODE_TWO else rows_total
    dev = fa.device
    outs = [torch.empty_like(fa) for _ in range(4)]
    entropy = torch.empty((), dtype=torch.float32, device=dev)
    dist = torch.empty((), dtype=torch.float64, device=dev)
    nprob = 6 if mode == _MODE_TWO else 3
    stats = torch.empty((nprob, 4), dtype=torch.float64, device=dev)
    need = L.otgan_matching_workspace_bytes(mode, rows, D)
    ws = _workspace(need, dev)
    s = _lib.stream_ptr()
    if mode == _MODE_TWO:
        rc = L.otgan_matching_two_batch_f32(fa.data_ptr(), fb.data_ptr(), rows, D, D, float(lam),
                                            int(iters), int(cost_kind), outs[0].data_ptr(),
                                            outs[1].data_ptr(), outs[2].data_ptr(),
                                            outs[3].data_ptr(), D, entropy.data_ptr(),
                                            dist.data_ptr(), stats.data_ptr(), ws.data_ptr(),
                                            ws.numel(), s)
        _lib.check(rc, "otgan_matching_two_batch_f32")
    else:
        rc = L.otgan_matching_single_batch_f32(fa.data_ptr(), fb.data_ptr(), rows, D, D, float(lam),
                                               int(iters), outs[0].data_ptr(), outs[1].data_ptr(),
                                               outs[2].data_ptr(), outs[3].data_ptr(), D,
                                               entropy.data_ptr(), dist.data_ptr(),
                                               stats.data_ptr(), ws.data_ptr(), ws.numel(), s)
        _lib.check(rc, "otgan_matching_single_batch_f32")
    lists = [list(torch.chunk(o, S, 0)) for o in outs]
    res = MatchedFeatures((*lists, entropy))
    res.flat = outs
    res.distance = dist
    res.stats = stats
    return res


def get_matched_features(features_a, features_b, sinkhorn_lambda, nr_sinkhorn_iter):
    """Two-batch matching (reference utils/matching.py:11-85)."""
    _check_lists(features_a, features_b)
    S = len(features_a)
    if S % 2 != 0:
        raise ValueError("the two-batch matching needs an even number of shards (train.py:34)")
    return _run(_MODE_TWO, _stack(features_a), _stack(features_b), S, sinkhorn_lambda,
                nr_sinkhorn_iter, COST_COSINE)


def cost_log_kernel(x, y, sinkhorn_lambda, diag_add=0.0, cost_kind=COST_COSINE):
    """K[n,m] = -lambda * (cost(x[n,D], y[m,D]) + diag_add*I)  (matching.py:31,50) -- staged
    entry point, used by data-parallel ranks to compute their own row slices."""
    L = _lib.lib()
    x, y = x.detach().contiguous(), y.detach().contiguous()
    n, D = x.shape
    m = y.shape[0]
    K = torch.empty((n, m), dtype=x.dtype, device=x.device)
    need = L.otgan_cost_matrix_workspace_bytes(n, m, D)
    ws = torch.empty(int(need), dtype=torch.uint8, device=x.device)
    rc = L.otgan_cost_matrix_f32(x.data_ptr(), y.data_ptr(), n, m, D, D, float(sinkhorn_lambda),
                                 int(cost_kind), float(diag_add), K.data_ptr(), ws.data_ptr(), int(need),
                                 _lib.stream_ptr())
    _lib.check(rc, "otgan_cost_matrix_f32")
    return K


def cost_log_kernels(xs, ys, sinkhorn_lambda, cost_kind=COST_COSINE):
    """K[p] = -lambda * cost(xs[p][n,D], ys[p][m,D]) for up to six equally shaped pairs in ONE launch
    (otgan_cost_matrix_batched_f32) -> [P, n, m].  A tensor that appears several times in xs / ys is staged once:
    a data-parallel rank passes its own rows as every x and the gathered halves as the ys (matching.py:29-39)."""
    import ctypes
    L = _lib.lib()
    xs = [x.detach().contiguous() for x in xs]
    ys = [y.detach().contiguous() for y in ys]
    P = len(xs)
    n, D = xs[0].shape
    m = ys[0].shape[0]
    for x, y in zip(xs, ys):
        if tuple(x.shape) != (n, D) or tuple(y.shape) != (m, D) or x.dtype != torch.float32 or not x.is_cuda:
            raise ValueError("cost_log_kernels needs equally shaped float32 CUDA blocks")
    K = torch.empty((P, n, m), dtype=torch.float32, device=xs[0].device)
    need = L.otgan_cost_matrix_batched_workspace_bytes(P, n, m, D)
    ws = _workspace(need, xs[0].device)
    arr = ctypes.c_void_p * P
    X = arr(*[x.data_ptr() for x in xs])
    Y = arr(*[y.data_ptr() for y in ys])
    rc = L.otgan_cost_matrix_batched_f32(ctypes.cast(X, ctypes.c_void_p), ctypes.cast(Y, ctypes.c_void_p), P, n, m, D, D,
                                         float(sinkhorn_lambda), int(cost_kind), None, K.data_ptr(), ws.data_ptr(),
                                         ws.numel(), _lib.stream_ptr())
    _lib.check(rc, "otgan_cost_matrix_batched_f32")
    return K


def get_matched_features_rows(features_a, features_b, sinkhorn_lambda, nr_sinkhorn_iter, row_begin,
                              row_count, log_kernels=None):
    """Two-batch matching over the full shard lists, producing only rows
    [row_begin, row_begin+row_count) of the four matched-feature arrays (the rows of the samples
    one data-parallel rank owns).  Returns (f_aa, f_bb, f_ab, f_ba) as [row_count, D] tensors,
    entropy, and the fp64 distance (closed form of calc_distance)."""
    _check_lists(features_a, features_b)
    if len(features_a) % 2 != 0:
        raise ValueError("the two-batch matching needs an even number of shards (train.py:34)")
    fa, fb = _stack(features_a), _stack(features_b)
    L = _lib.lib()
    rows_total, D = fa.shape
    N = rows_total // 2
    dev = fa.device
    if log_kernels is not None:
        log_kernels = log_kernels.contiguous()
        assert tuple(log_kernels.shape) == (6, N, N) and log_kernels.dtype == torch.float32
    outs = [torch.empty((row_count, D), dtype=fa.dtype, device=dev) for _ in range(4)]
    entropy = torch.empty((), dtype=torch.float32, device=dev)
    dist = torch.empty((), dtype=torch.float64, device=dev)
    stats = torch.empty((6, 4), dtype=torch.float64, device=dev)
    need = L.otgan_matching_workspace_bytes(_MODE_TWO, N, D)
    ws = _workspace(need, dev)
    rc = L.otgan_matching_two_batch_rows_f32(fa.data_ptr(), fb.data_ptr(), N, D, D, float(sinkhorn_lambda),
                                             int(nr_sinkhorn_iter), int(row_begin), int(row_count),
                                             _lib.ptr(log_kernels),
                                             outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(),
                                             outs[3].data_ptr(), D, entropy.data_ptr(), dist.data_ptr(),
                                             stats.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr())
    _lib.check(rc, "otgan_matching_two_batch_rows_f32")
    return outs, entropy, dist


def matched_feature_grads(fa, fb, sinkhorn_lambda, nr_sinkhorn_iter, need_b=True, rows=None, log_kernels=None):
    """Training-mode two-batch matching (otgan_matching_two_batch_grad_f32): the reference's injected gradients
    `features_a_a - features_a_b` (train.py:111) and `features_b_b - features_b_a` (train.py:125-126) directly, without
    forming the four matched arrays.  fa, fb: flat [2N, D] feature arrays (shards in order: rows [0, N) = mini-batch 1,
    utils/matching.py:16-19).  `need_b=False` (generator steps) skips the data-side gradient.  `rows=(row_begin,
    row_count)`: only that row range (inside one mini-batch: the samples of one data-parallel rank), `log_kernels` as in
    get_matched_features_rows.  Returns (grad_a, grad_b or None, entropy, distance): [rows, D] float32 tensors, a 0-d
    float32 and a 0-d float64 tensor (closed form of calc_distance from the Sinkhorn statistics)."""
    for t in (fa, fb):
        if not t.is_cuda:
            raise _lib.OtganError("matching needs CUDA (MI355X) tensors; there is no CPU fallback")
        if t.dtype != torch.float32 or t.dim() != 2 or t.shape != fa.shape or t.shape[0] % 2:
            raise ValueError("fa and fb must be float32 [2N, D] tensors of one shape")
    fa, fb = fa.detach().contiguous(), fb.detach().contiguous()
    L = _lib.lib()
    rows_total, D = fa.shape
    N = rows_total // 2
    dev = fa.device
    nrows = rows_total if rows is None else int(rows[1])
    grad_a = torch.empty((nrows, D), dtype=fa.dtype, device=dev)
    grad_b = torch.empty((nrows, D), dtype=fa.dtype, device=dev) if need_b else None
    entropy = torch.empty((), dtype=torch.float32, device=dev)
    dist = torch.empty((), dtype=torch.float64, device=dev)
    ws = _workspace(L.otgan_matching_grad_workspace_bytes(N, D), dev)
    if rows is None:
        if log_kernels is not None:
            raise ValueError("log_kernels are only taken by the row-range variant")
        rc = L.otgan_matching_two_batch_grad_f32(fa.data_ptr(), fb.data_ptr(), N, D, D, float(sinkhorn_lambda),
                                                 int(nr_sinkhorn_iter), grad_a.data_ptr(), _lib.ptr(grad_b), D,
                                                 entropy.data_ptr(), dist.data_ptr(), None, ws.data_ptr(), ws.numel(),
                                                 _lib.stream_ptr())
        _lib.check(rc, "otgan_matching_two_batch_grad_f32")
    else:
        if log_kernels is not None:
            log_kernels = log_kernels.contiguous()
            assert tuple(log_kernels.shape) == (6, N, N) and log_kernels.dtype == torch.float32
        rc = L.otgan_matching_two_batch_rows_grad_f32(fa.data_ptr(), fb.data_ptr(), N, D, D, float(sinkhorn_lambda),
                                                      int(nr_sinkhorn_iter), int(rows[0]), int(rows[1]),
                                                      _lib.ptr(log_kernels), grad_a.data_ptr(), _lib.ptr(grad_b), D,
                                                      entropy.data_ptr(), dist.data_ptr(), None, ws.data_ptr(),
                                                      ws.numel(), _lib.stream_ptr())
        _lib.check(rc, "otgan_matching_two_batch_rows_grad_f32")
    return grad_a, grad_b, entropy, dist


class FeatureStack:
    """The gathered features of one step as the matching GEMMs' operand (otgan_matching_stack_split_f32): six N-row blocks
    [a1 b1 b2 a2 a1 b1] as two scaled fp16 planes, of which a data-parallel rank fills the row ranges its two library calls
    read -- ONCE per step instead of once per call (round 5: 7 424 -> 3 328 split rows for a rank of eight in a generator
    step).  `FeatureStack.for_rank(...)` picks the ranges from the rank's position the way trainer._match needs them."""

    def __init__(self, fa, fb, ranges):
        import ctypes
        for t in (fa, fb):
            if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 2 or t.shape != fa.shape or t.shape[0] % 2:
                raise ValueError("fa and fb must be float32 [2N, D] CUDA tensors of one shape")
        fa, fb = fa.detach().contiguous(), fb.detach().contiguous()
        L = _lib.lib()
        self.N, self.D = fa.shape[0] // 2, fa.shape[1]
        nbytes = L.otgan_matching_stack_bytes(self.N, self.D)
        if not nbytes:
            raise _lib.OtganError("the split-precision matching engine does not take this shape (N >= 256, D % 32 == 0)")
        self.buf = torch.empty(nbytes, dtype=torch.uint8, device=fa.device)
        self.ranges = [(int(b), int(r)) for b, r in ranges]
        n = len(self.ranges)
        rb = (ctypes.c_int * n)(*[b for b, _ in self.ranges])
        rr = (ctypes.c_int * n)(*[r for _, r in self.ranges])
        rc = L.otgan_matching_stack_split_f32(fa.data_ptr(), fb.data_ptr(), self.N, self.D, self.D, n,
                                              ctypes.cast(rb, ctypes.c_void_p), ctypes.cast(rr, ctypes.c_void_p),
                                              self.buf.data_ptr(), _lib.stream_ptr())
        _lib.check(rc, "otgan_matching_stack_split_f32")

    @staticmethod
    def supported(N, D):
        return bool(_lib.lib().otgan_matching_stack_bytes(int(N), int(D)))

    @staticmethod
    def rank_plan(row_begin, row_count, N, need_b):
        """-> (ranges to split, stack rows of the rank's generated / data samples) for the rank that owns rows
        [row_begin, +row_count) of the [2N] global batch.  Stack rows: a1 0, b1 N, b2 2N, a2 3N, a1 4N, b1 5N."""
        half, r0 = divmod(int(row_begin), N)
        if half == 0:
            # g(a1) contracts over [N, 4N) = b1 b2 a2 (also the Y blocks of (a1,a2) (a1,b1) (a1,b2)); with the data-side gradient
            # g(b1) adds [4N, 5N) = a1 -- the rank's own generated rows are then read from that copy
            ranges = [(N, 4 * N)] if need_b else [(N, 3 * N), (r0, row_count)]
            own_gen = (4 * N if need_b else 0) + r0
            own_dat = N + r0
        else:
            # g(a2) contracts over [0, 3N) = a1 b1 b2 (b1, b2: the Y blocks of (b2,b1) (a2,b1) (a2,b2)); g(b2) adds [3N, 6N)
            ranges = [(0, 6 * N)] if need_b else [(0, 3 * N), (3 * N + r0, row_count)]
            own_gen = 3 * N + r0
            own_dat = 2 * N + r0
        return ranges, own_gen, own_dat

    def cost_slices(self, xrows, yrows, nrows, sinkhorn_lambda):
        """K[p] = -lambda * cosine cost of stack rows [xrows[p], +nrows) against the N-row block at yrows[p] -> [P, nrows, N]."""
        import ctypes
        L = _lib.lib()
        P = len(xrows)
        K = torch.empty((P, nrows, self.N), dtype=torch.float32, device=self.buf.device)
        ws = _workspace(max(L.otgan_cost_slices_stack_workspace_bytes(P, nrows, self.N, self.D), 256), self.buf.device)
        xr = (ctypes.c_long * P)(*[int(v) for v in xrows])
        yr = (ctypes.c_long * P)(*[int(v) for v in yrows])
        rc = L.otgan_cost_slices_stack_f32(self.buf.data_ptr(), self.N, self.D, P, ctypes.cast(xr, ctypes.c_void_p),
                                           ctypes.cast(yr, ctypes.c_void_p), int(nrows), float(sinkhorn_lambda), K.data_ptr(),
                                           ws.data_ptr(), ws.numel(), _lib.stream_ptr())
        _lib.check(rc, "otgan_cost_slices_stack_f32")
        return K

    def rows_grad(self, sinkhorn_lambda, nr_sinkhorn_iter, rows, log_kernels, need_b=True):
        """matched_feature_grads(..., rows=rows, log_kernels=log_kernels) reading this stack."""
        L = _lib.lib()
        N, D, dev = self.N, self.D, self.buf.device
        log_kernels = log_kernels.contiguous()
        assert tuple(log_kernels.shape) == (6, N, N) and log_kernels.dtype == torch.float32
        nrows = int(rows[1])
        grad_a = torch.empty((nrows, D), dtype=torch.float32, device=dev)
        grad_b = torch.empty((nrows, D), dtype=torch.float32, device=dev) if need_b else None
        entropy = torch.empty((), dtype=torch.float32, device=dev)
        dist = torch.empty((), dtype=torch.float64, device=dev)
        ws = _workspace(L.otgan_matching_grad_workspace_bytes(N, D), dev)
        rc = L.otgan_matching_two_batch_rows_grad_stack_f32(self.buf.data_ptr(), N, D, float(sinkhorn_lambda),
                                                            int(nr_sinkhorn_iter), int(rows[0]), nrows, log_kernels.data_ptr(),
                                                            grad_a.data_ptr(), _lib.ptr(grad_b), D, entropy.data_ptr(),
                                                            dist.data_ptr(), None, ws.data_ptr(), ws.numel(), _lib.stream_ptr())
        _lib.check(rc, "otgan_matching_two_batch_rows_grad_stack_f32")
        return grad_a, grad_b, entropy, dist


def matched_feature_grads_single_batch(fa, fb, sinkhorn_lambda, nr_sinkhorn_iter, need_b=True, rows=None, log_kernels=None):
    """Training-mode --single_batch matching (otgan_matching_single_batch_grad_f32 / _rows_grad_): the injected gradients
    `features_a_a - features_a_b` (train.py:111) and `features_b_b - features_b_a` (train.py:125-126) of
    get_matched_features_single_batch (utils/matching.py:88-136) directly.  fa, fb: flat [n, D] arrays (all shards);
    `rows=(row_begin, row_count)`: only the rows of one data-parallel rank; `log_kernels` [3, n, n]: the a-a, b-b (both with
    -lambda*999 on the diagonal) and a-b log-kernels, e.g. all-gathered row slices (utils/matching.py:99-104).  Returns
    (grad_a, grad_b or None, entropy, distance) like matched_feature_grads."""
    for t in (fa, fb):
        if not t.is_cuda:
            raise _lib.OtganError("matching needs CUDA (MI355X) tensors; there is no CPU fallback")
        if t.dtype != torch.float32 or t.dim() != 2 or t.shape != fa.shape:
            raise ValueError("fa and fb must be float32 [n, D] tensors of one shape")
    fa, fb = fa.detach().contiguous(), fb.detach().contiguous()
    L = _lib.lib()
    n, D = fa.shape
    dev = fa.device
    r0, cnt = (0, n) if rows is None else (int(rows[0]), int(rows[1]))
    grad_a = torch.empty((cnt, D), dtype=fa.dtype, device=dev)
    grad_b = torch.empty((cnt, D), dtype=fa.dtype, device=dev) if need_b else None
    entropy = torch.empty((), dtype=torch.float32, device=dev)
    dist = torch.empty((), dtype=torch.float64, device=dev)
    ws = _workspace(L.otgan_matching_single_batch_grad_workspace_bytes(n, D), dev)
    if log_kernels is not None:
        log_kernels = log_kernels.contiguous()
        assert tuple(log_kernels.shape) == (3, n, n) and log_kernels.dtype == torch.float32
    rc = L.otgan_matching_single_batch_rows_grad_f32(fa.data_ptr(), fb.data_ptr(), n, D, D, float(sinkhorn_lambda),
                                                     int(nr_sinkhorn_iter), r0, cnt, _lib.ptr(log_kernels),
                                                     grad_a.data_ptr(), _lib.ptr(grad_b), D, entropy.data_ptr(),
                                                     dist.data_ptr(), None, ws.data_ptr(), ws.numel(), _lib.stream_ptr())
    _lib.check(rc, "otgan_matching_single_batch_rows_grad_f32")
    return grad_a, grad_b, entropy, dist


def get_matched_features_single_batch(features_a, features_b, sinkhorn_lambda, nr_sinkhorn_iter):
    """Single-batch matching (reference utils/matching.py:88-136)."""
    _check_lists(features_a, features_b)
    return _run(_MODE_SINGLE, _stack(features_a), _stack(features_b), len(features_a),
                sinkhorn_lambda, nr_sinkhorn_iter, COST_COSINE)


def get_matched_features_random(features_a, features_b):
    """Random matching baseline (reference utils/matching.py:3-9)."""
    features_a, features_b = list(features_a), list(features_b)
    features_a_a = features_a[1:] + features_a[:1]
    features_b_b = features_b[1:] + features_b[:1]
    zero = torch.zeros((), dtype=torch.float32, device=features_a[0].device)
    return MatchedFeatures((features_a_a, features_b_b, features_b, features_a, zero))


def calc_distance(features_a, features_b, matched_features):
    """(sum_i nd_bb + nd_aa - 2 nd_ab) / (2*B*S), fp64 accumulation (matching.py:139-153).
    Returns a 0-d float64 CUDA tensor."""
    S = len(features_a)
    B = features_a[0].shape[0]
    f_aa, f_bb, f_ab = matched_features[0], matched_features[1], matched_features[2]
    a, b = _stack(features_a), _stack(features_b)
    aa, bb, ab = _stack(f_aa), _stack(f_bb), _stack(f_ab)
    if not a.is_cuda:
        raise _lib.OtganError("calc_distance needs CUDA (MI355X) tensors; there is no CPU fallback")
    rows, D = a.shape
    dist = torch.empty((), dtype=torch.float64, device=a.device)
    scratch = torch.empty(4, dtype=torch.float64, device=a.device)
    rc = _lib.lib().otgan_calc_distance_f32(a.data_ptr(), b.data_ptr(), aa.data_ptr(),
                                            bb.data_ptr(), ab.data_ptr(), rows, D,
                                            float(2 * B * S), dist.data_ptr(), scratch.data_ptr(),
                                            _lib.stream_ptr())
    _lib.check(rc, "otgan_calc_distance_f32")
    return dist


def closed_form_distance(matched_features):
    """Cancellation-free two-batch / single-batch distance from the per-problem statistics
    {sum(M), <M,C>} produced by the Sinkhorn kernel:
    nd = sum(M) - <M,C> per problem.  0-d float64 tensor, no extra kernel."""
    st = matched_features.stats
    T = st[:, 2] - st[:, 1]
    if st.shape[0] == 6:   # a1a2, b2b1, a1b1, a1b2, a2b1, a2b2
        rows2 = matched_features.flat[0].shape[0]  # 2N
        return (2 * T[0] + 2 * T[1] - (T[2] + T[3] + T[4] + T[5])) / (2.0 * rows2)
    rows = matched_features.flat[0].shape[0]
    return (T[1] + T[0] - 2 * T[2]) / (2.0 * rows)


# ---- toy variant (reference toy_example/matching_cpu.py) ------------------------------------
def toy_get_matched_features(features_a, features_b, sinkhorn_lambda, nr_sinkhorn_iter):
    """Plain `[2N, n]` tensors, squared-Euclidean/(2n) cost (matching_cpu.py:4-95)."""
    fa = features_a.detach().contiguous()
    fb = features_b.detach().contiguous()
    res = _run(_MODE_TWO, fa, fb, 1, sinkhorn_lambda, nr_sinkhorn_iter, COST_SQEUCLID_MEAN)
    out = MatchedFeatures((*res.flat, res[4]))
    out.flat, out.distance, out.stats = res.flat, res.distance, res.stats
    return out


def toy_calc_distance(features_a, features_b, matched_features):
    """(mean(b*bb) + mean(a*aa) - 2 mean(a*ab)) / 2 (matching_cpu.py:155-164)."""
    a = features_a.detach().contiguous()
    b = features_b.detach().contiguous()
    aa, bb, ab = [m.contiguous() for m in matched_features[:3]]
    rows, D = a.shape
    dist = torch.empty((), dtype=torch.float64, device=a.device)
    scratch = torch.empty(4, dtype=torch.float64, device=a.device)
    rc = _lib.lib().otgan_calc_distance_f32(a.data_ptr(), b.data_ptr(), aa.data_ptr(),
                                            bb.data_ptr(), ab.data_ptr(), rows, D,
                                            float(2 * rows * D), dist.data_ptr(),
                                            scratch.data_ptr(), _lib.stream_ptr())
    _lib.check(rc, "otgan_calc_distance_f32")
    return dist
