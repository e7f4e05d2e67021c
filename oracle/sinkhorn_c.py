"""ctypes binding of the plain-C CPU oracle (oracle/sinkhorn_c.c). Test infrastructure."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libsinkhorn_oracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/libsinkhorn_oracle.so missing: run `make -C oracle` "
                               "or __graft_entry__.build()")
        _LIB = ctypes.CDLL(path)
        fp = ctypes.POINTER(ctypes.c_float)
        dp = ctypes.POINTER(ctypes.c_double)
        _LIB.otgan_oracle_two_batch_f32.argtypes = [fp, fp, ctypes.c_int, ctypes.c_int,
                                                    ctypes.c_float, ctypes.c_int, ctypes.c_int,
                                                    fp, fp, fp, fp, dp, dp]
        _LIB.otgan_oracle_single_batch_f32.argtypes = [fp, fp, ctypes.c_int, ctypes.c_int,
                                                       ctypes.c_float, ctypes.c_int,
                                                       fp, fp, fp, fp, dp, dp]
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def num_threads():
    return int(lib().otgan_oracle_num_threads())


def two_batch(fa, fb, lam, iters, cost="cosine"):
    """fa, fb: [2N, D] float32. Returns f_aa, f_bb, f_ab, f_ba ([2N,D]), entropy, distance."""
    fa = np.ascontiguousarray(fa, np.float32)
    fb = np.ascontiguousarray(fb, np.float32)
    n2, d = fa.shape
    outs = [np.empty_like(fa) for _ in range(4)]
    ent, dist = ctypes.c_double(), ctypes.c_double()
    rc = lib().otgan_oracle_two_batch_f32(_p(fa), _p(fb), n2 // 2, d, float(lam), int(iters),
                                          0 if cost == "cosine" else 1, *[_p(o) for o in outs],
                                          ctypes.byref(ent), ctypes.byref(dist))
    assert rc == 0
    return (*outs, ent.value, dist.value)


def single_batch(fa, fb, lam, iters):
    fa = np.ascontiguousarray(fa, np.float32)
    fb = np.ascontiguousarray(fb, np.float32)
    n, d = fa.shape
    outs = [np.empty_like(fa) for _ in range(4)]
    ent, dist = ctypes.c_double(), ctypes.c_double()
    rc = lib().otgan_oracle_single_batch_f32(_p(fa), _p(fb), n, d, float(lam), int(iters),
                                             *[_p(o) for o in outs], ctypes.byref(ent),
                                             ctypes.byref(dist))
    assert rc == 0
    return (*outs, ent.value, dist.value)
