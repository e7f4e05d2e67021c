"""Test infrastructure (numpy): how exact are dot products of operands stored as split low-precision pieces?

Two schemes of csrc/gemm_x3.h are modelled on the host, products exact, accumulation in fp64, so that what is
measured is the REPRESENTATION error alone (the GPU adds its fp32 accumulation, ~3e-7, on top of either):

  * three bf16 pieces  x = hi + mid + lo (24 significand bits), six products (all with combined weight >= 2^-16);
  * two fp16 pieces of the scaled value  x 2^s = hi + lo (22 bits), three products (hi*hi, hi*lo, lo*hi), with ONE
    power-of-two scale per tensor chosen from its largest magnitude and a gain bound (here the bound is the largest
    magnitude itself or `headroom` times it, as for a Winograd-transformed tensor whose bound is not attained).

Used by tests/test_split_precision_cpu.py; nothing in the product path imports this file."""
import numpy as np


def bf16_round(v):
    u = np.asarray(v, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def split_bf16x3(x):
    h = bf16_round(x)
    r = (x - h).astype(np.float32)
    m = bf16_round(r)
    lo = bf16_round((r - m).astype(np.float32))
    return h, m, lo


def scale_exp(bound):
    """e with bound < 2^e (frexp exponent), 0 for an all-zero tensor: the scale is 2^(14 - e)."""
    return int(np.frexp(np.float32(bound))[1]) if bound > 0 else 0


def split_f16x2(x, headroom=1.0):
    e = scale_exp(float(np.abs(x).max()) * headroom)
    s = np.float32(2.0) ** (14 - e)
    xs = (x * s).astype(np.float32)
    h = xs.astype(np.float16).astype(np.float32)
    lo = (xs - h).astype(np.float16).astype(np.float32)
    return h, lo, s


def dots_bf16x3(A, B):
    ah, am, al = split_bf16x3(A)
    bh, bm, bl = split_bf16x3(B)
    acc = np.zeros((A.shape[0], B.shape[0]))
    for a, b in ((al, bh), (ah, bl), (am, bm), (am, bh), (ah, bm), (ah, bh)):
        acc += a.astype(np.float64) @ b.astype(np.float64).T
    return acc


def dots_f16x2(A, B, headroom=1.0):
    ah, al, sa = split_f16x2(A, headroom)
    bh, bl, sb = split_f16x2(B, headroom)
    acc = np.zeros((A.shape[0], B.shape[0]))
    for a, b in ((al, bh), (ah, bl), (ah, bh)):
        acc += a.astype(np.float64) @ b.astype(np.float64).T
    return acc / (float(sa) * float(sb))


def rel(a, ref):
    return float(np.linalg.norm(a - ref) / np.linalg.norm(ref))


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for K in (256, 1024, 4096):
        for name in ("gauss", "heavy", "tiny", "outlier"):
            A = rng.standard_normal((128, K))
            B = rng.standard_normal((128, K))
            if name == "heavy":
                A *= np.exp(2 * rng.standard_normal(A.shape)); B *= np.exp(2 * rng.standard_normal(B.shape))
            if name == "tiny":
                A *= 1e-7; B *= 1e-7
            if name == "outlier":
                A[0, 0] = 3e4
            A, B = A.astype(np.float32), B.astype(np.float32)
            ref = A.astype(np.float64) @ B.astype(np.float64).T
            print(f"K={K:5d} {name:8s} bf16x3 {rel(dots_bf16x3(A, B), ref):.2e}  f16x2 {rel(dots_f16x2(A, B), ref):.2e}  "
                  f"f16x2 (64x headroom) {rel(dots_f16x2(A, B, 64.0), ref):.2e}  fp32 matmul {rel((A @ B.T).astype(np.float64), ref):.2e}")
