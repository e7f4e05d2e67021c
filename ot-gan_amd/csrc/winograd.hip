// winograd.hip -- Winograd F(2x2,3x3) for the folded 5x5 upsampling convolutions (see winograd.h).
//
// Pipeline of one pass (all in the NHWC / channel-contiguous layouts of the rest of the library):
//   forward : V = B^T d B  (x, 4x4 patches)          [WF][T][Cin]
//             U = G g G^T  (four class filters)      [16][4*Cout][Cin]
//             M[f] = V[f] . U[f]^T   16 GEMMs        [WF][T][4*Cout]      <- all the MFMA work
//             y = A^T M A + bias, scattered to the class's output parity
//   dgrad   : the same with d = dy sampled per class, flipped filters, K = 4*Cout, N = Cin
//   wgrad   : dU[f] = V[f]^T . (A dY A^T)[f]  (K = tiles), then dweff = G^T dU G
// The transforms are streaming float4 kernels; the GEMMs use the shared block-GEMM engine.
// The Winograd-domain GEMMs of the convolution layers.  This file is compiled twice (Makefile): X3_PIECES = 2,
// WINO_NS = wino_p2 -- two scaled fp16 pieces per operand element, three MFMAs per product (default) -- and
// X3_PIECES = 3, WINO_NS = wino_p3 -- three bf16 pieces, 24 significand bits, six MFMAs (OTGAN_WINO_PIECES=3).
#ifndef X3_PIECES
#define X3_PIECES 2
#endif
#ifndef WINO_NS
#define WINO_NS wino_p2
#endif
#include "winograd.h"
#include "gemm_x3.h"
#include <type_traits>

#include <stdlib.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <map>
#include <mutex>
#include <vector>

#include "gemm_tile.h"

namespace WINO_NS {

namespace {

using Cfg = GemmCfg<2, 2, 2, 2, 32>;   // 128 x 128 x 32, 66 KB LDS, 2 workgroups / CU

template <auto Kern>
inline void ensure_lds(size_t bytes) {
  static const bool done = [bytes] {
    hipFuncSetAttribute(reinterpret_cast<const void*>(Kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                        (int)bytes);
    return true;
  }();
  (void)done;
}


// ---- F(4x4, 3x3) -------------------------------------------------------------------------------
// Minimal filtering with a 4x4 output tile: a 6x6 transformed tile, 36 products per tile and channel pair =
// 2.25 multiplies per output (direct: 9; F(2x2,3x3): 4).  Interpolation points {0, 1, -1, 1/2, -2, inf}: the data
// and output transforms have dyadic coefficients only (exact in fp32) and this asymmetric set is markedly better
// conditioned than the textbook {0, +-1, +-2}.  Measured against fp64 (exact products, fp32 accumulation -- what the
// split-precision GEMM delivers -- K = 256 .. 1024): 6.8e-7 relative L2, between F(2x2,3x3) on the same GEMM
// (2.2e-7) and a plain fp32 MFMA chain over the direct convolution (1.3e-6); {0, +-1, +-2}: 1.3e-6.
// The filter transform (thirds and fifteenths) is evaluated in fp64.
//   B^T (6x6)                       G (6x3)                   A^T (4x6)
//   1 -3/2   -2  3/2   1  0         1      0      0          1  1  1   1   1  0
//   0   -1  1/2  5/2   1  0         1/3    1/3    1/3        0  1 -1  1/2 -2  0
//   0    1 -5/2  1/2   1  0        -1/3    1/3   -1/3        0  1  1  1/4  4  0
//   0   -2   -1    2   1  0       -16/15  -8/15  -4/15       0  1 -1  1/8 -8  1
//   0  1/2   -1 -1/2   1  0        1/15   -2/15   4/15
//   0    1 -3/2   -2 3/2  1         0      0      1
// Row 0 of G picks tap 0 and row 5 tap 2 alone: a zero outer tap (the even-parity classes of the strided layers)
// makes the filter transform vanish at that index (structural zeros, see s2_present in gemm_x3.h).
constexpr int WA = kWA;        // transformed tile edge (6)
constexpr int WM = kWA - 2;    // output tile edge (4)
constexpr int WF = kWF;        // frequencies (36)
static_assert(WA == 6, "the transforms below are written for F(4x4, 3x3)");

typedef double f64x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// store k .. k+3 of row `row`, frequency f, of a [WF][rows][ld] operand: fp32 row-major in F, or (P non-null)
// as two fp16 planes of the SCALED value in the blocked layout; the scale of frequency f sits in the operand's
// header, X3_HDR floats in front of the planes (written by absmax_kernel before the producer runs)
// (`sc`: the scale of frequency f -- hdr_scale(P, f) for producers that run after the header is written, the producer's
// own table for those that derive the scales themselves, ScaleSrc)
__device__ __forceinline__ float hdr_scale(const u16* P, int f) {
#if X3_PIECES == 2
  return P ? (reinterpret_cast<const float*>(P) - X3_HDR)[16 + f] : 1.f;
#else
  return 1.f;
#endif
}
__device__ __forceinline__ void st_operand(float* F, u16* P, long rows, int ld, int f, long row, int k, f32x4 v, float sc) {
  if (P) {
    const long fs = op_fstride(rows, ld);
#if X3_PIECES == 2
    st_split4h(P, WF * fs, f * fs + op_off(row, k, ld >> 4), v * sc);
#else
    st_split4(P, WF * fs, f * fs + op_off(row, k, ld >> 4), v);   // three bf16 pieces: no scale
#endif
  } else {
    st4(F + ((long)f * rows + row) * ld + k, v);
  }
}

// thread -> (row, group of four k) of an operand producer.  Blocked operands: a wave covers 16 rows x 16 k
// = half a chunk, so every store instruction writes 512 contiguous bytes (whole cache lines; 32-byte row
// segments written from k-fastest threads measured 2-3x slower), a block 16 rows x 64 k.  fp32 operands: k
// fastest across the block.  Launch op_grid(rows, nk4) blocks of 256 threads.
__device__ __forceinline__ bool op_thread(bool blocked, long rows, int nk4, long& row, int& k4) {
  if (blocked) {
    const int kgroups = (nk4 + 15) >> 4;
    const long b = blockIdx.x;
    k4 = (int)(b % kgroups) * 16 + (threadIdx.x >> 6) * 4 + (threadIdx.x & 3);
    row = (b / kgroups) * 16 + ((threadIdx.x >> 2) & 15);
  } else {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    k4 = (int)(idx % nk4);
    row = idx / nk4;
  }
  return row < rows && k4 < nk4;
}
inline int op_grid(long rows, long nk4) { return (int)(((rows + 15) / 16) * ((nk4 + 15) / 16)); }

// ---- the one-dimensional transforms (a 2-D transform = columns, then rows), on float4 = four channels ------
// y = B^T x
__device__ __forceinline__ void bt1(const f32x4 (&x)[WA], f32x4 (&y)[WA]) {
  y[0] = x[0] - 1.5f * x[1] - 2.f * x[2] + 1.5f * x[3] + x[4];
  y[1] = 0.5f * x[2] - x[1] + 2.5f * x[3] + x[4];
  y[2] = x[1] - 2.5f * x[2] + 0.5f * x[3] + x[4];
  y[3] = 2.f * (x[3] - x[1]) - x[2] + x[4];
  y[4] = 0.5f * (x[1] - x[3]) - x[2] + x[4];
  y[5] = x[1] - 1.5f * x[2] - 2.f * x[3] + 1.5f * x[4] + x[5];
}
// y = A^T x
template <class V>
__device__ __forceinline__ void at1(const V (&x)[WA], V (&y)[WM]) {
  const V s = x[1] + x[2], d = x[1] - x[2];
  y[0] = x[0] + s + x[3] + x[4];
  y[1] = d + 0.5f * x[3] - 2.f * x[4];
  y[2] = s + 0.25f * x[3] + 4.f * x[4];
  y[3] = d + 0.125f * x[3] - 8.f * x[4] + x[5];
}
// x = A y   (adjoint of at1)
__device__ __forceinline__ void a1(const f32x4 (&y)[WM], f32x4 (&x)[WA]) {
  x[0] = y[0];
  x[1] = y[0] + y[1] + y[2] + y[3];
  x[2] = y[0] - y[1] + y[2] - y[3];
  x[3] = y[0] + 0.5f * y[1] + 0.25f * y[2] + 0.125f * y[3];
  x[4] = y[0] - 2.f * y[1] + 4.f * y[2] - 8.f * y[3];
  x[5] = y[3];
}
// u = G g   (fp64)
__device__ __forceinline__ void g1(const f64x4 (&g)[3], f64x4 (&u)[WA]) {
  u[0] = g[0];
  u[1] = (g[0] + g[1] + g[2]) * (1.0 / 3.0);
  u[2] = (g[1] - g[0] - g[2]) * (1.0 / 3.0);
  u[3] = (4.0 * g[0] + 2.0 * g[1] + g[2]) * (-4.0 / 15.0);
  u[4] = (g[0] - 2.0 * g[1] + 4.0 * g[2]) * (1.0 / 15.0);
  u[5] = g[2];
}
// g = G^T u   (adjoint of g1; fp32: a weight GRADIENT, measured 5e-7 .. 9e-7 from fp64 either way, and the fp64
// version needed 256 + 66 registers -- one wave per SIMD -- for 2.1 TB/s)
__device__ __forceinline__ void gt1(const f32x4 (&u)[WA], f32x4 (&g)[3]) {
  const f32x4 a = u[1] * (1.f / 3.f), b = u[2] * (1.f / 3.f), c = u[3] * (-4.f / 15.f), d = u[4] * (1.f / 15.f);
  g[0] = u[0] + a - b + 4.f * c + d;
  g[1] = a + b + 2.f * c - 2.f * d;
  g[2] = a - b + c + 4.f * d + u[5];
}
__device__ __forceinline__ f64x4 to_d(f32x4 v) { return __builtin_convertvector(v, f64x4); }
__device__ __forceinline__ f32x4 to_f(f64x4 v) { return __builtin_convertvector(v, f32x4); }

// U = G g G^T of a 3x3 filter (four channels), emitted frequency by frequency through `emit(f, value)`
template <class Emit>
__device__ __forceinline__ void tf_filter(const f32x4 (&g)[3][3], Emit&& emit) {
  f64x4 t[WA][3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const f64x4 col[3] = {to_d(g[0][j]), to_d(g[1][j]), to_d(g[2][j])};
    f64x4 u[WA];
    g1(col, u);
#pragma unroll
    for (int i = 0; i < WA; ++i) t[i][j] = u[i];
  }
#pragma unroll
  for (int i = 0; i < WA; ++i) {
    f64x4 u[WA];
    g1(t[i], u);
#pragma unroll
    for (int j = 0; j < WA; ++j) emit(i * WA + j, to_f(u[j]));
  }
}
// dg = G^T dU G; the columns of dU are pulled through `col(j, out[WA])`
template <class Col>
__device__ __forceinline__ void tf_filter_adj(Col&& col, f32x4 (&dg)[3][3]) {
  f32x4 p[3][WA];
#pragma unroll
  for (int j = 0; j < WA; ++j) {
    f32x4 u[WA], g[3];
    col(j, u);
    gt1(u, g);
#pragma unroll
    for (int i = 0; i < 3; ++i) p[i][j] = g[i];
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) gt1(p[i], dg[i]);
}

// ---- operand scales --------------------------------------------------------------------------------------
// Largest magnitude of the tensor an operand is a transform of, and from it the 36 per-frequency scales
// 2^(14 - e_f), |transform_f| <= gain_i gain_j fold amax < 2^(e_f) (gemm_x3.h), into the operand's header.  One
// launch: every block leaves its maximum in a scratch slot, the last one to finish (a counter that it resets)
// reduces the slots and writes the header.  max is order-independent: deterministic.  A NaN anywhere makes every
// scale NaN (fmaxf would drop it), an infinity too: the layer's output is then NaN, as the fp32 engine's would be.
struct AmaxArgs {
  const float* x;
  long rows, ld;        // rows of C floats, row stride ld
  int C;
  float* hdr;
  float gain[WA];       // absolute row sums of the 1-D transform matrix
  float fold;           // taps summed into one filter tap before the transform (1, or 4 for un-folded upsampling filters)
  int floor_one;        // ELU / CELU applied inside the transform: |act(x)| <= max(|x|, 1)
  int record_only;      // hdr is an amax RECORD (common.h): sub-slot 0 = the maximum, the other sub-slots 0, no scales
  int nrec;             // scales_from_amax_kernel: x points to this many consecutive records (their maximum counts)
  float* scratch;
  unsigned* counter;
};
constexpr float kGainBt[WA] = {7.f, 5.f, 5.f, 6.f, 3.f, 7.f};                         // B^T (data / dgrad-data transform)
constexpr float kGainG[WA] = {1.f, 1.f, 1.f, 28.f / 15.f, 7.f / 15.f, 1.f};           // G (filter transform)
constexpr float kGainA[WA] = {1.f, 4.f, 4.f, 1.875f, 15.f, 1.f};                      // A (output-adjoint transform)
constexpr int kAmaxBlocks = 2048, kAmaxSlots = 64;

// header of an operand from the largest magnitude of its source tensor (threads 0 .. 35 one frequency each)
__device__ __forceinline__ void write_scales(const AmaxArgs& a, float amax, int tid) {
  if (a.record_only == 2) {   // max-accumulate into a record the caller zeroed (or that other producers share)
    if (tid == 0) {
      unsigned* p = reinterpret_cast<unsigned*>(a.hdr);
      const unsigned b = __float_as_uint(amax) & 0x7fffffffu;
      if (b > *p) *p = b;      // (one writer: the last block of this launch; other producers of the record run before / after it on the stream)
    }
    return;
  }
  if (a.record_only) {
    if (tid < kAmaxSub) a.hdr[tid * kAmaxSubStride] = tid == 0 ? amax : 0.f;
    return;
  }
  if (a.floor_one && amax == amax) amax = fmaxf(amax, 1.f);
  if (tid < WF) {
    const int i = tid / WA, j = tid - i * WA;
    const float bound = amax * a.fold * (a.gain[i] * a.gain[j]);
    float sc = 1.f, inv = 1.f;
    if (!(bound <= 3.0e38f)) {          // NaN or infinite
      sc = inv = __builtin_nanf("");
    } else if (bound > 0.f) {
      const int e = x3_scale_exp(amax * a.fold, a.gain[i], a.gain[j]);
      sc = __builtin_ldexpf(1.f, 14 - e);
      inv = __builtin_ldexpf(1.f, e - 14);
    }
    a.hdr[16 + tid] = sc;
    a.hdr[64 + tid] = inv;
  }
  if (tid == 0) a.hdr[0] = amax;
}

__global__ __launch_bounds__(256) void absmax_kernel(AmaxArgs a) {
  __shared__ float red[4];
  __shared__ int last;
  const int tid = threadIdx.x, c4 = a.C >> 2;
  const long n4 = a.rows * c4;
  float m = 0.f;
  bool bad = false;
  auto take = [&](f32x4 v) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bad = bad || !(v[j] == v[j]);
      m = fmaxf(m, fabsf(v[j]));
    }
  };
  const long stride = (long)gridDim.x * 256;
  long i = (long)blockIdx.x * 256 + tid;
  if (a.ld == a.C) {
    // contiguous rows: eight independent 16-byte loads in flight per thread
    for (; i + 7 * stride < n4; i += 8 * stride) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = ld4(a.x + 4 * (i + u * stride));
#pragma unroll
      for (int u = 0; u < 8; ++u) take(v[u]);
    }
  }
  for (; i < n4; i += stride) {
    const long row = i / c4;
    const int q = (int)(i - row * c4);
    take(ld4(a.x + row * a.ld + 4 * q));
  }
  if (bad) m = __builtin_nanf("");
  auto wave_max = [&](float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float w = __shfl_xor(v, o);
      v = (v == v && w == w) ? fmaxf(v, w) : __builtin_nanf("");
    }
    return v;
  };
  auto block_max = [&](float v) {
    v = wave_max(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    float r = red[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) r = (r == r && red[w] == red[w]) ? fmaxf(r, red[w]) : __builtin_nanf("");
    return r;
  };
  m = block_max(m);
  if (tid == 0) {
    // device-scope (cache-bypassing) store, acknowledged before the counter moves; no fence: a device-scope release
    // would write back the whole L2 once per block
    __hip_atomic_store(&a.scratch[blockIdx.x], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    last = __hip_atomic_fetch_add(a.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
  }
  __syncthreads();
  if (!last) return;
  float r = 0.f;
  for (int i = tid; i < (int)gridDim.x; i += 256) {
    const float w = __hip_atomic_load(&a.scratch[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    r = (r == r && w == w) ? fmaxf(r, w) : __builtin_nanf("");
  }
  const float amax = block_max(r);
  write_scales(a, amax, tid);
  if (tid == 0) *a.counter = 0;
}
// the 36 scales from an amax record the caller already has (a.x = the record)
__global__ __launch_bounds__(64) void scales_from_amax_kernel(AmaxArgs a) { write_scales(a, amax_records_value(a.x, a.nrec), threadIdx.x); }

// ---- streaming kernels --------------------------------------------------------------------
// A "view" of a small-grid image: element (n, a, b, c) at p[n*sn + a*sh + b*sw + c].
struct View {
  const float* p;
  long sn, sh, sw;
};
struct WView {
  float* p;
  long sn, sh, sw;
};

// V[f][t][coff + c] = (B^T d B)[f];  d = 6x6 patch at (4ta-1, 4tb-1), zero outside [0,H)x[0,W).
// blockIdx.z selects one of up to four views (dgrad: the four output-parity classes of dy).
// Where a producer kernel takes the scales of its two-piece operand from: rec == null -- the header in front of the planes
// (an earlier launch wrote it: absmax_kernel); else it derives them from the amax record itself, every workgroup for its
// own use, and the first one also leaves them in the header for the GEMM that follows (round 3: one 5 us launch per
// operand less, 17 per DCGAN step).  Same arithmetic as write_scales().
struct ScaleSrc {
  const float* rec;
  int nrec;            // consecutive records behind rec (otgan_conv_desc::x_amax_count / dy_amax_count)
  float gain[WA];
  float fold;
  int floor_one;
};
__device__ __forceinline__ void producer_scale_table(const ScaleSrc& ss, u16* P, float* s_sc) {
#if X3_PIECES == 2
  if (P) {
    const int tid = threadIdx.x;
    if (tid < WF) {
      float* hdr = reinterpret_cast<float*>(P) - X3_HDR;
      float sc;
      if (ss.rec) {
        float amax = amax_records_value(ss.rec, ss.nrec);
        if (ss.floor_one && amax == amax) amax = fmaxf(amax, 1.f);
        const int i = tid / WA, j = tid - i * WA;
        const float bound = amax * ss.fold * (ss.gain[i] * ss.gain[j]);
        float inv = 1.f;
        sc = 1.f;
        if (!(bound <= 3.0e38f)) {          // NaN or infinite
          sc = inv = __builtin_nanf("");
        } else if (bound > 0.f) {
          const int e = x3_scale_exp(amax * ss.fold, ss.gain[i], ss.gain[j]);
          sc = __builtin_ldexpf(1.f, 14 - e);
          inv = __builtin_ldexpf(1.f, e - 14);
        }
        if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) {
          hdr[16 + tid] = sc;
          hdr[64 + tid] = inv;
          if (tid == 0) hdr[0] = amax;
        }
      } else {
        sc = hdr[16 + tid];
      }
      s_sc[tid] = sc;
    }
    __syncthreads();
  }
#endif
}
struct InArgs {
  ScaleSrc ss;
  View v[4];
  int coff[4];
  int H, W, TH, TW, C;   // C = channels of the view
  long T;
  int ldv;               // row length of V
  float* V;
  int s2_skip;           // >= 0: strided layer, the (class = blockIdx.z, f) blocks absent under this index are not stored
  u16* P;                // non-null: write the operand as split planes (blocked layout, op_off) instead of V
  int up;                // 1: the view is a stored small image read through a 2x nearest-neighbour upsample (H, W = upsampled)
  int Cpad;              // > C: k columns [C, Cpad) of the operand are written as zeros (K padded to the GEMM's granule)
};
template <int ACT>
__device__ __forceinline__ f32x4 wino_act(f32x4 v) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (ACT == 1) v[j] = fmaxf(v[j], 0.f);
    if (ACT == 2) v[j] = v[j] > 0.f ? v[j] : expm1f(v[j]);
  }
  return v;
}
// ACT / DOUBLED: the pre-activation of the strided layers is applied to the patch before the transform; CReLU /
// CELU emit two transformed patches (channels c and Creal + c of the view's slot): blockIdx.y = 0 transforms
// act(x), 1 act(-x).  The patch is consumed column by column (B^T d), the rows of (B^T d) B are stored as they
// are produced: 36 float4 of state per thread.
template <int ACT, bool DOUBLED>
__global__ __launch_bounds__(256) void wino_input_kernel(InArgs a) {
  __shared__ float s_sc[WF];
  producer_scale_table(a.ss, a.P, s_sc);
  long t;
  int k4;
  if (!op_thread(a.P != nullptr, a.T, (a.Cpad > a.C ? a.Cpad : a.C) >> 2, t, k4)) return;
  const int c = k4 * 4;
  const int tb = (int)(t % a.TW), ta = (int)((t / a.TW) % a.TH);
  const long n = t / ((long)a.TW * a.TH);
  const int cls = blockIdx.z;
  const View v = a.v[cls];
  const int pass = DOUBLED ? (int)blockIdx.y : 0;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  if (c >= a.C) {   // K padding (single-view, non-doubled callers only)
    for (int f = 0; f < WF; ++f) st_operand(a.V, a.P, a.T, a.ldv, f, t, c, zero, 1.f);
    return;
  }
  f32x4 T[WA][WA];
#pragma unroll
  for (int j = 0; j < WA; ++j) {
    const int q = WM * tb - 1 + j;
    f32x4 col[WA], o[WA];
#pragma unroll
    for (int i = 0; i < WA; ++i) {
      const int r = WM * ta - 1 + i;
      const bool ok = (unsigned)r < (unsigned)a.H && (unsigned)q < (unsigned)a.W;
      f32x4 e = ok ? ld4(v.p + n * v.sn + (r >> a.up) * v.sh + (q >> a.up) * v.sw + c) : zero;
      if (ACT != 0 || DOUBLED) e = wino_act<ACT>(pass ? -e : e);
      col[i] = e;
    }
    bt1(col, o);
#pragma unroll
    for (int i = 0; i < WA; ++i) T[i][j] = o[i];
  }
  const int k0 = a.coff[cls] + c + (pass ? a.C : 0);
#pragma unroll
  for (int i = 0; i < WA; ++i) {
    f32x4 o[WA];
    bt1(T[i], o);
#pragma unroll
    for (int j = 0; j < WA; ++j)
      if (a.s2_skip < 0 || s2_present(cls, i * WA + j, a.s2_skip)) st_operand(a.V, a.P, a.T, a.ldv, i * WA + j, t, k0, o[j], s_sc[i * WA + j]);
  }
}

// dst(n, 4ta+i, 4tb+j, c) (+)= (A^T M A)[i][j] + bias[c],  M[f] = Mh[f][t][coff + c]
struct OutArgs {
  WView v[4];
  int coff[4];
  int TH, TW, C;
  long T;
  int ldm;
  const float* Mh;
  const float* bias;
  int accumulate;
  float* amax;          // amax record of the values written (common.h: amax_commit), or null
  WView gv[4];          // wino_output_glu_kernel: the gated output y[..., :C/2] * sigmoid(y[..., C/2:]) per class view
};
__global__ __launch_bounds__(256) void wino_output_kernel(OutArgs a) {
  const int c4n = a.C >> 2;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= a.T * c4n) return;   // (an exited lane reads as 0 in the record's wave reduction: ds_bpermute of a disabled lane)
  unsigned mb = 0u;
  const int c = (int)(idx % c4n) * 4;
  const long t = idx / c4n;
  const int tb = (int)(t % a.TW), ta = (int)((t / a.TW) % a.TH);
  const long n = t / ((long)a.TW * a.TH);
  const WView v = a.v[blockIdx.z];
  const float* in = a.Mh + t * a.ldm + a.coff[blockIdx.z] + c;
  const long fs = a.T * a.ldm;
  f32x4 S[WM][WA];
#pragma unroll
  for (int j = 0; j < WA; ++j) {
    f32x4 col[WA], o[WM];
#pragma unroll
    for (int i = 0; i < WA; ++i) col[i] = ld4(in + (i * WA + j) * fs);
    at1(col, o);
#pragma unroll
    for (int i = 0; i < WM; ++i) S[i][j] = o[i];
  }
  f32x4 b = {0.f, 0.f, 0.f, 0.f};
  if (a.bias) b = ld4(a.bias + c);
#pragma unroll
  for (int i = 0; i < WM; ++i) {
    f32x4 y[WM];
    at1(S[i], y);
#pragma unroll
    for (int j = 0; j < WM; ++j) {
      float* dst = v.p + n * v.sn + (WM * ta + i) * v.sh + (WM * tb + j) * v.sw + c;
      f32x4 o = y[j] + b;
      if (a.accumulate) o += ld4(dst);
      mb = amax_bits4(o, mb);     // the value that ends up in memory (with `accumulate`: the sum)
      st4(dst, o);
    }
  }
  if (a.amax) amax_commit(a.amax, mb);
}

// The generator's layers end in a gated linear unit over the channel halves (models/dcgan.py:35-36, 50): the same transform
// for channel quad c of the value half and of the gate half by one thread, which writes y (the backward pass needs the
// pre-activation) AND the gated product -- pointwise.hip's glu_fwd4_kernel (a read of y, one launch) disappears; same
// arithmetic, bit-identical.  The value half goes to memory first and comes back from the cache when the gate is known
// (16 quads: holding them would cost 64 registers of a kernel that lives on its loads in flight).
__device__ __forceinline__ float wino_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }
__global__ __launch_bounds__(256) void wino_output_glu_kernel(OutArgs a) {
  const int ch = a.C >> 1, c4n = ch >> 2;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= a.T * c4n) return;
  unsigned mb = 0u;
  const int c = (int)(idx % c4n) * 4;
  const long t = idx / c4n;
  const int tb = (int)(t % a.TW), ta = (int)((t / a.TW) % a.TH);
  const long n = t / ((long)a.TW * a.TH);
  const WView v = a.v[blockIdx.z], gv = a.gv[blockIdx.z];
  const long fs = a.T * a.ldm;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int cc = c + half * ch;
    const float* in = a.Mh + t * a.ldm + a.coff[blockIdx.z] + cc;
    f32x4 S[WM][WA];
#pragma unroll
    for (int j = 0; j < WA; ++j) {
      f32x4 col[WA], o[WM];
#pragma unroll
      for (int i = 0; i < WA; ++i) col[i] = ld4(in + (i * WA + j) * fs);
      at1(col, o);
#pragma unroll
      for (int i = 0; i < WM; ++i) S[i][j] = o[i];
    }
    f32x4 b = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) b = ld4(a.bias + cc);
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      f32x4 y[WM];
      at1(S[i], y);
#pragma unroll
      for (int j = 0; j < WM; ++j) {
        float* dst = v.p + n * v.sn + (WM * ta + i) * v.sh + (WM * tb + j) * v.sw + c;
        const f32x4 o = y[j] + b;
        st4(dst + half * ch, o);
        if (half) {
          const f32x4 val = ld4(dst);
          f32x4 gl;
#pragma unroll
          for (int k = 0; k < 4; ++k) gl[k] = val[k] * wino_sigmoid(o[k]);
          mb = amax_bits4(gl, mb);
          st4(gv.p + n * gv.sn + (WM * ta + i) * gv.sh + (WM * tb + j) * gv.sw + c, gl);
        }
      }
    }
  }
  if (a.amax) amax_commit(a.amax, mb);
}

// zero columns [k0, k1) of every row and frequency of a blocked operand (K padded to the GEMM's granule)
__global__ __launch_bounds__(256) void op_zero_cols_kernel(u16* P, long rows, int ld, int k0, int k1) {
  long row;
  int k4;
  if (!op_thread(true, rows, (k1 - k0) >> 2, row, k4)) return;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  for (int f = 0; f < WF; ++f) st_operand(nullptr, P, rows, ld, f, row, k0 + 4 * k4, z, 1.f);   // zeros: no scale (the header may not exist yet)
}

// dM[f][t][coff + c] = (A dY A^T)[f],  dY = the 4x4 tile of the view at (4ta, 4tb)
__global__ __launch_bounds__(256) void wino_outadj_kernel(InArgs a) {
  __shared__ float s_sc[WF];
  producer_scale_table(a.ss, a.P, s_sc);
  long t;
  int k4;
  if (!op_thread(a.P != nullptr, a.T, a.C >> 2, t, k4)) return;
  const int c = k4 * 4;
  const int tb = (int)(t % a.TW), ta = (int)((t / a.TW) % a.TH);
  const long n = t / ((long)a.TW * a.TH);
  const View v = a.v[blockIdx.z];
  f32x4 u[WA][WM];
#pragma unroll
  for (int j = 0; j < WM; ++j) {
    f32x4 col[WM], o[WA];
#pragma unroll
    for (int i = 0; i < WM; ++i) col[i] = ld4(v.p + n * v.sn + (WM * ta + i) * v.sh + (WM * tb + j) * v.sw + c);
    a1(col, o);
#pragma unroll
    for (int i = 0; i < WA; ++i) u[i][j] = o[i];
  }
#pragma unroll
  for (int i = 0; i < WA; ++i) {
    f32x4 o[WA];
    a1(u[i], o);
#pragma unroll
    for (int j = 0; j < WA; ++j) st_operand(a.V, a.P, a.T, a.ldv, i * WA + j, t, a.coff[blockIdx.z] + c, o[j], s_sc[i * WA + j]);
  }
}

// forward filters: U[f][cls*Cout + co][ci] from weffT[cls][co][tap*Cin + ci]
__global__ __launch_bounds__(256) void wino_filter_fwd_kernel(const float* __restrict__ weffT, long cls_stride,
                                                            int Cin, int Cout, float* __restrict__ U, u16* P) {
  const long rows = 4L * Cout;
  long row;
  int k4;
  if (!op_thread(P != nullptr, rows, Cin >> 2, row, k4)) return;
  const int ci = k4 * 4;
  const int cls = (int)(row / Cout), co = (int)(row % Cout);
  const float* src = weffT + cls * cls_stride + (long)co * 9 * Cin + ci;
  f32x4 g[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) g[i][j] = ld4(src + (long)(i * 3 + j) * Cin);
  tf_filter(g, [&](int f, f32x4 v) { st_operand(U, P, rows, Cin, f, row, ci, v, hdr_scale(P, f)); });
}

// backward filters (flipped taps): U'[f][ci][cls*Cout + co] from weff[cls][tap][ci][co]
__global__ __launch_bounds__(256) void wino_filter_bwd_kernel(const float* __restrict__ weff, long cls_stride,
                                                            int Cin, int Cout, float* __restrict__ U, u16* P) {
  const int c4n = Cout >> 2;
  long row;
  int k4;
  if (!op_thread(P != nullptr, Cin, 4 * c4n, row, k4)) return;
  const int ci = (int)row, cls = k4 / c4n, co = (k4 % c4n) * 4;
  const float* src = weff + cls * cls_stride + (long)ci * Cout + co;
  f32x4 g[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) g[i][j] = ld4(src + (long)((2 - i) * 3 + (2 - j)) * Cin * Cout);
  tf_filter(g, [&](int f, f32x4 v) { st_operand(U, P, Cin, 4 * Cout, f, ci, cls * Cout + co, v, hdr_scale(P, f)); });
}

// The same two transforms straight from the UN-folded 5x5 weights (the fold of the 2x nearest-neighbour upsampling,
// conv.hip fold_weights_kernel, applied in registers: tap k of output parity p lands on folded tap k/2 (p = 0) or
// (k+1)/2 (p = 1); sums in the fold kernel's order, so the filters are bit-identical to the two-step route).
__device__ __forceinline__ int fold5_tap(int p, int k) { return p ? (k + 1) >> 1 : k >> 1; }

// U[f][cls*Cout + co][ci] from wT[co][(kh*5 + kw)*Cin + ci]
__global__ __launch_bounds__(256) void wino_filter_fwd_unfolded_kernel(const float* __restrict__ wT, int Cin, int Cout,
                                                                     float* __restrict__ U, u16* P) {
  const long rows = 4L * Cout;
  long row;
  int k4;
  if (!op_thread(P != nullptr, rows, Cin >> 2, row, k4)) return;
  const int ci = k4 * 4;
  const int cls = (int)(row / Cout), co = (int)(row % Cout);
  const int ph = cls >> 1, pw = cls & 1;
  const float* src = wT + (long)co * 25 * Cin + ci;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 g[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) g[i][j] = zero;
#pragma unroll
  for (int kh = 0; kh < 5; ++kh)
#pragma unroll
    for (int kw = 0; kw < 5; ++kw) {
      const f32x4 v = ld4(src + (long)(kh * 5 + kw) * Cin);
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
          if (fold5_tap(ph, kh) == i && fold5_tap(pw, kw) == j) g[i][j] += v;
    }
  tf_filter(g, [&](int f, f32x4 v) { st_operand(U, P, rows, Cin, f, row, ci, v, hdr_scale(P, f)); });
}

// U'[f][ci][cls*Cout + co] (flipped taps) from w[kh*5 + kw][ci][co]
__global__ __launch_bounds__(256) void wino_filter_bwd_unfolded_kernel(const float* __restrict__ w, int Cin, int Cout,
                                                                     float* __restrict__ U, u16* P) {
  const int c4n = Cout >> 2;
  long row;
  int k4;
  if (!op_thread(P != nullptr, Cin, 4 * c4n, row, k4)) return;
  const int ci = (int)row, cls = k4 / c4n, co = (k4 % c4n) * 4;
  const int ph = cls >> 1, pw = cls & 1;
  const float* src = w + (long)ci * Cout + co;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 g[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) g[i][j] = zero;
#pragma unroll
  for (int kh = 0; kh < 5; ++kh)
#pragma unroll
    for (int kw = 0; kw < 5; ++kw) {
      const f32x4 v = ld4(src + (long)(kh * 5 + kw) * Cin * Cout);
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
          if (fold5_tap(ph, kh) == 2 - i && fold5_tap(pw, kw) == 2 - j) g[i][j] += v;
    }
  tf_filter(g, [&](int f, f32x4 v) { st_operand(U, P, Cin, 4 * Cout, f, ci, cls * Cout + co, v, hdr_scale(P, f)); });
}

// dweff[cls][tap][ci][co] = (G^T dU G)[tap],  dU[f] = sum over splits of slab[split][f][ci][cls*Cout + co]
__global__ __launch_bounds__(256) void wino_filter_adj_kernel(const float* __restrict__ slabs, int nsplit,
                                                            long split_stride, int Cin, int Cout,
                                                            float* __restrict__ dweff, long cls_stride) {
  const int c4n = Cout >> 2;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= 4L * Cin * c4n) return;
  const int co = (int)(idx % c4n) * 4;
  const long r = idx / c4n;
  const int ci = (int)(r % Cin), cls = (int)(r / Cin);
  const long ldu = 4L * Cout, fs = (long)Cin * ldu;
  const float* src = slabs + (long)ci * ldu + (long)cls * Cout + co;
  f32x4 dg[3][3];
  tf_filter_adj(
      [&](int j, f32x4(&u)[WA]) {
#pragma unroll
        for (int i = 0; i < WA; ++i) {
          f32x4 s = ld4(src + (i * WA + j) * fs);
          for (int k = 1; k < nsplit; ++k) s += ld4(src + k * split_stride + (i * WA + j) * fs);
          u[i] = s;
        }
      },
      dg);
  float* dst = dweff + cls * cls_stride + (long)ci * Cout + co;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) st4(dst + (long)(i * 3 + j) * Cin * Cout, dg[i][j]);
}


// The same with the un-folding of a 5 x 5 'SAME' upsampling layer in the same pass (round 4): dw[kh][kw][ci][co] = sum over
// the four output-parity classes of (G^T dU G)[th(ph, kh)][tw(pw, kw)] -- conv.hip's unfold_wgrad_kernel, whose read of
// dweff (36 / 25 of the weight bytes) and launch disappear.  The four classes of a (ci, co-quad) sit in ADJACENT lanes
// and are added by quad broadcasts in class order from 0, as unfold_wgrad_kernel adds them (equal to 1 - 2 ulp, not bit
// for bit: the compiler contracts the last multiply-adds of the transform differently here).  A first version with one
// thread walking the four classes had a quarter of the threads and four times the dependent loads each: 143 us per
// launch against 119 us for the two kernels it replaced; this one 69 us.
__device__ __forceinline__ float quad_lane(float v, int lane_in_quad) {   // value of lane `lane_in_quad` of the caller's quad
  const int x = __float_as_int(v);
  int r;
  switch (lane_in_quad) {
    case 0: r = __builtin_amdgcn_update_dpp(0, x, 0x00, 0xF, 0xF, true); break;
    case 1: r = __builtin_amdgcn_update_dpp(0, x, 0x55, 0xF, 0xF, true); break;
    case 2: r = __builtin_amdgcn_update_dpp(0, x, 0xAA, 0xF, 0xF, true); break;
    default: r = __builtin_amdgcn_update_dpp(0, x, 0xFF, 0xF, 0xF, true); break;
  }
  return __int_as_float(r);
}
__global__ __launch_bounds__(256) void wino_filter_adj_unfold5_kernel(const float* __restrict__ slabs, int nsplit,
                                                                    long split_stride, int Cin, int Cout,
                                                                    float* __restrict__ dw) {
  const int c4n = Cout >> 2;
  const long idx = (long)blockIdx.x * 64 + (threadIdx.x >> 2);
  const int cls = threadIdx.x & 3;
  if (idx >= (long)Cin * c4n) return;          // (whole quads leave together)
  const int co = (int)(idx % c4n) * 4;
  const int ci = (int)(idx / c4n);
  const long ldu = 4L * Cout, fs = (long)Cin * ldu;
  const float* src = slabs + (long)ci * ldu + (long)cls * Cout + co;
  f32x4 dg[3][3];
  tf_filter_adj(
      [&](int j, f32x4(&u)[WA]) {
#pragma unroll
        for (int i = 0; i < WA; ++i) {
          f32x4 sv = ld4(src + (i * WA + j) * fs);
          for (int k = 1; k < nsplit; ++k) sv += ld4(src + k * split_stride + (i * WA + j) * fs);
          u[i] = sv;
        }
      },
      dg);
  // tap of the 3 x 3 class filter that filter row / column k folds into: floor((p + k - 2) / 2) - its minimum
  constexpr int TH[2][5] = {{0, 0, 1, 1, 2}, {0, 1, 1, 2, 2}};
  const bool ph = (cls >> 1) != 0, pw = (cls & 1) != 0;
  float* dst = dw + (long)ci * Cout + co;
#pragma unroll
  for (int kh = 0; kh < 5; ++kh) {
    f32x4 row[3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int c = 0; c < 4; ++c) row[j][c] = ph ? dg[TH[1][kh]][j][c] : dg[TH[0][kh]][j][c];
#pragma unroll
    for (int kw = 0; kw < 5; ++kw) {
      f32x4 sum;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float x = pw ? row[TH[1][kw]][c] : row[TH[0][kw]][c];
        float acc = 0.f + quad_lane(x, 0);
        acc += quad_lane(x, 1);
        acc += quad_lane(x, 2);
        acc += quad_lane(x, 3);
        sum[c] = acc;
      }
      if (((kh * 5 + kw) & 3) == cls) st4(dst + (long)(kh * 5 + kw) * Cin * Cout, sum);
    }
  }
}

// ---- 5x5 stride-2 layers as four stride-1 3x3 sub-convolutions ------------------------------
// Input row 2a + kh - 1 of output row a: kh = 0,2,4 read the ODD input rows at sub-image offsets
// -1,0,+1, kh = 1,3 the EVEN rows at offsets 0,+1.  With the 2-tap windows zero-padded to three
// taps every (row parity, column parity) class is a 3x3 'SAME' correlation of the class's
// sub-image X[r][c] = act(x)[2r+pi][2c+pj] on the OUTPUT grid, and the sum over the four classes
// folds into the contraction index: 36 GEMMs with K = 4*Ceff.  The zero tap empties one frequency index per
// dimension of the even-parity classes: 25 + 30 + 30 + 36 = 121 of the 144 (class, frequency) blocks remain --
// 121 products per 4x4 output tile and channel pair instead of 400 (F(2x2,3x3): 49 per 2x2 tile = 196).
__device__ __forceinline__ int s2_tap(int parity, int i) {   // filter tap of window slot i, or -1
  return parity ? 2 * i : (i == 0 ? -1 : 2 * i - 1);
}

// forward filters: U[f][co][cls*Ceff + ce] from wT[co][(kh*5+kw)*Ceff + ce]
// (plain: a 3x3 stride-1 layer -- one class, wT[co][(i*3+j)*Ceff + ce], every block present)
__global__ __launch_bounds__(256) void wino_s2_filter_fwd_kernel(const float* __restrict__ wT, int Ceff, int Cout,
                                                               float* __restrict__ U, u16* P, int plain, int ld) {
  const int c4n = Ceff >> 2, ncls = plain ? 1 : 4, kk = plain ? 3 : 5;
  long row;
  int k4;
  if (!op_thread(P != nullptr, Cout, ncls * c4n, row, k4)) return;
  const int co = (int)row, cls = k4 / c4n, ce = (k4 % c4n) * 4;
  const int pi = cls >> 1, pj = cls & 1;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 g[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int kh = plain ? i : s2_tap(pi, i), kw = plain ? j : s2_tap(pj, j);
      g[i][j] = (kh >= 0 && kw >= 0) ? ld4(wT + ((long)co * kk * kk + kh * kk + kw) * Ceff + ce) : zero;
    }
  tf_filter(g, [&](int f, f32x4 v) {
    if (plain || s2_present(cls, f, 0))   // absent blocks are never read by the GEMM
      st_operand(U, P, Cout, ld, f, co, cls * Ceff + ce, v, hdr_scale(P, f));
  });
}

// backward filters (flipped): U'[f][cls*Ceff + ce][co] from w[kh*5+kw][ce][co]
__global__ __launch_bounds__(256) void wino_s2_filter_bwd_kernel(const float* __restrict__ w, int Ceff, int Cout,
                                                               float* __restrict__ U, u16* P, int plain, int Kp) {
  long r;                               // cls*Ceff + ce
  int k4;
  const long rows = (plain ? 1L : 4L) * Ceff;
  const int kk = plain ? 3 : 5;
  if (!op_thread(P != nullptr, rows, Kp >> 2, r, k4)) return;
  const int co = k4 * 4;
  if (co >= Cout) {   // K padding: zero columns [Cout, Kp)
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (int f = 0; f < WF; ++f) st_operand(U, P, rows, Kp, f, r, co, z, 1.f);
    return;
  }
  const int ce = (int)(r % Ceff), cls = (int)(r / Ceff);
  const int pi = cls >> 1, pj = cls & 1;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 g[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int kh = plain ? 2 - i : s2_tap(pi, 2 - i), kw = plain ? 2 - j : s2_tap(pj, 2 - j);
      g[i][j] = (kh >= 0 && kw >= 0) ? ld4(w + ((long)(kh * kk + kw) * Ceff + ce) * Cout + co) : zero;
    }
  tf_filter(g, [&](int f, f32x4 v) { st_operand(U, P, rows, Kp, f, r, co, v, hdr_scale(P, f)); });
}

// dw[kh*5+kw][ce][co] = (G^T dU G)[i][j] of the tap's class; dU[f] = sum of slab[split][f][cls*Ceff+ce][co]
__global__ __launch_bounds__(256) void wino_s2_filter_adj_kernel(const float* __restrict__ slabs, int nsplit,
                                                               long split_stride, int Ceff, int Cout,
                                                               float* __restrict__ dw, int plain) {
  const int c4n = Cout >> 2;
  const long rows = (plain ? 1L : 4L) * Ceff;
  const int kk = plain ? 3 : 5;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= rows * c4n) return;
  const int co = (int)(idx % c4n) * 4;
  const long r = idx / c4n;
  const int ce = (int)(r % Ceff), cls = (int)(r / Ceff);
  const int pi = cls >> 1, pj = cls & 1;
  const long fs = rows * Cout;
  const float* src = slabs + r * Cout + co;
  f32x4 dg[3][3];
  tf_filter_adj(
      [&](int j, f32x4(&u)[WA]) {
#pragma unroll
        for (int i = 0; i < WA; ++i) {
          f32x4 sacc = {0.f, 0.f, 0.f, 0.f};
          if (plain || s2_present(cls, i * WA + j, 0)) {
            sacc = ld4(src + (i * WA + j) * fs);
            for (int k = 1; k < nsplit; ++k) sacc += ld4(src + k * split_stride + (i * WA + j) * fs);
          }
          u[i] = sacc;
        }
      },
      dg);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int kh = plain ? i : s2_tap(pi, i), kw = plain ? j : s2_tap(pj, j);
      if (kh >= 0 && kw >= 0) st4(dw + ((long)(kh * kk + kw) * Ceff + ce) * Cout + co, dg[i][j]);
    }
}

// input gradient of a strided layer: per class the 4x4 tile of d/d(act(+-x)) at the class's
// sub-image positions, combined through the activation derivative:
//   dx = act'(x) * G[c] - act'(-x) * G[C + c]        (DOUBLED),   dx = act'(x) * G[c]  otherwise
struct OutS2Args {
  WView dx[4];          // per class: sub-image view of dx
  View x[4];            // per class: sub-image view of x (activation derivative)
  int TH, TW, C, Ceff;  // C = real channels
  long T;
  int ldm;              // 4*Ceff
  const float* Xh;
  int accumulate;
  int plain;            // one class, nothing structurally zero (a 3x3 stride-1 layer)
  int up;               // plain only: x / dx are half-resolution images behind a 2x nearest-neighbour upsample --
                        // the 4x4 tile of gradients is summed over its 2x2 groups
  float* amax;          // amax record of the values written, or null
};
// (A^T M A) of the class's M, rows i0 .. i0+1 only (two output rows at a time keep the register count down)
// DOUBLED (CReLU / CELU): a thread owns TWO channels (both halves of each): the two 36-value column passes of four
// channels needed 256 VGPRs plus 160 AGPR copies -- one wave per SIMD.  Otherwise four channels.
template <int ACT, bool DOUBLED>
__global__ __launch_bounds__(256) void wino_s2_output_kernel(OutS2Args a) {
  constexpr int VW = DOUBLED ? 2 : 4;
  typedef float VT __attribute__((ext_vector_type(VW)));
  auto ldv = [](const float* p) { return *reinterpret_cast<const VT*>(p); };
  const int cvn = a.C / VW;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= a.T * cvn) return;
  unsigned mb = 0u;
  const int c = (int)(idx % cvn) * VW;
  const long t = idx / cvn;
  const int tb = (int)(t % a.TW), ta = (int)((t / a.TW) % a.TH);
  const long n = t / ((long)a.TW * a.TH);
  const int cls = blockIdx.z;
  const int pcls = a.plain ? 3 : cls;   // class 3 has no absent frequencies
  const WView dv = a.dx[cls];
  const View xv = a.x[cls];
  const float* in = a.Xh + t * a.ldm + (long)cls * a.Ceff + c;
  const long fs = a.T * a.ldm;
  VT zero;
#pragma unroll
  for (int q = 0; q < VW; ++q) zero[q] = 0.f;
  VT Sp[WM][WA], Sn[DOUBLED ? WM : 1][WA];
#pragma unroll
  for (int j = 0; j < WA; ++j) {
    VT col[WA], o[WM];
#pragma unroll
    for (int i = 0; i < WA; ++i) col[i] = s2_present(pcls, i * WA + j, WA - 1) ? ldv(in + (i * WA + j) * fs) : zero;
    at1(col, o);
#pragma unroll
    for (int i = 0; i < WM; ++i) Sp[i][j] = o[i];
    if (DOUBLED) {
#pragma unroll
      for (int i = 0; i < WA; ++i) col[i] = s2_present(pcls, i * WA + j, WA - 1) ? ldv(in + a.C + (i * WA + j) * fs) : zero;
      at1(col, o);
#pragma unroll
      for (int i = 0; i < WM; ++i) Sn[DOUBLED ? i : 0][j] = o[i];
    }
  }
  if (a.up) {
    VT pp[2][2], pn[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) pp[i][j] = pn[i][j] = zero;
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      VT yp[WM], yn[WM];
      at1(Sp[i], yp);
      if (DOUBLED) at1(Sn[DOUBLED ? i : 0], yn);
#pragma unroll
      for (int j = 0; j < WM; ++j) {
        pp[i >> 1][j >> 1] += yp[j];
        if (DOUBLED) pn[i >> 1][j >> 1] += yn[j];
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        VT o = pp[i][j];
        if (ACT != 0 || DOUBLED) {
          const VT x4 = ldv(xv.p + n * xv.sn + (2 * ta + i) * xv.sh + (2 * tb + j) * xv.sw + c);
#pragma unroll
          for (int q = 0; q < VW; ++q) {
            const float xq = x4[q];
            float dp = 1.f, dn = 1.f;
            if (ACT == 1) { dp = xq > 0.f ? 1.f : 0.f; dn = -xq > 0.f ? 1.f : 0.f; }
            if (ACT == 2) { dp = xq > 0.f ? 1.f : expf(xq); dn = -xq > 0.f ? 1.f : expf(-xq); }
            o[q] = dp * pp[i][j][q];
            if (DOUBLED) o[q] -= dn * pn[i][j][q];
          }
        }
        float* dst = dv.p + n * dv.sn + (2 * ta + i) * dv.sh + (2 * tb + j) * dv.sw + c;
        if (a.accumulate) o += ldv(dst);
#pragma unroll
        for (int q = 0; q < VW; ++q) {      // (round 4: a stored pixel is written by this thread alone: its record too)
          const unsigned b = amax_bits(o[q]);
          mb = b > mb ? b : mb;
        }
        *reinterpret_cast<VT*>(dst) = o;
      }
    if (a.amax) amax_commit(a.amax, mb);
    return;
  }
#pragma unroll
  for (int i = 0; i < WM; ++i) {
    VT yp[WM], yn[WM];
    at1(Sp[i], yp);
    if (DOUBLED) at1(Sn[DOUBLED ? i : 0], yn);
#pragma unroll
    for (int j = 0; j < WM; ++j) {
      const long off = n * dv.sn + (WM * ta + i) * dv.sh + (WM * tb + j) * dv.sw + c;
      VT o = yp[j];
      if (ACT != 0 || DOUBLED) {
        const VT x4 = ldv(xv.p + n * xv.sn + (WM * ta + i) * xv.sh + (WM * tb + j) * xv.sw + c);
#pragma unroll
        for (int q = 0; q < VW; ++q) {
          const float xq = x4[q];
          float dp = 1.f, dn = 1.f;
          if (ACT == 1) { dp = xq > 0.f ? 1.f : 0.f; dn = -xq > 0.f ? 1.f : 0.f; }
          if (ACT == 2) { dp = xq > 0.f ? 1.f : expf(xq); dn = -xq > 0.f ? 1.f : expf(-xq); }
          o[q] = dp * yp[j][q];
          if (DOUBLED) o[q] -= dn * yn[j][q];
        }
      }
      float* dst = dv.p + off;
      if (a.accumulate) o += ldv(dst);
#pragma unroll
      for (int q = 0; q < VW; ++q) {      // the value that ends up in memory (with `accumulate`: the sum)
        const unsigned b = amax_bits(o[q]);
        mb = b > mb ? b : mb;
      }
      *reinterpret_cast<VT*>(dst) = o;
    }
  }
  if (a.amax) amax_commit(a.amax, mb);
}


template <bool TN>
__global__ __launch_bounds__(Cfg::THREADS) void wino_bgemm_kernel(BgArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // workgroups go round-robin over the 8 XCDs (one L2 each): give each XCD its own row tiles
  // (or column tiles when there are too few row tiles), so that tiles sharing an operand share an L2
  const int x = blockIdx.x;
  int tm, tn;
  if (a.xmap == 1) {
    const int xcd = x & 7, idx = x >> 3;
    tn = idx % a.tiles_n;
    tm = (idx / a.tiles_n) * 8 + xcd;
  } else if (a.xmap == 2) {
    const int xcd = x & 7, idx = x >> 3;
    tm = idx % a.tiles_m;
    tn = (idx / a.tiles_m) * 8 + xcd;
  } else {
    tn = x % a.tiles_n;
    tm = x / a.tiles_n;
  }
  const int f = lpt_frequency(a.seg_mode, blockIdx.z);
  if (a.seg_mode == 2 || a.seg_mode == 3) {
    // skip a tile whose rows (wgrad) / columns (dgrad) all belong to classes absent at f
    const int lo = (a.seg_mode == 2 ? tn * Cfg::BN : tm * Cfg::BM);
    const int ext = (a.seg_mode == 2 ? a.N : a.M);
    int hi = lo + (a.seg_mode == 2 ? Cfg::BN : Cfg::BM) - 1;
    if (hi >= ext) hi = ext - 1;
    bool any = false;
    for (int c = lo / a.seg_len; c <= hi / a.seg_len; ++c) any = any || s2_present(c, f, a.seg_skip);
    if (!any) return;
  }
  const int nkt_all = (a.K + Cfg::BK - 1) / Cfg::BK;
  const int kt0 = blockIdx.y * a.kt_per_split;
  int nkt = nkt_all - kt0;
  if (nkt > a.kt_per_split) nkt = a.kt_per_split;
  typename Cfg::acc_t acc[Cfg::MT][Cfg::NT];
  zero_acc<Cfg>(acc);
  const int m0 = tm * Cfg::BM, n0 = tn * Cfg::BN;
  if (nkt > 0) {
    if (TN) {
      using LA = MatLoaderR<Cfg, Cfg::BM, true>;
      using LB = MatLoaderR<Cfg, Cfg::BN, true>;
      LA la;
      LB lb;
      la.init(a.A + f * a.sA + (long)kt0 * Cfg::BK * a.lda + m0, a.lda, a.M - m0, a.K - kt0 * Cfg::BK);
      lb.init(a.B + f * a.sB + (long)kt0 * Cfg::BK * a.ldb + n0, a.ldb, a.N - n0, a.K - kt0 * Cfg::BK);
      gemm_mainloop<Cfg>(la, lb, nkt, smem, acc);
    } else if (a.seg_mode == 1) {
      // forward of a strided layer: contract only over the classes present at f (<= 2 contiguous runs)
      using LA = MatLoaderK<Cfg, Cfg::BM, true>;
      using LB = MatLoaderK<Cfg, Cfg::BN, true>;
      int c = 0;
      while (c < 4) {
        if (!s2_present(c, f, a.seg_skip)) {
          ++c;
          continue;
        }
        int e = c + 1;
        while (e < 4 && s2_present(e, f, a.seg_skip)) ++e;
        const int klen = (e - c) * a.seg_len;
        LA la;
        LB lb;
        la.init(a.A + f * a.sA + (long)m0 * a.lda + (long)c * a.seg_len, a.lda, a.M - m0, klen);
        lb.init(a.B + f * a.sB + (long)n0 * a.ldb + (long)c * a.seg_len, a.ldb, a.N - n0, klen);
        gemm_mainloop<Cfg>(la, lb, klen / Cfg::BK, smem, acc);
        c = e;
      }
    } else {
      using LA = MatLoaderK<Cfg, Cfg::BM, true>;
      using LB = MatLoaderK<Cfg, Cfg::BN, true>;
      LA la;
      LB lb;
      la.init(a.A + f * a.sA + (long)m0 * a.lda + (long)kt0 * Cfg::BK, a.lda, a.M - m0, a.K - kt0 * Cfg::BK);
      lb.init(a.B + f * a.sB + (long)n0 * a.ldb + (long)kt0 * Cfg::BK, a.ldb, a.N - n0, a.K - kt0 * Cfg::BK);
      gemm_mainloop<Cfg>(la, lb, nkt, smem, acc);
    }
  }
  float* C = a.C + f * a.sC + blockIdx.y * a.sSplit;
  foreach_acc<Cfg>(acc, [&](int r, int c, int, int, int, float v) {
    const int m = m0 + r, n = n0 + c;
    if (m < a.M && n < a.N) C[(long)m * a.ldc + n] = v;
  });
}




// Frequency-major grid of the split-precision GEMMs (BgArgs::fmap): the 36 frequencies are dealt to the 8 XCDs by
// longest-processing-time-first on their work (strided layers: 4, 2 or 1 parity classes are present at a
// frequency -- contraction runs in the forward pass, surviving tiles in dgrad / wgrad), each XCD's queue longest
// first.  Returns the grid size in x.  (The tile-residue maps of round 1 -- BgArgs::xmap 0 / 1 / 2 -- remain for the fp32
// engine and the matching GEMMs.)
constexpr bool use_fmap() { return true; }
unsigned build_fmap(BgArgs& b) { return x3_build_fmap(b); }

#if X3_PIECES == 2
// The 256 x 128 tile kernel (two workgroups per compute unit; gemm_x3.h) takes every launch it can: two-piece operands, at
// least four K stages, an even stage count (K splits included).  OTGAN_X3_NARROW=0 keeps every launch on the 256 x 256 tile
// (the bit-identity test of the two tiles, tests/test_gemm_engines_gpu.py).
bool x3_narrow_on() {
  static const bool on = [] { const char* e = getenv("OTGAN_X3_NARROW"); return !(e && e[0] == '0'); }();
  return on && use_fmap();
}
template <bool TL>
bool launch_narrow(const BgArgs& b0, int nsplit, int min_k, hipStream_t s) {
  if (!x3_narrow_on() || b0.ztab || b0.N < X3N_BN || min_k < 4 * X3_SK || (min_k / X3_SK) % 2) return false;
  if (b0.seg_mode == 1 && (b0.seg_len / X3_SK) % 2) return false;
  BgArgs b = b0;
  b.tiles_n = (b.N + X3N_BN - 1) / X3N_BN;
  const unsigned gx = build_fmap(b);
  ensure_lds<wino_bgemm_x3n_kernel<TL>>(X3N_LDS);
  b.x_total = gx;
  hipLaunchKernelGGL((wino_bgemm_x3n_kernel<TL>), dim3(gx, nsplit, 1), dim3(X3_THREADS), X3N_LDS, s, b);
  return true;
}
#else
template <bool TL>
bool launch_narrow(const BgArgs&, int, int, hipStream_t) { return false; }
#endif

template <bool TN>
void launch_bgemm(const BgArgs& a, int nsplit, hipStream_t s) {
  size_t lds;
  if (TN) lds = sizeof(float) * 2 * (MatLoaderR<Cfg, Cfg::BM, true>::FLOATS + MatLoaderR<Cfg, Cfg::BN, true>::FLOATS);
  else lds = sizeof(float) * 2 * (MatLoaderK<Cfg, Cfg::BM, true>::FLOATS + MatLoaderK<Cfg, Cfg::BN, true>::FLOATS);
  // executed fp32-equivalent FLOP: WF GEMMs, minus the skipped (class, frequency) blocks of the strided layers
  double flop = 2.0 * WF * (double)a.M * a.N * a.K;
  if (a.seg_mode) flop *= 121.0 / 144.0;
  const bool x3 = !TN && a.Ap != nullptr;
  ProfScope ps(x3 ? OTGAN_PROF_WINO_GEMM_X3 : OTGAN_PROF_WINO_GEMM, x3 ? (double)X3_NTERM * flop : flop, 0.0, s);
  if (x3) {
    ensure_lds<wino_bgemm_x3_kernel<true, false>>(X3_LDS);
    ensure_lds<wino_bgemm_x3_kernel<false, false>>(X3_LDS);
    BgArgs b = a;
    b.tiles_m = (a.M + X3_BM - 1) / X3_BM;
    b.tiles_n = (a.N + X3_BN - 1) / X3_BN;
    const bool m_ok = b.tiles_m % 8 == 0, n_ok = b.tiles_n % 8 == 0;
    if (a.M >= a.N) b.xmap = m_ok ? 1 : n_ok ? 2 : 0;
    else b.xmap = n_ok ? 2 : m_ok ? 1 : 0;
    if (nsplit == 1) b.kt_per_split = a.K / X3_BK;
    b.sAp = op_fstride(a.M, a.K); b.sBp = op_fstride(a.N, a.K);
    b.pA = WF * b.sAp; b.pB = WF * b.sBp;
    b.rbA = (a.M + 31) / 32; b.rbB = (a.N + 31) / 32; b.kblocks = a.K / 16;
    // the pipelined kernel needs >= 4 stages of 16 in every block: the shortest K run of a strided forward is one
    // class (seg_len), the shortest K split is the last one
    int min_k = a.K;
    if (a.seg_mode == 1) min_k = a.seg_len;
    else if (nsplit > 1) min_k = a.K - (nsplit - 1) * b.kt_per_split * X3_BK;
    dim3 grid(b.tiles_m * b.tiles_n, nsplit, WF);
    if (use_fmap()) grid = dim3(build_fmap(b), nsplit, 1);
    // the 256 x 128 tile kernel first (measured best on every DCGAN / DenseNet shape); what it cannot take (three-piece
    // operands, odd stage counts, fewer than four stages) runs on the 256 x 256 tile
    if (launch_narrow<false>(b, nsplit, min_k, s)) return;
    if (min_k >= 4 * X3_SK) hipLaunchKernelGGL((wino_bgemm_x3_kernel<true, false>), grid, dim3(X3_THREADS), X3_LDS, s, b);
    else hipLaunchKernelGGL((wino_bgemm_x3_kernel<false, false>), grid, dim3(X3_THREADS), X3_LDS, s, b);
    return;
  }
  ensure_lds<wino_bgemm_kernel<TN>>(lds);
  BgArgs b = a;
  // partition the LARGER operand across the XCDs (each part is then fetched by one L2 only);
  // the smaller one is re-read by all eight
  const bool m_ok = a.tiles_m % 8 == 0, n_ok = a.tiles_n % 8 == 0;
  if (a.M >= a.N) b.xmap = m_ok ? 1 : n_ok ? 2 : 0;
  else b.xmap = n_ok ? 2 : m_ok ? 1 : 0;
  const dim3 grid(a.tiles_m * a.tiles_n, nsplit, WF);
  hipLaunchKernelGGL((wino_bgemm_kernel<TN>), grid, dim3(Cfg::THREADS), lds, s, b);
}

// C[f][M][N] (slabs per K split) = sum_k A[f][k][m] B[f][k][n] with t-leading split-precision operands
// (Ap: blocked [K rows][M cols], Bp: blocked [K rows][N cols]); K % 32 == 0, M % 32 == 0, N % 32 == 0.
void launch_bgemm_tl(const BgArgs& a, int nsplit, hipStream_t s) {
  double flop = 2.0 * WF * (double)a.M * a.N * a.K;
  if (a.seg_mode) flop *= 121.0 / 144.0;
  ProfScope ps(OTGAN_PROF_WINO_GEMM_X3, (double)X3_NTERM * flop, 0.0, s);
  ensure_lds<wino_bgemm_x3_kernel<true, true>>(X3_LDS);
  ensure_lds<wino_bgemm_x3_kernel<false, true>>(X3_LDS);
  BgArgs b = a;
  b.tiles_m = (a.M + X3_BM - 1) / X3_BM;
  b.tiles_n = (a.N + X3_BN - 1) / X3_BN;
  const bool m_ok = b.tiles_m % 8 == 0, n_ok = b.tiles_n % 8 == 0;
  if (a.M >= a.N) b.xmap = m_ok ? 1 : n_ok ? 2 : 0;
  else b.xmap = n_ok ? 2 : m_ok ? 1 : 0;
  if (nsplit == 1) b.kt_per_split = a.K / X3_BK;
  b.sAp = op_fstride(a.K, a.M); b.sBp = op_fstride(a.K, a.N);
  b.pA = WF * b.sAp; b.pB = WF * b.sBp;
  b.cbA = a.M / 16; b.cbB = a.N / 16;
  const int min_k = nsplit > 1 ? a.K - (nsplit - 1) * b.kt_per_split * X3_BK : a.K;
  dim3 grid(b.tiles_m * b.tiles_n, nsplit, WF);
  if (use_fmap()) grid = dim3(build_fmap(b), nsplit, 1);
  if (launch_narrow<true>(b, nsplit, min_k, s)) return;
  if (min_k >= 4 * X3_SK) hipLaunchKernelGGL((wino_bgemm_x3_kernel<true, true>), grid, dim3(X3_THREADS), X3_LDS, s, b);
  else hipLaunchKernelGGL((wino_bgemm_x3_kernel<false, true>), grid, dim3(X3_THREADS), X3_LDS, s, b);
}

inline int grid1(long n) { return (int)((n + 255) / 256); }

// split-precision (bf16 x 3) operands for the NT GEMMs unless OTGAN_WINO_FP32=1
bool use_x3() {
  static const bool on = [] {
    const char* e = getenv("OTGAN_WINO_FP32");
    return !(e && e[0] == '1');
  }();
  return on;
}
// wgrad on the fp16 / bf16 pipe whenever the other passes are
bool use_x3_wgrad() { return use_x3(); }
// ... straight from the forward-layout operands (t-leading GEMM) where the shape allows (tiles and channels multiples of
// 32); the transposing producers (wino_prodT_kernel) serve the other shapes
bool use_x3_wgrad_tl() { return use_x3_wgrad(); }
// floats of workspace that hold n operand elements (three bf16 planes = 6 bytes per element)
inline size_t operand_floats(size_t n) { return X3_HDR + (X3_NP * n + 1) / 2; }
// the planes of a split operand follow its header
inline u16* op_planes(float* base) { return reinterpret_cast<u16*>(base + X3_HDR); }
// ... and the GEMM reads its output scale from the headers of both operands (two-piece build only)
inline const float* op_hdr(const float* base) { return X3_NP == 2 ? base : nullptr; }

// scratch slots and completion counters of absmax_kernel: owned by the library, one set per process (one process
// per GPU), 64 launches may be in flight at once
struct AmaxScratch {
  float* slots = nullptr;
  unsigned* counters = nullptr;
  int device = -1;
};
AmaxScratch& amax_scratch() {
  static AmaxScratch sc;
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (sc.device != dev) {
    void* p = nullptr;
    const size_t bytes = (size_t)kAmaxSlots * kAmaxBlocks * sizeof(float) + kAmaxSlots * sizeof(unsigned);
    if (hipMalloc(&p, bytes) == hipSuccess) {
      (void)hipMemset(p, 0, bytes);
      sc.slots = static_cast<float*>(p);
      sc.counters = reinterpret_cast<unsigned*>(sc.slots + (size_t)kAmaxSlots * kAmaxBlocks);
      sc.device = dev;
    }
  }
  return sc;
}
// scales of the operand at `base` (header) for a transform with row gains `gain` of the tensor x[rows][C] (row stride
// ld); `given`: the caller's amax record of that tensor (otgan_layers.h) -- then only the 36 scales are computed
void op_scales(const float* x, long rows, int C, long ld, float* base, const float (&gain)[WA], float fold, bool floor_one,
               hipStream_t s, const float* given = nullptr, bool record_only = false, bool record_accumulate = false,
               int given_count = 1) {
  if (X3_NP != 2) return;   // three bf16 pieces carry the full exponent range: no scales
  static std::atomic<unsigned> seq{0};
  AmaxArgs a;
  a.nrec = given_count > 1 ? given_count : 1;
  a.x = x; a.rows = rows; a.ld = rows == 1 ? C : ld; a.C = C; a.hdr = base;
  for (int i = 0; i < WA; ++i) a.gain[i] = gain[i];
  a.fold = fold; a.floor_one = floor_one ? 1 : 0;
  a.record_only = record_only ? (record_accumulate ? 2 : 1) : 0;
  if (given) {
    a.x = given;
    hipLaunchKernelGGL(scales_from_amax_kernel, dim3(1), dim3(64), 0, s, a);
    return;
  }
  AmaxScratch& sc = amax_scratch();
  const unsigned slot = seq.fetch_add(1) % kAmaxSlots;
  a.scratch = sc.slots + (size_t)slot * kAmaxBlocks;
  a.counter = sc.counters + slot;
  const long n4 = rows * (C / 4);
  constexpr long cap = 256;     // one workgroup per compute unit
  long blocks = (n4 + 256 * 8 - 1) / (256 * 8);
  if (blocks > cap) blocks = cap;
  if (blocks > kAmaxBlocks) blocks = kAmaxBlocks;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
}
// elements of a [WF][rows][K] operand in either layout (rows padded to 32, K to 16)
inline size_t op_elems(size_t rows, size_t K) { return WF * ((rows + 31) / 32 * 32) * ((K + 15) / 16 * 16); }

// the four output-parity classes of a [N, 2H, 2W, ld] buffer as small-grid views
template <class V, class P>
void class_views(const WinoGeo& g, P base, int ld, V (&v)[4]) {
  const long OW = 2L * g.W, OH = 2L * g.H;
  for (int cls = 0; cls < 4; ++cls) {
    const int ph = cls >> 1, pw = cls & 1;
    v[cls].p = base + (ph * OW + pw) * ld;
    v[cls].sn = OH * OW * ld;
    v[cls].sh = 2 * OW * ld;
    v[cls].sw = 2L * ld;
  }
}

// K splits of the wgrad GEMM on the bf16 pipe (256 x 256 tiles: few tiles, long K)
int x3_wgrad_splits(int M, int N, long T) {
  const int blocks = ((M + X3_BM - 1) / X3_BM) * ((N + X3_BN - 1) / X3_BN) * WF;
  constexpr int target = 256;
  int ns = (target + blocks - 1) / blocks;
  if (ns > 16) ns = 16;
  const int nkt = (int)((T + X3_BK - 1) / X3_BK);
  while (ns > 1 && nkt / ns < 8) --ns;
  return ns < 1 ? 1 : ns;
}

int wgrad_splits(const WinoGeo& g) {
  const long T = wino_tiles(g);
  const int blocks = ((g.Cin + 127) / 128) * ((4 * g.Cout + 127) / 128) * WF;
  int ns = (1024 + blocks - 1) / blocks;
  if (ns > 8) ns = 8;
  const int nkt = (int)((T + Cfg::BK - 1) / Cfg::BK);
  while (ns > 1 && nkt / ns < 8) --ns;
  return ns < 1 ? 1 : ns;
}

}  // namespace

// scales of an operand whose producer kernel (wino_input_kernel / wino_outadj_kernel, `ia`) is launched next: with the
// caller's amax record the producer derives them itself (ScaleSrc) -- no launch; without one, the reduction as before.
void producer_scales(InArgs& ia, const float* x, long rows, int C, long ld, float* base, const float (&gain)[WA], float fold,
                     bool floor_one, hipStream_t s, const float* given, int given_count = 1) {
  ia.ss.rec = nullptr;
  ia.ss.nrec = 1;
  if (X3_NP != 2) return;
  if (given) {
    ia.ss.rec = given;
    ia.ss.nrec = given_count > 1 ? given_count : 1;
    for (int i = 0; i < WA; ++i) ia.ss.gain[i] = gain[i];
    ia.ss.fold = fold;
    ia.ss.floor_one = floor_one ? 1 : 0;
    return;
  }
  op_scales(x, rows, C, ld, base, gain, fold, floor_one, s, given, false, false, given_count);
}
void wino_absmax(const float* x, long rows, int C, long ld, float* record, hipStream_t s, bool accumulate) {
  const float unit[WA] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
  op_scales(x, rows, C, ld, record, unit, 1.f, false, s, nullptr, true, accumulate);
}

bool winograd_enabled() {
  static const bool on = [] {
    const char* e = getenv("OTGAN_DISABLE_WINOGRAD");
    return !(e && e[0] == '1');
  }();
  return on;
}

// forward and weight gradient build the SAME transformed-input operand (same kernel, same layout, same scales) when
// both run on the split-precision engine with the t-leading weight-gradient GEMM: its size, else 0
size_t wino_x_operand_floats(const WinoGeo& g) {
  const long T = wino_tiles(g);
  const bool ok = use_x3() && g.Cin % X3_BK == 0 && use_x3_wgrad_tl() && T % 32 == 0 && (4 * g.Cout) % 32 == 0;
  return ok ? operand_floats(op_elems(T, g.Cin)) : 0;
}

size_t wino_fwd_ws_floats(const WinoGeo& g) {
  const size_t T = (size_t)wino_tiles(g);
  return operand_floats(op_elems(T, g.Cin)) + operand_floats(op_elems(T, 4 * g.Cout)) +
         operand_floats(std::max(op_elems(4 * g.Cout, g.Cin), op_elems(g.Cin, 4 * g.Cout))) +
         WF * T * (size_t)(4 * g.Cout > g.Cin ? 4 * g.Cout : g.Cin);
}
size_t wino_dgrad_ws_floats(const WinoGeo& g) { return wino_fwd_ws_floats(g); }
size_t wino_wgrad_ws_floats(const WinoGeo& g) {
  const size_t T = (size_t)wino_tiles(g);
  const size_t Tp = (T + 63) / 64 * 64;
  const int ns = std::max(wgrad_splits(g), x3_wgrad_splits(g.Cin, 4 * g.Cout, (long)Tp));
  return operand_floats(std::max(op_elems(g.Cin, Tp), op_elems(Tp, g.Cin))) +
         operand_floats(std::max(op_elems(4 * g.Cout, Tp), op_elems(Tp, 4 * g.Cout))) + (size_t)ns * WF * 4 * g.Cout * g.Cin;
}

size_t wino_filter_floats(const WinoGeo& g, int which) {
  return (which == 0 || which == 2) ? operand_floats(op_elems(4 * g.Cout, g.Cin)) : operand_floats(op_elems(g.Cin, 4 * g.Cout));
}
int wino_prepare_filters(const WinoGeo& g, int which, const float* w, long cls_stride, float* out, hipStream_t s) {
  const int N4 = 4 * g.Cout;
  if (which == 2) {          // forward filters from the un-folded transposed weights wT[Cout][25*Cin]
    const bool x3 = use_x3() && g.Cin % X3_BK == 0;
    if (x3) op_scales(w, 1, 25 * g.Cin * g.Cout, 0, out, kGainG, 4.f, false, s, g.w_amax);   // a class tap = up to 2 x 2 of the 25
    hipLaunchKernelGGL(wino_filter_fwd_unfolded_kernel, dim3(op_grid(N4, g.Cin / 4)), dim3(256), 0, s, w, g.Cin, g.Cout, out,
                       x3 ? op_planes(out) : nullptr);
    return OTGAN_OK;
  }
  if (which == 3) {          // dgrad filters from the un-folded HWIO weights w[25][Cin][Cout]
    const bool x3 = use_x3() && N4 % X3_BK == 0;
    if (x3) op_scales(w, 1, 25 * g.Cin * g.Cout, 0, out, kGainG, 4.f, false, s, g.w_amax);
    hipLaunchKernelGGL(wino_filter_bwd_unfolded_kernel, dim3(op_grid(g.Cin, g.Cout)), dim3(256), 0, s, w, g.Cin, g.Cout, out,
                       x3 ? op_planes(out) : nullptr);
    return OTGAN_OK;
  }
  if (which == 0) {
    const bool x3 = use_x3() && g.Cin % X3_BK == 0;
    if (x3) op_scales(w, 4, 9 * g.Cin * g.Cout, cls_stride, out, kGainG, 1.f, false, s);
    hipLaunchKernelGGL(wino_filter_fwd_kernel, dim3(op_grid(N4, g.Cin / 4)), dim3(256), 0, s, w, cls_stride, g.Cin,
                       g.Cout, out, x3 ? op_planes(out) : nullptr);
  } else {
    const bool x3 = use_x3() && N4 % X3_BK == 0;
    if (x3) op_scales(w, 4, 9 * g.Cin * g.Cout, cls_stride, out, kGainG, 1.f, false, s);
    hipLaunchKernelGGL(wino_filter_bwd_kernel, dim3(op_grid(g.Cin, g.Cout)), dim3(256), 0, s, w, cls_stride, g.Cin,
                       g.Cout, out, x3 ? op_planes(out) : nullptr);
  }
  return OTGAN_OK;
}

int wino_fwd(const WinoGeo& g, const float* x, const float* weffT, long cls_stride, const float* bias, float* y,
             float* ws, hipStream_t s, const float* prep) {
  const long T = wino_tiles(g);
  const int N4 = 4 * g.Cout;
  const bool x3 = use_x3() && g.Cin % X3_BK == 0;
  const size_t nV = op_elems(T, g.Cin), nU = op_elems(N4, g.Cin);
  float* V = (g.x_op && x3) ? g.x_op : ws;
  float* Uws = ws + operand_floats(nV);
  float* Mh = Uws + operand_floats(nU);
  float* U = prep ? const_cast<float*>(prep) : Uws;
  u16* VP = x3 ? op_planes(V) : nullptr;
  u16* UP = x3 ? op_planes(U) : nullptr;
  if (!prep) wino_prepare_filters(g, 0, weffT, cls_stride, U, s);
  InArgs ia;
  memset(&ia, 0, sizeof(ia));
  ia.s2_skip = -1;
  ia.v[0].p = x; ia.v[0].sn = (long)g.H * g.W * g.ldx; ia.v[0].sh = (long)g.W * g.ldx; ia.v[0].sw = g.ldx;
  ia.H = g.H; ia.W = g.W; ia.TH = g.H / WM; ia.TW = g.W / WM; ia.C = g.Cin; ia.T = T; ia.ldv = g.Cin; ia.V = V;
  ia.P = VP;
  if (x3) producer_scales(ia, x, (long)g.N * g.H * g.W, g.Cin, g.ldx, V, kGainBt, 1.f, false, s, g.x_amax);
  hipLaunchKernelGGL((wino_input_kernel<0, false>), dim3(op_grid(T, g.Cin / 4), 1, 1), dim3(256), 0, s, ia);
  BgArgs b;
  memset(&b, 0, sizeof(b));
  b.Ap = VP; b.Bp = UP; b.pA = (long)nV; b.pB = (long)nU;
  if (x3) { b.hdrA = op_hdr(V); b.hdrB = op_hdr(U); }
  b.A = V; b.B = U; b.C = Mh; b.M = (int)T; b.N = N4; b.K = g.Cin;
  b.lda = g.Cin; b.ldb = g.Cin; b.ldc = N4;
  b.sA = T * g.Cin; b.sB = (long)N4 * g.Cin; b.sC = T * N4;
  b.tiles_m = (int)((T + Cfg::BM - 1) / Cfg::BM); b.tiles_n = (N4 + Cfg::BN - 1) / Cfg::BN;
  b.kt_per_split = (g.Cin + Cfg::BK - 1) / Cfg::BK;
  launch_bgemm<false>(b, 1, s);
  OutArgs oa;
  memset(&oa, 0, sizeof(oa));
  class_views(g, y + g.y_coff, g.ldy, oa.v);
  for (int cls = 0; cls < 4; ++cls) oa.coff[cls] = cls * g.Cout;
  oa.TH = g.H / WM; oa.TW = g.W / WM; oa.C = g.Cout; oa.T = T; oa.ldm = N4; oa.Mh = Mh; oa.bias = bias;
  if (g.glu_out) {
    class_views(g, g.glu_out, g.Cout / 2, oa.gv);
    oa.amax = g.glu_amax;
    hipLaunchKernelGGL(wino_output_glu_kernel, dim3(grid1(T * (g.Cout / 8)), 1, 4), dim3(256), 0, s, oa);
    return OTGAN_OK;
  }
  hipLaunchKernelGGL(wino_output_kernel, dim3(grid1(T * (g.Cout / 4)), 1, 4), dim3(256), 0, s, oa);
  return OTGAN_OK;
}

int wino_dgrad(const WinoGeo& g, const float* dy, const float* weff, long cls_stride, float* dx, int lddx,
               int accumulate, float* ws, hipStream_t s, const float* prep) {
  const long T = wino_tiles(g);
  const int K4 = 4 * g.Cout;
  const bool x3 = use_x3() && K4 % X3_BK == 0;
  const size_t nV = op_elems(T, K4), nU = op_elems(g.Cin, K4);
  float* DV = ws;                            // [WF][T][4*Cout]
  float* Uws = DV + operand_floats(nV);      // [WF][Cin][4*Cout]
  float* Xh = Uws + operand_floats(nU);      // [WF][T][Cin]
  float* U = prep ? const_cast<float*>(prep) : Uws;
  u16* VP = x3 ? op_planes(DV) : nullptr;
  u16* UP = x3 ? op_planes(U) : nullptr;
  if (!prep) wino_prepare_filters(g, 1, weff, cls_stride, U, s);
  InArgs ia;
  memset(&ia, 0, sizeof(ia));
  ia.s2_skip = -1;
  class_views(g, dy + g.y_coff, g.ldy, ia.v);
  for (int cls = 0; cls < 4; ++cls) ia.coff[cls] = cls * g.Cout;
  ia.H = g.H; ia.W = g.W; ia.TH = g.H / WM; ia.TW = g.W / WM; ia.C = g.Cout; ia.T = T; ia.ldv = K4; ia.V = DV;
  ia.P = VP;
  if (x3) producer_scales(ia, dy + g.y_coff, (long)g.N * 4 * g.H * g.W, g.Cout, g.ldy, DV, kGainBt, 1.f, false, s, g.dy_amax);
  hipLaunchKernelGGL((wino_input_kernel<0, false>), dim3(op_grid(T, g.Cout / 4), 1, 4), dim3(256), 0, s, ia);
  BgArgs b;
  memset(&b, 0, sizeof(b));
  b.Ap = VP; b.Bp = UP; b.pA = (long)nV; b.pB = (long)nU;
  if (x3) { b.hdrA = op_hdr(DV); b.hdrB = op_hdr(U); }
  b.A = DV; b.B = U; b.C = Xh; b.M = (int)T; b.N = g.Cin; b.K = K4;
  b.lda = K4; b.ldb = K4; b.ldc = g.Cin;
  b.sA = T * K4; b.sB = (long)g.Cin * K4; b.sC = T * g.Cin;
  b.tiles_m = (int)((T + Cfg::BM - 1) / Cfg::BM); b.tiles_n = (g.Cin + Cfg::BN - 1) / Cfg::BN;
  b.kt_per_split = (K4 + Cfg::BK - 1) / Cfg::BK;
  launch_bgemm<false>(b, 1, s);
  OutArgs oa;
  memset(&oa, 0, sizeof(oa));
  oa.v[0].p = dx; oa.v[0].sn = (long)g.H * g.W * lddx; oa.v[0].sh = (long)g.W * lddx; oa.v[0].sw = lddx;
  oa.TH = g.H / WM; oa.TW = g.W / WM; oa.C = g.Cin; oa.T = T; oa.ldm = g.Cin; oa.Mh = Xh; oa.bias = nullptr;
  oa.accumulate = accumulate;
  hipLaunchKernelGGL(wino_output_kernel, dim3(grid1(T * (g.Cin / 4)), 1, 1), dim3(256), 0, s, oa);
  return OTGAN_OK;
}

int wino_wgrad(const WinoGeo& g, const float* x, const float* dy, float* dweff, long cls_stride, float* ws,
               hipStream_t s, float* dw_unfolded5) {
  const long T = wino_tiles(g);
  const int N4 = 4 * g.Cout;
  if (use_x3_wgrad_tl() && T % 32 == 0 && g.Cin % 32 == 0 && N4 % 32 == 0) {
    // the operands of forward (V[tile][Cin]) and dgrad (dM[tile][4 Cout]) as they are: t-leading GEMM over the tiles
    const int ns = x3_wgrad_splits(g.Cin, N4, T);
    const size_t nV = op_elems(T, g.Cin), nM = op_elems(T, N4);
    float* Vb = g.x_op ? g.x_op : ws;    // the forward pass of this x left its operand there
    u16* VP = op_planes(Vb);
    float* Mb = ws + operand_floats(nV);
    u16* MP = op_planes(Mb);
    float* slabs = ws + operand_floats(nV) + operand_floats(nM);
    if (!g.x_op) {
      InArgs ia;
      memset(&ia, 0, sizeof(ia));
      ia.s2_skip = -1;
      ia.v[0].p = x; ia.v[0].sn = (long)g.H * g.W * g.ldx; ia.v[0].sh = (long)g.W * g.ldx; ia.v[0].sw = g.ldx;
      ia.H = g.H; ia.W = g.W; ia.TH = g.H / WM; ia.TW = g.W / WM; ia.C = g.Cin; ia.T = T; ia.ldv = g.Cin; ia.P = VP;
      producer_scales(ia, x, (long)g.N * g.H * g.W, g.Cin, g.ldx, Vb, kGainBt, 1.f, false, s, g.x_amax);
      hipLaunchKernelGGL((wino_input_kernel<0, false>), dim3(op_grid(T, g.Cin / 4), 1, 1), dim3(256), 0, s, ia);
    }
    InArgs da;
    memset(&da, 0, sizeof(da));
    da.s2_skip = -1;
    class_views(g, dy + g.y_coff, g.ldy, da.v);
    for (int cls = 0; cls < 4; ++cls) da.coff[cls] = cls * g.Cout;
    da.H = g.H; da.W = g.W; da.TH = g.H / WM; da.TW = g.W / WM; da.C = g.Cout; da.T = T; da.ldv = N4; da.P = MP;
    producer_scales(da, dy + g.y_coff, (long)g.N * 4 * g.H * g.W, g.Cout, g.ldy, Mb, kGainA, 1.f, false, s, g.dy_amax);
    hipLaunchKernelGGL(wino_outadj_kernel, dim3(op_grid(T, g.Cout / 4), 1, 4), dim3(256), 0, s, da);
    BgArgs b;
    memset(&b, 0, sizeof(b));
    b.Ap = VP; b.Bp = MP; b.hdrA = op_hdr(Vb); b.hdrB = op_hdr(Mb);
    b.C = slabs; b.M = g.Cin; b.N = N4; b.K = (int)T;
    b.ldc = N4; b.sC = (long)g.Cin * N4; b.sSplit = (long)WF * g.Cin * N4;
    b.kt_per_split = (int)((T / X3_BK + ns - 1) / ns);
    launch_bgemm_tl(b, ns, s);
    if (dw_unfolded5)
      hipLaunchKernelGGL(wino_filter_adj_unfold5_kernel, dim3(grid1(4L * g.Cin * (g.Cout / 4))), dim3(256), 0, s, slabs, ns,
                         (long)WF * g.Cin * N4, g.Cin, g.Cout, dw_unfolded5);
    else
      hipLaunchKernelGGL(wino_filter_adj_kernel, dim3(grid1(4L * g.Cin * (g.Cout / 4))), dim3(256), 0, s, slabs, ns,
                         (long)WF * g.Cin * N4, g.Cin, g.Cout, dweff, cls_stride);
    return OTGAN_OK;
  }
  const int ns = wgrad_splits(g);
  const size_t nV = WF * (size_t)T * g.Cin, nM = WF * (size_t)T * N4;
  float* V = ws;                              // [WF][T][Cin]
  float* dM = V + operand_floats(nV);         // [WF][T][4*Cout]
  float* slabs = dM + operand_floats(nM);     // [ns][WF][Cin][4*Cout]
  InArgs ia;
  memset(&ia, 0, sizeof(ia));
  ia.s2_skip = -1;
  ia.v[0].p = x; ia.v[0].sn = (long)g.H * g.W * g.ldx; ia.v[0].sh = (long)g.W * g.ldx; ia.v[0].sw = g.ldx;
  ia.H = g.H; ia.W = g.W; ia.TH = g.H / WM; ia.TW = g.W / WM; ia.C = g.Cin; ia.T = T; ia.ldv = g.Cin; ia.V = V;
  hipLaunchKernelGGL((wino_input_kernel<0, false>), dim3(op_grid(T, g.Cin / 4), 1, 1), dim3(256), 0, s, ia);
  InArgs da;
  memset(&da, 0, sizeof(da));
  da.s2_skip = -1;
  class_views(g, dy + g.y_coff, g.ldy, da.v);
  for (int cls = 0; cls < 4; ++cls) da.coff[cls] = cls * g.Cout;
  da.H = g.H; da.W = g.W; da.TH = g.H / WM; da.TW = g.W / WM; da.C = g.Cout; da.T = T; da.ldv = N4; da.V = dM;
  hipLaunchKernelGGL(wino_outadj_kernel, dim3(op_grid(T, g.Cout / 4), 1, 4), dim3(256), 0, s, da);
  BgArgs b;
  memset(&b, 0, sizeof(b));
  b.A = V; b.B = dM; b.C = slabs; b.M = g.Cin; b.N = N4; b.K = (int)T;
  b.lda = g.Cin; b.ldb = N4; b.ldc = N4;
  b.sA = T * g.Cin; b.sB = T * N4; b.sC = (long)g.Cin * N4; b.sSplit = (long)WF * g.Cin * N4;
  b.tiles_m = (g.Cin + Cfg::BM - 1) / Cfg::BM; b.tiles_n = (N4 + Cfg::BN - 1) / Cfg::BN;
  const int nkt = (int)((T + Cfg::BK - 1) / Cfg::BK);
  b.kt_per_split = (nkt + ns - 1) / ns;
  launch_bgemm<true>(b, ns, s);
  if (dw_unfolded5)
    hipLaunchKernelGGL(wino_filter_adj_unfold5_kernel, dim3(grid1(4L * g.Cin * (g.Cout / 4))), dim3(256), 0, s, slabs, ns,
                       (long)WF * g.Cin * N4, g.Cin, g.Cout, dw_unfolded5);
  else
    hipLaunchKernelGGL(wino_filter_adj_kernel, dim3(grid1(4L * g.Cin * (g.Cout / 4))), dim3(256), 0, s, slabs, ns,
                       (long)WF * g.Cin * N4, g.Cin, g.Cout, dweff, cls_stride);
  return OTGAN_OK;
}

// ---------------------------------------------------------------------------------------------
// stride-2 layers
// ---------------------------------------------------------------------------------------------
namespace {

// the four input-parity sub-images of a [N, H, W, ld] buffer as views on the H/2 x W/2 grid
template <class V, class P>
void parity_views(int H, int W, P base, int ld, V (&v)[4]) {
  for (int cls = 0; cls < 4; ++cls) {
    const int pi = cls >> 1, pj = cls & 1;
    v[cls].p = base + ((long)pi * W + pj) * ld;
    v[cls].sn = (long)H * W * ld;
    v[cls].sh = 2L * W * ld;
    v[cls].sw = 2L * ld;
  }
}

// the views a layer's passes transform: the four parity sub-images of a strided layer, the image itself of a plain one
template <class V, class P>
void s2_views(const WinoS2Geo& g, P base, int ld, V (&v)[4]) {
  if (!g.plain) return parity_views(g.H, g.W, base, ld, v);
  v[0].p = base;
  v[0].sn = (long)g.H * g.W * ld;
  v[0].sh = (long)g.W * ld;
  v[0].sw = ld;
}
inline int s2_k(const WinoS2Geo& g) { return wino_s2_classes(g) * g.Ceff; }
// contraction length of the input-gradient GEMM: Cout, padded (zero columns in both operands) to the split-precision
// GEMM's K granule for the plain layers (208, 144 outputs of the DenseNet transitions)
// contraction length of the forward GEMM: the effective channels, padded the same way for the plain layers (400
// effective channels of the 8x8 critic block)
inline int s2_kf(const WinoS2Geo& g) { return (g.plain && use_x3()) ? (s2_k(g) + X3_BK - 1) / X3_BK * X3_BK : s2_k(g); }
inline int s2_kp(const WinoS2Geo& g) { return (g.plain && use_x3()) ? (g.Cout + X3_BK - 1) / X3_BK * X3_BK : g.Cout; }
inline int s2_taps(const WinoS2Geo& g) { return g.plain ? 9 : 25; }

int s2_wgrad_splits(const WinoS2Geo& g) {
  const long T = wino_s2_tiles(g);
  const int blocks = ((s2_k(g) + 127) / 128) * ((g.Cout + 127) / 128) * WF;
  int ns = (1024 + blocks - 1) / blocks;
  if (ns > 8) ns = 8;
  const int nkt = (int)((T + Cfg::BK - 1) / Cfg::BK);
  while (ns > 1 && nkt / ns < 8) --ns;
  return ns < 1 ? 1 : ns;
}

void s2_input_transform(const WinoS2Geo& g, const float* x, float* V, u16* VP, hipStream_t s, int ld = 0) {
  const long T = wino_s2_tiles(g);
  if (ld == 0) ld = s2_k(g);
  // (the activation is applied inside the transform: |relu(+-x)| <= |x|, |elu(x)| <= max(|x|, 1))
  InArgs ia;
  memset(&ia, 0, sizeof(ia));
  if (VP) producer_scales(ia, x, (long)g.N * (g.H >> g.up) * (g.W >> g.up), g.C, g.ldx, reinterpret_cast<float*>(VP) - X3_HDR, kGainBt, 1.f, g.act == 2, s, g.x_amax, g.x_amax_count);
  ia.s2_skip = -1;
  s2_views(g, x, g.ldx, ia.v);
  if (g.up) {   // the stored image is half the grid (plain layers only)
    ia.up = 1;
    ia.v[0].sn = (long)(g.H / 2) * (g.W / 2) * g.ldx;
    ia.v[0].sh = (long)(g.W / 2) * g.ldx;
  }
  for (int cls = 0; cls < 4; ++cls) ia.coff[cls] = cls * g.Ceff;
  ia.H = wino_s2_out_h(g); ia.W = wino_s2_out_w(g); ia.TH = ia.H / WM; ia.TW = ia.W / WM; ia.C = g.C; ia.T = T; ia.ldv = ld;
  ia.V = V;
  ia.P = VP;
  ia.s2_skip = g.plain ? -1 : 0;
  if (VP && ld > s2_k(g))   // zero columns up to the GEMM's K granule
    hipLaunchKernelGGL(op_zero_cols_kernel, dim3(op_grid(T, (ld - s2_k(g)) / 4)), dim3(256), 0, s, VP, T, ld, s2_k(g), ld);
  const dim3 grid(op_grid(T, g.C / 4), g.doubled ? 2 : 1, wino_s2_classes(g)), blk(256);
  if (g.doubled) {
    if (g.act == 2) hipLaunchKernelGGL((wino_input_kernel<2, true>), grid, blk, 0, s, ia);
    else hipLaunchKernelGGL((wino_input_kernel<1, true>), grid, blk, 0, s, ia);
  } else if (g.act == 1) hipLaunchKernelGGL((wino_input_kernel<1, false>), grid, blk, 0, s, ia);
  else if (g.act == 2) hipLaunchKernelGGL((wino_input_kernel<2, false>), grid, blk, 0, s, ia);
  else hipLaunchKernelGGL((wino_input_kernel<0, false>), grid, blk, 0, s, ia);
}

}  // namespace

size_t wino_s2_x_operand_floats(const WinoS2Geo& g) {
  const long T = wino_s2_tiles(g);
  const int K4 = s2_k(g);
  const bool fwd_x3 = use_x3() && s2_kf(g) == K4 && K4 % X3_BK == 0 && (g.plain || g.Ceff % X3_BK == 0);
  const bool tl = use_x3_wgrad_tl() && T % 32 == 0 && g.Ceff % (g.plain ? 16 : 32) == 0 && g.Cout % 16 == 0;
  return (fwd_x3 && tl) ? operand_floats(op_elems(T, K4)) : 0;
}

size_t wino_s2_fwd_ws_floats(const WinoS2Geo& g) {
  const size_t T = (size_t)wino_s2_tiles(g), K4 = (size_t)s2_k(g);
  const size_t Kp = (size_t)s2_kp(g), Kf = (size_t)s2_kf(g);
  return operand_floats(op_elems(T, Kf)) + operand_floats(op_elems(T, Kp)) +
         operand_floats(std::max(op_elems(g.Cout, Kf), op_elems(K4, Kp))) + WF * T * K4;
}
size_t wino_s2_dgrad_ws_floats(const WinoS2Geo& g) { return wino_s2_fwd_ws_floats(g); }
size_t wino_s2_wgrad_ws_floats(const WinoS2Geo& g) {
  const size_t T = (size_t)wino_s2_tiles(g), K4 = (size_t)s2_k(g);
  const size_t Tp = (T + 63) / 64 * 64;
  const int ns = std::max(s2_wgrad_splits(g), x3_wgrad_splits((int)K4, g.Cout, (long)Tp));
  return operand_floats(std::max(op_elems(K4, Tp), op_elems(Tp, K4))) + operand_floats(std::max(op_elems(g.Cout, Tp), op_elems(Tp, g.Cout))) +
         (size_t)ns * WF * K4 * g.Cout;
}

size_t wino_s2_filter_floats(const WinoS2Geo& g, int which) {
  return which == 0 ? operand_floats(op_elems(g.Cout, s2_kf(g))) : operand_floats(op_elems(s2_k(g), s2_kp(g)));
}
int wino_s2_prepare_filters(const WinoS2Geo& g, int which, const float* w, float* out, hipStream_t s) {
  if (which == 0) {
    const int Kf = s2_kf(g);
    const bool x3 = use_x3() && Kf % X3_BK == 0 && (g.plain || g.Ceff % X3_BK == 0);
    if (x3) op_scales(w, 1, s2_taps(g) * g.Ceff * g.Cout, 0, out, kGainG, 1.f, false, s, g.w_amax);
    hipLaunchKernelGGL(wino_s2_filter_fwd_kernel, dim3(op_grid(g.Cout, s2_k(g) / 4)), dim3(256), 0, s, w, g.Ceff, g.Cout, out,
                       x3 ? op_planes(out) : nullptr, g.plain, x3 ? Kf : s2_k(g));
    if (x3 && Kf > s2_k(g))
      hipLaunchKernelGGL(op_zero_cols_kernel, dim3(op_grid(g.Cout, (Kf - s2_k(g)) / 4)), dim3(256), 0, s, op_planes(out),
                         (long)g.Cout, Kf, s2_k(g), Kf);
  } else {
    const int Kp = s2_kp(g);
    const bool x3 = use_x3() && Kp % X3_BK == 0;
    if (x3) op_scales(w, 1, s2_taps(g) * g.Ceff * g.Cout, 0, out, kGainG, 1.f, false, s, g.w_amax);
    hipLaunchKernelGGL(wino_s2_filter_bwd_kernel, dim3(op_grid(s2_k(g), Kp / 4)), dim3(256), 0, s, w, g.Ceff,
                       g.Cout, out, x3 ? op_planes(out) : nullptr, g.plain, Kp);
  }
  return OTGAN_OK;
}

int wino_s2_fwd(const WinoS2Geo& g, const float* x, const float* wT, const float* bias, float* y, float* ws,
                hipStream_t s, const float* prep) {
  const long T = wino_s2_tiles(g);
  const int K4 = s2_kf(g);                    // classes x effective channels (+ zero columns up to the K granule: plain layers)
  const bool x3 = use_x3() && K4 % X3_BK == 0 && (g.plain || g.Ceff % X3_BK == 0);
  const size_t nV = op_elems(T, K4), nU = op_elems(g.Cout, K4);
  float* V = (g.x_op && x3) ? g.x_op : ws;    // [WF][T][4*Ceff]
  float* Uws = ws + operand_floats(nV);       // [WF][Cout][4*Ceff]
  float* Mh = Uws + operand_floats(nU);       // [WF][T][Cout]
  float* U = prep ? const_cast<float*>(prep) : Uws;
  u16* VP = x3 ? op_planes(V) : nullptr;
  u16* UP = x3 ? op_planes(U) : nullptr;
  if (!prep) wino_s2_prepare_filters(g, 0, wT, U, s);
  s2_input_transform(g, x, V, VP, s, K4);
  BgArgs b;
  memset(&b, 0, sizeof(b));
  b.Ap = VP; b.Bp = UP; b.pA = (long)nV; b.pB = (long)nU;
  if (x3) { b.hdrA = op_hdr(V); b.hdrB = op_hdr(U); }
  b.A = V; b.B = U; b.C = Mh; b.M = (int)T; b.N = g.Cout; b.K = K4;
  b.lda = K4; b.ldb = K4; b.ldc = g.Cout;
  b.sA = T * K4; b.sB = (long)g.Cout * K4; b.sC = T * g.Cout;
  b.tiles_m = (int)((T + Cfg::BM - 1) / Cfg::BM); b.tiles_n = (g.Cout + Cfg::BN - 1) / Cfg::BN;
  b.kt_per_split = (K4 + Cfg::BK - 1) / Cfg::BK;
  b.seg_mode = g.plain ? 0 : 1; b.seg_len = g.Ceff; b.seg_skip = 0;
  launch_bgemm<false>(b, 1, s);
  OutArgs oa;
  memset(&oa, 0, sizeof(oa));
  const int OH = wino_s2_out_h(g), OW = wino_s2_out_w(g);
  oa.v[0].p = y + g.y_coff; oa.v[0].sn = (long)OH * OW * g.ldy; oa.v[0].sh = (long)OW * g.ldy; oa.v[0].sw = g.ldy;
  oa.TH = OH / WM; oa.TW = OW / WM; oa.C = g.Cout; oa.T = T; oa.ldm = g.Cout; oa.Mh = Mh; oa.bias = bias;
  oa.accumulate = g.y_accumulate;
  oa.amax = g.y_amax_out;
  hipLaunchKernelGGL(wino_output_kernel, dim3(grid1(T * (g.Cout / 4)), 1, 1), dim3(256), 0, s, oa);
  return OTGAN_OK;
}

int wino_s2_dgrad(const WinoS2Geo& g, const float* dy, const float* w, const float* x, float* dx, int lddx,
                  int accumulate, float* ws, hipStream_t s, const float* prep) {
  const long T = wino_s2_tiles(g);
  const int K4 = s2_k(g);
  const int OH = wino_s2_out_h(g), OW = wino_s2_out_w(g);
  const int Kp = s2_kp(g);                    // Cout, or Cout padded with zero columns (plain layers)
  const bool x3 = use_x3() && Kp % X3_BK == 0;
  const size_t nV = op_elems(T, Kp), nU = op_elems(K4, Kp);
  float* DV = ws;                             // [WF][T][Kp]
  float* Uws = DV + operand_floats(nV);       // [WF][4*Ceff][Kp]
  float* Xh = Uws + operand_floats(nU);       // [WF][T][4*Ceff]
  float* U = prep ? const_cast<float*>(prep) : Uws;
  u16* VP = x3 ? op_planes(DV) : nullptr;
  u16* UP = x3 ? op_planes(U) : nullptr;
  if (!prep) wino_s2_prepare_filters(g, 1, w, U, s);
  InArgs ia;
  memset(&ia, 0, sizeof(ia));
  ia.s2_skip = -1;
  ia.v[0].p = dy + g.y_coff; ia.v[0].sn = (long)OH * OW * g.ldy; ia.v[0].sh = (long)OW * g.ldy; ia.v[0].sw = g.ldy;
  ia.H = OH; ia.W = OW; ia.TH = OH / WM; ia.TW = OW / WM; ia.C = g.Cout; ia.T = T; ia.ldv = Kp; ia.V = DV;
  ia.P = VP;
  ia.Cpad = Kp;
  if (x3) producer_scales(ia, dy + g.y_coff, (long)g.N * OH * OW, g.Cout, g.ldy, DV, kGainBt, 1.f, false, s, g.dy_amax, g.dy_amax_count);
  hipLaunchKernelGGL((wino_input_kernel<0, false>), dim3(op_grid(T, Kp / 4), 1, 1), dim3(256), 0, s, ia);
  BgArgs b;
  memset(&b, 0, sizeof(b));
  b.Ap = VP; b.Bp = UP; b.pA = (long)nV; b.pB = (long)nU;
  if (x3) { b.hdrA = op_hdr(DV); b.hdrB = op_hdr(U); }
  b.A = DV; b.B = U; b.C = Xh; b.M = (int)T; b.N = K4; b.K = Kp;
  b.lda = Kp; b.ldb = Kp; b.ldc = K4;
  b.sA = T * Kp; b.sB = (long)K4 * Kp; b.sC = T * K4;
  b.tiles_m = (int)((T + Cfg::BM - 1) / Cfg::BM); b.tiles_n = (K4 + Cfg::BN - 1) / Cfg::BN;
  b.kt_per_split = (Kp + Cfg::BK - 1) / Cfg::BK;
  b.seg_mode = g.plain ? 0 : 2; b.seg_len = g.Ceff; b.seg_skip = WA - 1;
  launch_bgemm<false>(b, 1, s);
  OutS2Args oa;
  memset(&oa, 0, sizeof(oa));
  s2_views(g, dx, lddx, oa.dx);
  s2_views(g, x, g.ldx, oa.x);
  oa.plain = g.plain;
  if (g.up) {   // x and dx live on the half-resolution grid
    oa.up = 1;
    oa.dx[0].sn = (long)(g.H / 2) * (g.W / 2) * lddx; oa.dx[0].sh = (long)(g.W / 2) * lddx;
    oa.x[0].sn = (long)(g.H / 2) * (g.W / 2) * g.ldx; oa.x[0].sh = (long)(g.W / 2) * g.ldx;
  }
  oa.TH = OH / WM; oa.TW = OW / WM; oa.C = g.C; oa.Ceff = g.Ceff; oa.T = T; oa.ldm = K4; oa.Xh = Xh;
  oa.accumulate = accumulate;
  oa.amax = g.dx_amax_out;
  const dim3 grid(grid1(T * (g.C / (g.doubled ? 2 : 4))), 1, wino_s2_classes(g)), blk(256);
  if (g.doubled) {
    if (g.act == 2) hipLaunchKernelGGL((wino_s2_output_kernel<2, true>), grid, blk, 0, s, oa);
    else hipLaunchKernelGGL((wino_s2_output_kernel<1, true>), grid, blk, 0, s, oa);
  } else if (g.act == 1) hipLaunchKernelGGL((wino_s2_output_kernel<1, false>), grid, blk, 0, s, oa);
  else if (g.act == 2) hipLaunchKernelGGL((wino_s2_output_kernel<2, false>), grid, blk, 0, s, oa);
  else hipLaunchKernelGGL((wino_s2_output_kernel<0, false>), grid, blk, 0, s, oa);
  return OTGAN_OK;
}

int wino_s2_wgrad(const WinoS2Geo& g, const float* x, const float* dy, float* dw, float* ws, hipStream_t s) {
  const long T = wino_s2_tiles(g);
  const int K4 = s2_k(g);
  const int OH = wino_s2_out_h(g), OW = wino_s2_out_w(g);
  if (use_x3_wgrad_tl() && T % 32 == 0 && g.Ceff % (g.plain ? 16 : 32) == 0 && g.Cout % 16 == 0) {
    // the forward operand V[tile][4 Ceff] (absent (class, frequency) blocks unwritten: their rows of the result are
    // masked by the adjoint filter transform) and the dgrad operand dM[tile][Cout]: t-leading GEMM over the tiles
    const int ns = x3_wgrad_splits(K4, g.Cout, T);
    const size_t nV = op_elems(T, K4), nM = op_elems(T, g.Cout);
    float* Vb = g.x_op ? g.x_op : ws;    // the forward pass of this x left its operand there
    u16* VP = op_planes(Vb);
    float* Mb = ws + operand_floats(nV);
    u16* MP = op_planes(Mb);
    float* slabs = ws + operand_floats(nV) + operand_floats(nM);
    if (!g.x_op) s2_input_transform(g, x, nullptr, VP, s);
    InArgs da;
    memset(&da, 0, sizeof(da));
    da.s2_skip = -1;
    da.v[0].p = dy + g.y_coff; da.v[0].sn = (long)OH * OW * g.ldy; da.v[0].sh = (long)OW * g.ldy; da.v[0].sw = g.ldy;
    da.H = OH; da.W = OW; da.TH = OH / WM; da.TW = OW / WM; da.C = g.Cout; da.T = T; da.ldv = g.Cout; da.P = MP;
    producer_scales(da, dy + g.y_coff, (long)g.N * OH * OW, g.Cout, g.ldy, Mb, kGainA, 1.f, false, s, g.dy_amax, g.dy_amax_count);
    hipLaunchKernelGGL(wino_outadj_kernel, dim3(op_grid(T, g.Cout / 4), 1, 1), dim3(256), 0, s, da);
    BgArgs b;
    memset(&b, 0, sizeof(b));
    b.Ap = VP; b.Bp = MP; b.hdrA = op_hdr(Vb); b.hdrB = op_hdr(Mb);
    b.C = slabs; b.M = K4; b.N = g.Cout; b.K = (int)T;
    b.ldc = g.Cout; b.sC = (long)K4 * g.Cout; b.sSplit = (long)WF * K4 * g.Cout;
    b.kt_per_split = (int)((T / X3_BK + ns - 1) / ns);
    b.seg_mode = g.plain ? 0 : 3; b.seg_len = g.Ceff; b.seg_skip = 0;
    launch_bgemm_tl(b, ns, s);
    hipLaunchKernelGGL(wino_s2_filter_adj_kernel, dim3(grid1((long)K4 * (g.Cout / 4))), dim3(256), 0, s, slabs, ns,
                       (long)WF * K4 * g.Cout, g.Ceff, g.Cout, dw, g.plain);
    return OTGAN_OK;
  }
  const int ns = s2_wgrad_splits(g);
  const size_t nV = WF * (size_t)T * K4, nM = WF * (size_t)T * g.Cout;
  float* V = ws;                              // [WF][T][4*Ceff]
  float* dM = V + operand_floats(nV);         // [WF][T][Cout]
  float* slabs = dM + operand_floats(nM);     // [ns][WF][4*Ceff][Cout]
  s2_input_transform(g, x, V, nullptr, s);
  InArgs da;
  memset(&da, 0, sizeof(da));
  da.s2_skip = -1;
  da.v[0].p = dy + g.y_coff; da.v[0].sn = (long)OH * OW * g.ldy; da.v[0].sh = (long)OW * g.ldy; da.v[0].sw = g.ldy;
  da.H = OH; da.W = OW; da.TH = OH / WM; da.TW = OW / WM; da.C = g.Cout; da.T = T; da.ldv = g.Cout; da.V = dM;
  hipLaunchKernelGGL(wino_outadj_kernel, dim3(op_grid(T, g.Cout / 4), 1, 1), dim3(256), 0, s, da);
  BgArgs b;
  memset(&b, 0, sizeof(b));
  b.A = V; b.B = dM; b.C = slabs; b.M = K4; b.N = g.Cout; b.K = (int)T;
  b.lda = K4; b.ldb = g.Cout; b.ldc = g.Cout;
  b.sA = T * K4; b.sB = T * g.Cout; b.sC = (long)K4 * g.Cout; b.sSplit = (long)WF * K4 * g.Cout;
  b.tiles_m = (K4 + Cfg::BM - 1) / Cfg::BM; b.tiles_n = (g.Cout + Cfg::BN - 1) / Cfg::BN;
  const int nkt = (int)((T + Cfg::BK - 1) / Cfg::BK);
  b.kt_per_split = (nkt + ns - 1) / ns;
  b.seg_mode = g.plain ? 0 : 3; b.seg_len = g.Ceff; b.seg_skip = 0;
  launch_bgemm<true>(b, ns, s);
  hipLaunchKernelGGL(wino_s2_filter_adj_kernel, dim3(grid1((long)K4 * (g.Cout / 4))), dim3(256), 0, s, slabs, ns,
                     (long)WF * K4 * g.Cout, g.Ceff, g.Cout, dw, g.plain);
  return OTGAN_OK;
}

// ---------------------------------------------------------------------------------------------
// 3x3 on a 2x upsampled image with a doubled ReLU (DenseNet generator transitions): forward
// ---------------------------------------------------------------------------------------------
namespace {
// U[f][co][ce] from the un-folded wT[co][(i*3 + j)*Ceff + ce]
__global__ __launch_bounds__(256) void wino_up3_filter_fwd_kernel(const float* __restrict__ wT, int Ceff, int Cout,
                                                                float* __restrict__ U, u16* P) {
  long row;
  int k4;
  if (!op_thread(P != nullptr, Cout, Ceff >> 2, row, k4)) return;
  const int co = (int)row, ce = k4 * 4;
  f32x4 g[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) g[i][j] = ld4(wT + ((long)co * 9 + i * 3 + j) * Ceff + ce);
  tf_filter(g, [&](int f, f32x4 v) { st_operand(U, P, Cout, Ceff, f, co, ce, v, hdr_scale(P, f)); });
}
}  // namespace

size_t wino_up3_filter_floats(const WinoUp3Geo& g) { return operand_floats(op_elems(g.Cout, g.Ceff)); }
size_t wino_up3_fwd_ws_floats(const WinoUp3Geo& g) {
  const size_t T = (size_t)wino_up3_tiles(g);
  return operand_floats(op_elems(T, g.Ceff)) + WF * T * (size_t)g.Cout;
}
int wino_up3_prepare_filters(const WinoUp3Geo& g, const float* wT, float* out, hipStream_t s) {
  op_scales(wT, 1, 9 * g.Ceff * g.Cout, 0, out, kGainG, 1.f, false, s, g.w_amax);
  hipLaunchKernelGGL(wino_up3_filter_fwd_kernel, dim3(op_grid(g.Cout, g.Ceff / 4)), dim3(256), 0, s, wT, g.Ceff, g.Cout, out,
                     op_planes(out));
  return OTGAN_OK;
}
int wino_up3_fwd(const WinoUp3Geo& g, const float* x, const float* bias, float* y, float* ws, hipStream_t s,
                 const float* prep) {
  const long T = wino_up3_tiles(g);
  const int OH = 2 * g.H, OW = 2 * g.W;
  const size_t nV = op_elems(T, g.Ceff), nU = op_elems(g.Cout, g.Ceff);
  float* V = g.x_op ? g.x_op : ws;
  float* Mh = ws + operand_floats(nV);
  float* U = const_cast<float*>(prep);
  InArgs ia;
  memset(&ia, 0, sizeof(ia));
  producer_scales(ia, x, (long)g.N * g.H * g.W, g.C, g.ldx, V, kGainBt, 1.f, false, s, g.x_amax);   // |relu(+-x)| <= |x|
  ia.s2_skip = -1;
  ia.up = 1;
  ia.v[0].p = x; ia.v[0].sn = (long)g.H * g.W * g.ldx; ia.v[0].sh = (long)g.W * g.ldx; ia.v[0].sw = g.ldx;
  ia.H = OH; ia.W = OW; ia.TH = OH / WM; ia.TW = OW / WM; ia.C = g.C; ia.T = T; ia.ldv = g.Ceff; ia.V = V;
  ia.P = op_planes(V);
  hipLaunchKernelGGL((wino_input_kernel<1, true>), dim3(op_grid(T, g.C / 4), 2, 1), dim3(256), 0, s, ia);
  BgArgs b;
  memset(&b, 0, sizeof(b));
  b.Ap = op_planes(V); b.Bp = op_planes(U); b.pA = (long)nV; b.pB = (long)nU;
  b.hdrA = op_hdr(V); b.hdrB = op_hdr(U);
  b.A = V; b.B = U; b.C = Mh; b.M = (int)T; b.N = g.Cout; b.K = g.Ceff;
  b.lda = g.Ceff; b.ldb = g.Ceff; b.ldc = g.Cout;
  b.sA = T * g.Ceff; b.sB = (long)g.Cout * g.Ceff; b.sC = T * g.Cout;
  b.tiles_m = (int)((T + Cfg::BM - 1) / Cfg::BM); b.tiles_n = (g.Cout + Cfg::BN - 1) / Cfg::BN;
  b.kt_per_split = (g.Ceff + Cfg::BK - 1) / Cfg::BK;
  launch_bgemm<false>(b, 1, s);
  OutArgs oa;
  memset(&oa, 0, sizeof(oa));
  oa.v[0].p = y + g.y_coff; oa.v[0].sn = (long)OH * OW * g.ldy; oa.v[0].sh = (long)OW * g.ldy; oa.v[0].sw = g.ldy;
  oa.TH = OH / WM; oa.TW = OW / WM; oa.C = g.Cout; oa.T = T; oa.ldm = g.Cout; oa.Mh = Mh; oa.bias = bias;
  oa.amax = g.y_amax_out;
  hipLaunchKernelGGL(wino_output_kernel, dim3(grid1(T * (g.Cout / 4)), 1, 1), dim3(256), 0, s, oa);
  return OTGAN_OK;
}

}  // namespace WINO_NS
