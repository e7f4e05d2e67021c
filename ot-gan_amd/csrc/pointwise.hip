// pointwise.hip -- HBM-bound blocks of the generator / critic and the optimiser:
// weight normalisation (fwd/bwd), column reductions (bias gradients), GLU, tanh, the
// CReLU + flatten + L2-normalise feature head (fwd/bwd), Adam / Adamax / Nesterov, EMA.
// Replaces reference utils/nn.py:29-87,176-181 and models/dcgan.py:16-19,35-36,50.
// All kernels are float4-vectorised, grid-stride, one pass over each array.
#include "common.h"
#include "../../include/otgan.h"

namespace {

constexpr int kChunks = 256;  // max row chunks of the two-stage (deterministic) column reductions

inline int grid_for(long n, int per_thread = 4) {
  long b = ceil_div_l(n, 256L * per_thread);
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

// ---- column reductions: partial[chunk][c] = sum_{r in chunk} op(a[r][c], b[r][c]) ----------
// OP 0: a, 1: a*a, 2: a*b.   Block = 256 threads: lane = column (64 per block), 4 waves split
// the chunk's rows; coalesced 256-byte row segments.
template <int OP>
__global__ __launch_bounds__(256) void colreduce_kernel(const float* __restrict__ a,
                                                        const float* __restrict__ b, long rows,
                                                        int cols, long lda,
                                                        float* __restrict__ partial) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int chunk = blockIdx.y;
  const long per = ceil_div_l(rows, (long)gridDim.y);
  const long r0 = chunk * per;
  long r1 = r0 + per;
  if (r1 > rows) r1 = rows;
  float s = 0.f;
  if (c < cols) {
    for (long r = r0 + wave; r < r1; r += 4) {
      const float av = a[r * lda + c];
      if (OP == 0) s += av;
      else if (OP == 1) s += av * av;
      else s += av * b[r * lda + c];
    }
  }
  __shared__ float red[4][64];
  red[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && c < cols)
    partial[(long)chunk * cols + c] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

// MODE 0: out = sum; 1: out = rsqrt(max(sum, 1e-12))  (tf.nn.l2_normalize epsilon, nn.py:176)
// 256 threads = 64 columns x 4 chunk lanes (fixed summation order: deterministic).
template <int MODE>
__global__ __launch_bounds__(256) void colreduce_finish_kernel(const float* __restrict__ partial,
                                                               int nchunk, int cols,
                                                               float* __restrict__ out) {
  const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  float s = 0.f;
  if (c < cols)
    for (int k = part; k < nchunk; k += 4) s += partial[(long)k * cols + c];
  __shared__ float red[4][64];
  red[part][lane] = s;
  __syncthreads();
  if (part == 0 && c < cols) {
    const float t = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    out[c] = MODE == 1 ? rsqrtf(fmaxf(t, 1e-12f)) : t;
  }
}

// float4 variant: a block covers 64 columns (16 lanes x float4) and 16 row lanes
// (4 per wave), i.e. every wave-load moves 4 rows x 256 bytes.
template <int OP>
__global__ __launch_bounds__(256) void colreduce4_kernel(const float* __restrict__ a,
                                                         const float* __restrict__ b, long rows,
                                                         int cols, long lda,
                                                         float* __restrict__ partial) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cq = lane & 15, rsub = (lane >> 4) + 4 * wave;  // 16 column quads x 16 row lanes
  const int c = blockIdx.x * 64 + 4 * cq;
  const int chunk = blockIdx.y;
  const long per = ceil_div_l(rows, (long)gridDim.y);
  const long r0 = chunk * per;
  long r1 = r0 + per;
  if (r1 > rows) r1 = rows;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < cols) {
    for (long r = r0 + rsub; r < r1; r += 16) {
      const float4 av = *reinterpret_cast<const float4*>(a + r * lda + c);
      if (OP == 0) {
        s.x += av.x; s.y += av.y; s.z += av.z; s.w += av.w;
      } else if (OP == 1) {
        s.x += av.x * av.x; s.y += av.y * av.y; s.z += av.z * av.z; s.w += av.w * av.w;
      } else {
        const float4 bv = *reinterpret_cast<const float4*>(b + r * lda + c);
        s.x += av.x * bv.x; s.y += av.y * bv.y; s.z += av.z * bv.z; s.w += av.w * bv.w;
      }
    }
  }
  __shared__ float4 red[16][17];
  red[rsub][cq] = s;
  __syncthreads();
  if (threadIdx.x < 16 && c < cols) {
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float4 v = red[k][threadIdx.x];
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    const int cc = blockIdx.x * 64 + 4 * threadIdx.x;
    *reinterpret_cast<float4*>(partial + (long)chunk * cols + cc) = t;
  }
}

template <int OP>
int colreduce(const float* a, const float* b, long rows, int cols, long lda, float* partial,
              int* nchunk_out, hipStream_t s) {
  const int colblocks = ceil_div(cols, 64);
  long nchunk = ceil_div(1024, colblocks);              // aim for >= 1024 workgroups
  if (nchunk > kChunks) nchunk = kChunks;
  if (nchunk > ceil_div_l(rows, 64)) nchunk = ceil_div_l(rows, 64);
  if (nchunk < 1) nchunk = 1;
  dim3 grid(colblocks, (int)nchunk);
  const bool vec = (cols % 4 == 0) && (lda % 4 == 0) && ((reinterpret_cast<uintptr_t>(a) & 15) == 0) &&
                   (!b || (reinterpret_cast<uintptr_t>(b) & 15) == 0) &&
                   ((reinterpret_cast<uintptr_t>(partial) & 15) == 0);
  if (vec) hipLaunchKernelGGL(colreduce4_kernel<OP>, grid, dim3(256), 0, s, a, b, rows, cols, lda, partial);
  else hipLaunchKernelGGL(colreduce_kernel<OP>, grid, dim3(256), 0, s, a, b, rows, cols, lda, partial);
  *nchunk_out = (int)nchunk;
  OTGAN_CHECK_LAUNCH("colreduce");
  return OTGAN_OK;
}

// ---- weight norm -----------------------------------------------------------------------------
// w[k][c] = V[k][c] * g[c] * inv[c]; also the transposed copy wT[c][k] through a 32x32 LDS tile.
__global__ __launch_bounds__(256) void weightnorm_apply_kernel(const float* __restrict__ V,
                                                               const float* __restrict__ g,
                                                               const float* __restrict__ inv,
                                                               int K, int Cout,
                                                               float* __restrict__ w,
                                                               float* __restrict__ wT,
                                                               float* __restrict__ rec) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  unsigned mb = 0u;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = k0 + ty + 8 * i, c = c0 + tx;
    float v = 0.f;
    if (k < K && c < Cout) {
      v = V[(long)k * Cout + c] * (g[c] * inv[c]);
      w[(long)k * Cout + c] = v;
    }
    tile[ty + 8 * i][tx] = v;
    const unsigned b = amax_bits(v);
    mb = b > mb ? b : mb;
  }
  if (wT) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = c0 + ty + 8 * i, k = k0 + tx;
      if (k < K && c < Cout) wT[(long)c * K + k] = tile[tx][ty + 8 * i];
    }
  }
  // amax record of the normalised weights (the Winograd filter operands are scaled by it: otgan_conv_desc::w_amax)
  if (rec) amax_commit(rec, mb);
}

// dV = g*inv*(dw - V*dot*inv^2),  dg = dot*inv,  dot[c] = sum_k dw[k][c] V[k][c]
__global__ void weightnorm_bwd_kernel(const float* __restrict__ V, const float* __restrict__ g,
                                      const float* __restrict__ inv, const float* __restrict__ dw,
                                      const float* __restrict__ dot, long total, int Cout,
                                      float* __restrict__ dV) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cout);
    const float iv = inv[c];
    dV[i] = g[c] * iv * (dw[i] - V[i] * dot[c] * iv * iv);
  }
}
// colreduce_finish_kernel<0> + weightnorm_dg_kernel in one launch: dot[c] = sum of the row-chunk partials (same order),
// dg[c] = dot[c] * inv[c]
__global__ __launch_bounds__(256) void wn_dot_finish_kernel(const float* __restrict__ partial, int nchunk, int cols,
                                                            const float* __restrict__ inv, float* __restrict__ dot,
                                                            float* __restrict__ dg) {
  const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  float s = 0.f;
  if (c < cols)
    for (int k = part; k < nchunk; k += 4) s += partial[(long)k * cols + c];
  __shared__ float red[4][64];
  red[part][lane] = s;
  __syncthreads();
  if (part == 0 && c < cols) {
    const float t = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    dot[c] = t;
    dg[c] = t * inv[c];
  }
}
__global__ void weightnorm_dg_kernel(const float* __restrict__ dot, const float* __restrict__ inv,
                                     int Cout, float* __restrict__ dg) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < Cout) dg[c] = dot[c] * inv[c];
}

// ---- weight norm of the 16-output growth layers of a dense block, all layers in one launch ---------------
// (DenseNet: 96 such layers per network; per layer the three / five launches above are a few microseconds of work
// each and the step's Python + launch overhead between them left the GPU idle.)  One workgroup per layer: thread
// (row lane, column quad) walks the rows K of [K][16] matrices with float4 loads, fixed-order LDS tree over the 128
// row lanes (deterministic), second walk writes the results.
constexpr int kWnThreads = 512, kWnRowLanes = kWnThreads / 4;
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
struct WnFwdArgs {
  otgan_wn_fwd_layer l[OTGAN_WN_MAX_LAYERS];
};
struct WnBwdArgs {
  otgan_wn_bwd_layer l[OTGAN_WN_MAX_LAYERS];
};
__device__ __forceinline__ float4 wn_tree(float4 v, float4 (*red)[4], int rl, int cq) {
  red[rl][cq] = v;
  __syncthreads();
#pragma unroll
  for (int h = kWnRowLanes / 2; h > 0; h >>= 1) {
    if (rl < h) {
      const float4 a = red[rl][cq], b = red[rl + h][cq];
      red[rl][cq] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
    __syncthreads();
  }
  const float4 r = red[0][cq];
  __syncthreads();
  return r;
}
__global__ __launch_bounds__(kWnThreads) void wn_fwd_batched_kernel(WnFwdArgs a) {
  __shared__ float4 red[kWnRowLanes][4];
  const otgan_wn_fwd_layer& L = a.l[blockIdx.x];
  const int cq = threadIdx.x & 3, rl = threadIdx.x >> 2;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int r = rl; r < L.K; r += kWnRowLanes) {
    const float4 v = *reinterpret_cast<const float4*>(L.V + (long)r * 16 + 4 * cq);
    s.x += v.x * v.x; s.y += v.y * v.y; s.z += v.z * v.z; s.w += v.w * v.w;
  }
  s = wn_tree(s, red, rl, cq);
  const float4 g = *reinterpret_cast<const float4*>(L.g + 4 * cq);
  const float4 iv = make_float4(rsqrtf(fmaxf(s.x, 1e-12f)), rsqrtf(fmaxf(s.y, 1e-12f)), rsqrtf(fmaxf(s.z, 1e-12f)),
                                rsqrtf(fmaxf(s.w, 1e-12f)));
  if (rl == 0) *reinterpret_cast<float4*>(L.inv + 4 * cq) = iv;
  const float4 sc = make_float4(g.x * iv.x, g.y * iv.y, g.z * iv.z, g.w * iv.w);
  for (int r = rl; r < L.K; r += kWnRowLanes) {
    const float4 v = *reinterpret_cast<const float4*>(L.V + (long)r * 16 + 4 * cq);
    const float4 w = make_float4(v.x * sc.x, v.y * sc.y, v.z * sc.z, v.w * sc.w);
    *reinterpret_cast<float4*>(L.w + (long)r * 16 + 4 * cq) = w;
    if (L.wT) {
      float* t = L.wT + (long)(4 * cq) * L.K + r;
      t[0] = w.x; t[L.K] = w.y; t[2L * L.K] = w.z; t[3L * L.K] = w.w;
    }
  }
}
// dw row (tap, e) of a layer = row e - (rows of the parts before) of the part it falls into
__device__ __forceinline__ const float* wn_dw_row(const otgan_wn_bwd_layer& L, int tap, int e) {
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const otgan_wn_part& p = L.part[i];
    if (e < p.nrows || i == 2) {
      const int ee = p.perm ? p.perm[e] : e;
      return p.p + ((long)tap * p.nrows + ee) * p.rstride;
    }
    e -= p.nrows;
  }
  return nullptr;
}
__global__ __launch_bounds__(kWnThreads) void wn_bwd_batched_kernel(WnBwdArgs a) {
  __shared__ float4 red[kWnRowLanes][4];
  const otgan_wn_bwd_layer& L = a.l[blockIdx.x];
  const int cq = threadIdx.x & 3, rl = threadIdx.x >> 2;
  const int K = L.taps * L.Ceff;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int r = rl; r < K; r += kWnRowLanes) {
    const int tap = r / L.Ceff, e = r - tap * L.Ceff;
    const float4 d = *reinterpret_cast<const float4*>(wn_dw_row(L, tap, e) + 4 * cq);
    const float4 v = *reinterpret_cast<const float4*>(L.V + (long)r * 16 + 4 * cq);
    s.x += d.x * v.x; s.y += d.y * v.y; s.z += d.z * v.z; s.w += d.w * v.w;
  }
  s = wn_tree(s, red, rl, cq);
  const float4 g = *reinterpret_cast<const float4*>(L.g + 4 * cq);
  const float4 iv = *reinterpret_cast<const float4*>(L.inv + 4 * cq);
  if (rl == 0) *reinterpret_cast<float4*>(L.dg + 4 * cq) = make_float4(s.x * iv.x, s.y * iv.y, s.z * iv.z, s.w * iv.w);
  const float4 sc = make_float4(g.x * iv.x, g.y * iv.y, g.z * iv.z, g.w * iv.w);
  const float4 pr = make_float4(s.x * iv.x * iv.x, s.y * iv.y * iv.y, s.z * iv.z * iv.z, s.w * iv.w * iv.w);
  for (int r = rl; r < K; r += kWnRowLanes) {
    const int tap = r / L.Ceff, e = r - tap * L.Ceff;
    const float4 d = *reinterpret_cast<const float4*>(wn_dw_row(L, tap, e) + 4 * cq);
    const float4 v = *reinterpret_cast<const float4*>(L.V + (long)r * 16 + 4 * cq);
    *reinterpret_cast<float4*>(L.dV + (long)r * 16 + 4 * cq) =
        make_float4(sc.x * (d.x - v.x * pr.x), sc.y * (d.y - v.y * pr.y), sc.z * (d.z - v.z * pr.z), sc.w * (d.w - v.w * pr.w));
  }
}

// ---- GLU / tanh --------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__global__ void glu_fwd_kernel(const float* __restrict__ x, long rows, int C, float* __restrict__ y) {
  const long total = rows * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const long r = i / C;
    const int c = (int)(i - r * C);
    const float a = x[r * 2 * C + c], l = x[r * 2 * C + C + c];
    y[i] = a * sigmoidf_(l);
  }
}
__global__ void glu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, long rows,
                               int C, float* __restrict__ dx) {
  const long total = rows * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const long r = i / C;
    const int c = (int)(i - r * C);
    const float a = x[r * 2 * C + c], l = x[r * 2 * C + C + c];
    const float s = sigmoidf_(l), d = dy[i];
    dx[r * 2 * C + c] = d * s;
    dx[r * 2 * C + C + c] = d * a * s * (1.f - s);
  }
}
// float4 forms (C % 4 == 0) that also leave the largest |output| in an amax record (rec may be NULL): the next
// Winograd layer scales its fp16 operands by it, and reading the tensor once more just for that was 0.33 ms of a
// DCGAN step (absmax_kernel, 20 launches)
__global__ __launch_bounds__(256) void glu_fwd4_kernel(const float* __restrict__ x, long rows, int C4,
                                                       float* __restrict__ y, float* __restrict__ rec) {
  const long total = rows * C4;
  unsigned mb = 0u;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / C4;
    const long c = i - r * C4;
    const f32x4 a = *reinterpret_cast<const f32x4*>(x + 4 * (r * 2 * C4 + c));
    const f32x4 l = *reinterpret_cast<const f32x4*>(x + 4 * (r * 2 * C4 + C4 + c));
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = a[k] * sigmoidf_(l[k]);
    *reinterpret_cast<f32x4*>(y + 4 * i) = o;
    mb = amax_bits4(o, mb);
  }
  if (rec) amax_commit(rec, mb);
}
__global__ __launch_bounds__(256) void glu_bwd4_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                       long rows, int C4, float* __restrict__ dx,
                                                       float* __restrict__ rec) {
  const long total = rows * C4;
  unsigned mb = 0u;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / C4;
    const long c = i - r * C4;
    const f32x4 a = *reinterpret_cast<const f32x4*>(x + 4 * (r * 2 * C4 + c));
    const f32x4 l = *reinterpret_cast<const f32x4*>(x + 4 * (r * 2 * C4 + C4 + c));
    const f32x4 d = *reinterpret_cast<const f32x4*>(dy + 4 * i);
    f32x4 da, dl;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float s = sigmoidf_(l[k]);
      da[k] = d[k] * s;
      dl[k] = d[k] * a[k] * s * (1.f - s);
    }
    *reinterpret_cast<f32x4*>(dx + 4 * (r * 2 * C4 + c)) = da;
    *reinterpret_cast<f32x4*>(dx + 4 * (r * 2 * C4 + C4 + c)) = dl;
    mb = amax_bits4(dl, amax_bits4(da, mb));
  }
  if (rec) amax_commit(rec, mb);
}
// GLU backward that also leaves the column sums of what it writes (the bias gradient of the convolution in front of the
// GLU: otherwise one more read of dx, 268 MB for the generator's last gated layer).  Same work split as
// colreduce4_kernel: a block covers 64 of the C columns of dy (their a and gate halves of x / dx) and a chunk of rows,
// 16 row lanes per block; partial[chunk][2 C] is summed by colreduce_finish_kernel.
__global__ __launch_bounds__(256) void glu_bwd4_colsum_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                              long rows, int C, float* __restrict__ dx,
                                                              float* __restrict__ rec, float* __restrict__ partial) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cq = lane & 15, rsub = (lane >> 4) + 4 * wave;
  const int c = blockIdx.x * 64 + 4 * cq;
  const int chunk = blockIdx.y;
  const long per = ceil_div_l(rows, (long)gridDim.y);
  const long r0 = chunk * per;
  long r1 = r0 + per;
  if (r1 > rows) r1 = rows;
  f32x4 sa = {0.f, 0.f, 0.f, 0.f}, sl = {0.f, 0.f, 0.f, 0.f};
  unsigned mb = 0u;
  if (c < C) {
    for (long r = r0 + rsub; r < r1; r += 16) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(x + r * 2 * C + c);
      const f32x4 l = *reinterpret_cast<const f32x4*>(x + r * 2 * C + C + c);
      const f32x4 d = *reinterpret_cast<const f32x4*>(dy + r * C + c);
      f32x4 da, dl;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float sg = sigmoidf_(l[k]);
        da[k] = d[k] * sg;
        dl[k] = d[k] * a[k] * sg * (1.f - sg);
      }
      *reinterpret_cast<f32x4*>(dx + r * 2 * C + c) = da;
      *reinterpret_cast<f32x4*>(dx + r * 2 * C + C + c) = dl;
      sa += da;
      sl += dl;
      mb = amax_bits4(dl, amax_bits4(da, mb));
    }
  }
  __shared__ f32x4 red[2][16][17];
  red[0][rsub][cq] = sa;
  red[1][rsub][cq] = sl;
  __syncthreads();
  if (threadIdx.x < 32) {
    const int half = threadIdx.x >> 4, q = threadIdx.x & 15;
    const int cc = blockIdx.x * 64 + 4 * q;
    if (cc < C) {
      f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 16; ++k) t += red[half][k][q];
      *reinterpret_cast<f32x4*>(partial + (long)chunk * 2 * C + half * C + cc) = t;
    }
  }
  if (rec) amax_commit(rec, mb);
}
__global__ void tanh_fwd_kernel(const float* __restrict__ x, long n, float* __restrict__ y) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = tanhf(x[i]);
}
__global__ void tanh_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, long n,
                                float* __restrict__ dx) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    dx[i] = dy[i] * (1.f - y[i] * y[i]);
}

// ---- feature head ------------------------------------------------------------------------------
__device__ __forceinline__ double block_sum_d(double v, double* red) {
  v = wave_sum_d(v);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  double s = 0;
  for (int k = 0; k < (int)(blockDim.x >> 6); ++k) s += red[k];
  return s;
}

// one block per sample: norm = sqrt(sum x^2) (= ||[relu(x), relu(-x)]||); f = crelu(x)/norm.
// Round 3: float4 walks with the whole block's loads in flight (the scalar one-element-per-thread loops of rounds 1-2
// reached 1.4 TB/s: a sample is only 64 KB, so what counts is how many bytes a block has outstanding); C % 4 == 0
// and 16-byte aligned tensors, else the scalar kernels below.
template <int THREADS>
__global__ __launch_bounds__(THREADS) void feature_head_fwd4_kernel(const float* __restrict__ x, int HW, int C4,
                                                                    float* __restrict__ f, float* __restrict__ norm) {
  __shared__ double red[THREADS / 64];
  const int n = blockIdx.x;
  const long per4 = (long)HW * C4;
  const f32x4* xp = reinterpret_cast<const f32x4*>(x) + n * per4;
  double s = 0;
  for (long i = threadIdx.x; i < per4; i += THREADS) {
    const f32x4 v = xp[i];
    s += (double)v[0] * v[0] + (double)v[1] * v[1] + (double)v[2] * v[2] + (double)v[3] * v[3];
  }
  s = block_sum_d(s, red);
  const float nrm = (float)sqrt(s);
  if (threadIdx.x == 0) norm[n] = nrm;
  f32x4* fp = reinterpret_cast<f32x4*>(f) + n * per4 * 2;
  for (long i = threadIdx.x; i < per4; i += THREADS) {
    const long p = i / C4;
    const long c = i - p * C4;
    const f32x4 v = xp[i];          // (second read: L2)
    f32x4 a, b;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      a[k] = fmaxf(v[k], 0.f) / nrm;        // models/dcgan.py:16,19 (no epsilon)
      b[k] = fmaxf(-v[k], 0.f) / nrm;
    }
    fp[p * 2 * C4 + c] = a;
    fp[p * 2 * C4 + C4 + c] = b;
  }
}
// du = (df - f (f.df)) / norm ;  dx = du[+] * [x>0] - du[-] * [x<0]
template <int THREADS>
__global__ __launch_bounds__(THREADS) void feature_head_bwd4_kernel(const float* __restrict__ x, const float* __restrict__ f,
                                                                    const float* __restrict__ norm, const float* __restrict__ df,
                                                                    int HW, int C4, float* __restrict__ dx, float* __restrict__ rec) {
  __shared__ double red[THREADS / 64];
  const int n = blockIdx.x;
  const long per4 = (long)HW * C4;
  const f32x4* fp = reinterpret_cast<const f32x4*>(f) + n * per4 * 2;
  const f32x4* dfp = reinterpret_cast<const f32x4*>(df) + n * per4 * 2;
  double s = 0;
  for (long i = threadIdx.x; i < 2 * per4; i += THREADS) {
    const f32x4 a = fp[i], b = dfp[i];
    s += (double)a[0] * b[0] + (double)a[1] * b[1] + (double)a[2] * b[2] + (double)a[3] * b[3];
  }
  s = block_sum_d(s, red);
  const float dot = (float)s, inv = 1.f / norm[n];
  const f32x4* xp = reinterpret_cast<const f32x4*>(x) + n * per4;
  f32x4* dxp = reinterpret_cast<f32x4*>(dx) + n * per4;
  unsigned mb = 0u;
  for (long i = threadIdx.x; i < per4; i += THREADS) {
    const long p = i / C4;
    const long c = i - p * C4;
    const f32x4 v = xp[i];
    const long ip = p * 2 * C4 + c, in = ip + C4;
    const f32x4 fpv = fp[ip], fnv = fp[in], dpv = dfp[ip], dnv = dfp[in];
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float dup = (dpv[k] - fpv[k] * dot) * inv;
      const float dun = (dnv[k] - fnv[k] * dot) * inv;
      o[k] = v[k] > 0.f ? dup : (v[k] < 0.f ? -dun : 0.f);
    }
    dxp[i] = o;
    mb = amax_bits4(o, mb);
  }
  if (rec) amax_commit(rec, mb);
}

// one block per sample: norm = sqrt(sum x^2) (= ||[relu(x), relu(-x)]||); f = crelu(x)/norm
__global__ __launch_bounds__(256) void feature_head_fwd_kernel(const float* __restrict__ x, int HW,
                                                               int C, float* __restrict__ f,
                                                               float* __restrict__ norm) {
  __shared__ double red[4];
  const int n = blockIdx.x;
  const long per = (long)HW * C;
  const float* xp = x + n * per;
  double s = 0;
  for (long i = threadIdx.x; i < per; i += blockDim.x) s += (double)xp[i] * (double)xp[i];
  s = block_sum_d(s, red);
  const float nrm = (float)sqrt(s);
  if (threadIdx.x == 0) norm[n] = nrm;
  float* fp = f + n * per * 2;
  for (long i = threadIdx.x; i < per; i += blockDim.x) {
    const long p = i / C;
    const int c = (int)(i - p * C);
    const float v = xp[i];
    fp[p * 2 * C + c] = fmaxf(v, 0.f) / nrm;        // models/dcgan.py:16,19 (no epsilon)
    fp[p * 2 * C + C + c] = fmaxf(-v, 0.f) / nrm;
  }
}
// du = (df - f (f.df)) / norm ;  dx = du[+] * [x>0] - du[-] * [x<0]
__global__ __launch_bounds__(256) void feature_head_bwd_kernel(const float* __restrict__ x,
                                                               const float* __restrict__ f,
                                                               const float* __restrict__ norm,
                                                               const float* __restrict__ df, int HW,
                                                               int C, float* __restrict__ dx,
                                                               float* __restrict__ rec) {
  __shared__ double red[4];
  const int n = blockIdx.x;
  const long per = (long)HW * C;
  const float* fp = f + n * per * 2;
  const float* dfp = df + n * per * 2;
  double s = 0;
  for (long i = threadIdx.x; i < 2 * per; i += blockDim.x) s += (double)fp[i] * (double)dfp[i];
  s = block_sum_d(s, red);
  const float dot = (float)s, inv = 1.f / norm[n];
  const float* xp = x + n * per;
  float* dxp = dx + n * per;
  unsigned mb = 0u;
  for (long i = threadIdx.x; i < per; i += blockDim.x) {
    const long p = i / C;
    const int c = (int)(i - p * C);
    const float v = xp[i];
    const long ip = p * 2 * C + c, in = ip + C;
    const float dup = (dfp[ip] - fp[ip] * dot) * inv;
    const float dun = (dfp[in] - fp[in] * dot) * inv;
    const float o = v > 0.f ? dup : (v < 0.f ? -dun : 0.f);
    dxp[i] = o;
    const unsigned b = amax_bits(o);
    mb = b > mb ? b : mb;
  }
  if (rec) amax_commit(rec, mb);
}

// ---- optimisers / EMA ----------------------------------------------------------------------------
// one element of the reference's Adam (nn.py:61-69): returns the updated parameter.  Shared by both kernels so that
// they round identically (fp contraction is decided per expression tree).
__device__ __forceinline__ float adam_elem(float p, float gi, float* __restrict__ v, float* __restrict__ mg, long j, float lr,
                                           float mom1, float om1, float mom2, float om2, float c1, float c2) {
  // no fused multiply-adds here: which product of `a * b + c * d` keeps its rounding is the compiler's choice per
  // kernel, and the two kernels (and the reference's TensorFlow CPU graph, which has none) must agree bit for bit
#pragma clang fp contract(off)
  float vhat;
  if (mom1 > 0.f) {
    const float vt = mom1 * v[j] + om1 * gi;            // nn.py:61
    v[j] = vt;
    vhat = vt / c1;                                     // nn.py:62
  } else {
    vhat = gi;
  }
  const float mgt = mom2 * mg[j] + om2 * gi * gi;           // nn.py:66
  mg[j] = mgt;
  const float mghat = mgt / c2;                              // nn.py:67
  const float step = lr * (vhat / sqrtf(mghat + 1e-8f));     // nn.py:68-69 (eps inside sqrt)
  return p - step;
}
__device__ __forceinline__ float ema_elem(float sh, float p, float decay, float omd) {
#pragma clang fp contract(off)
  return decay * sh + omd * p;
}
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ v,
                            float* __restrict__ mg, long n, float lr, float mom1, float om1,
                            float mom2, float om2, float c1, float c2, const float* __restrict__ coef) {
  if (coef) {      // (see adam_gather_kernel)
    c1 = coef[0];
    c2 = coef[1];
  }
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    p[i] = adam_elem(p[i], g[i], v, mg, i, lr, mom1, om1, mom2, om2, c1, c2);
}
__global__ void adamax_kernel(float* __restrict__ p, const float* __restrict__ g,
                              float* __restrict__ v, float* __restrict__ mg, long n, float lr,
                              float mom1, float om1, float mom2) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float gi = g[i];
    float vt = gi;
    if (mom1 > 0.f) {
      vt = mom1 * v[i] + om1 * gi;  // nn.py:39
      v[i] = vt;
    }
    const float mgt = fmaxf(mom2 * mg[i] + 1e-8f, fabsf(gi));  // nn.py:43
    mg[i] = mgt;
    p[i] -= lr * (vt / mgt);                                    // nn.py:44-45
  }
}
__global__ void nesterov_kernel(float* __restrict__ p, const float* __restrict__ g,
                                float* __restrict__ v, long n, float lr, float mom1, float opm) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float vo = v[i];
    const float vn = mom1 * vo - lr * g[i];            // nn.py:83
    p[i] = p[i] - mom1 * vo + opm * vn;                // nn.py:84
    v[i] = vn;
  }
}
// Adam over a flat parameter buffer whose gradient arrives as one tensor per variable (what autograd returns): blockIdx.y
// = variable, so no thread ever searches for its segment; same arithmetic, element by element, as adam_kernel.  With
// `sh` the EMA of the updated parameters (train.py:63-64,223: shadow <- decay * shadow + (1 - decay) * p) rides along:
// one read of p less than a separate pass, one launch less.
// ---- batched strided 2-D copies (otgan_copy2d_batched_f32) ----
struct Copy2dSegs {
  const float* src[OTGAN_COPY2D_MAX_SEGMENTS];
  float* dst[OTGAN_COPY2D_MAX_SEGMENTS];
  int rows[OTGAN_COPY2D_MAX_SEGMENTS], cols[OTGAN_COPY2D_MAX_SEGMENTS];
  long sld[OTGAN_COPY2D_MAX_SEGMENTS], dld[OTGAN_COPY2D_MAX_SEGMENTS];
};
__global__ __launch_bounds__(256) void copy2d_batched_kernel(Copy2dSegs g) {
  const int sg = blockIdx.y;
  const int rows = g.rows[sg], cols = g.cols[sg];
  const float* __restrict__ src = g.src[sg];
  float* __restrict__ dst = g.dst[sg];
  const long sld = g.sld[sg], dld = g.dld[sg];
  const long total = (long)rows * cols;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / cols, c = i - r * cols;
    dst[r * dld + c] = src[r * sld + c];
  }
}
extern "C" int otgan_copy2d_batched_f32(const float* const* src, float* const* dst, const int* rows, const int* cols,
                                        const long* src_ld, const long* dst_ld, int nseg, void* stream) {
  OTGAN_CHECK_ARG(src && dst && rows && cols && src_ld && dst_ld && nseg > 0 && nseg <= OTGAN_COPY2D_MAX_SEGMENTS, "bad arguments");
  Copy2dSegs g;
  long most = 0;
  for (int i = 0; i < nseg; ++i) {
    OTGAN_CHECK_ARG(src[i] && dst[i] && rows[i] > 0 && cols[i] > 0 && src_ld[i] >= cols[i] && dst_ld[i] >= cols[i], "bad segment");
    g.src[i] = src[i]; g.dst[i] = dst[i]; g.rows[i] = rows[i]; g.cols[i] = cols[i]; g.sld[i] = src_ld[i]; g.dld[i] = dst_ld[i];
    const long t = (long)rows[i] * cols[i];
    most = t > most ? t : most;
  }
  long bx = (most + 256 * 4 - 1) / (256 * 4);
  if (bx > 256) bx = 256;
  hipLaunchKernelGGL(copy2d_batched_kernel, dim3((unsigned)bx, (unsigned)nseg), dim3(256), 0, (hipStream_t)stream, g);
  OTGAN_CHECK_LAUNCH("copy2d_batched");
  return OTGAN_OK;
}

// ---- batched 3-D gathers (otgan_gather3d_batched_f32) ----
struct Gather3dSegs {
  const float* src[OTGAN_GATHER3D_MAX_SEGMENTS];
  float* dst[OTGAN_GATHER3D_MAX_SEGMENTS];
  int n0[OTGAN_GATHER3D_MAX_SEGMENTS];
  long ss0[OTGAN_GATHER3D_MAX_SEGMENTS], ss1[OTGAN_GATHER3D_MAX_SEGMENTS];
  long ds0[OTGAN_GATHER3D_MAX_SEGMENTS], ds1[OTGAN_GATHER3D_MAX_SEGMENTS];
  int n1, n2, base1;
  const int* map1;
  float* amax;      // zeroed amax record (common.h: amax_commit) of the values written, or null
};
__global__ __launch_bounds__(256) void gather3d_batched_kernel(Gather3dSegs g) {
  const int sg = blockIdx.y;
  const float* __restrict__ src = g.src[sg];
  float* __restrict__ dst = g.dst[sg];
  const long ss0 = g.ss0[sg], ss1 = g.ss1[sg], ds0 = g.ds0[sg], ds1 = g.ds1[sg];
  const int n1 = g.n1, n2 = g.n2;
  const long total = (long)g.n0[sg] * n1 * n2;
  unsigned mb = 0u;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int i2 = (int)(i % n2);
    const long r = i / n2;
    const int i1 = (int)(r % n1);
    const long i0 = r / n1;
    const int j1 = g.base1 + (g.map1 ? g.map1[i1] : i1);
    const float v = src[i0 * ss0 + j1 * ss1 + i2];
    const unsigned b = amax_bits(v);
    mb = b > mb ? b : mb;
    dst[i0 * ds0 + i1 * ds1 + i2] = v;
  }
  if (g.amax) amax_commit(g.amax, mb);     // (every thread of the workgroup arrives here: the loop has no early exit)
}
extern "C" int otgan_gather3d_batched_f32(const float* const* src, float* const* dst, const int* n0, int n1, int n2,
                                          const long* src_s0, const long* src_s1, const long* dst_s0, const long* dst_s1,
                                          int base1, const int* map1_dev, float* amax_out, int nseg, void* stream) {
  OTGAN_CHECK_ARG(src && dst && n0 && src_s0 && src_s1 && dst_s0 && dst_s1 && n1 > 0 && n2 > 0 && base1 >= 0 && nseg > 0 &&
                  nseg <= OTGAN_GATHER3D_MAX_SEGMENTS, "bad arguments");
  Gather3dSegs g;
  long most = 0;
  for (int i = 0; i < nseg; ++i) {
    OTGAN_CHECK_ARG(src[i] && dst[i] && n0[i] > 0, "bad segment");
    g.src[i] = src[i]; g.dst[i] = dst[i]; g.n0[i] = n0[i];
    g.ss0[i] = src_s0[i]; g.ss1[i] = src_s1[i]; g.ds0[i] = dst_s0[i]; g.ds1[i] = dst_s1[i];
    const long t = (long)n0[i] * n1 * n2;
    most = t > most ? t : most;
  }
  g.n1 = n1; g.n2 = n2; g.base1 = base1; g.map1 = map1_dev; g.amax = amax_out;
  long bx = (most + 256 * 4 - 1) / (256 * 4);
  if (bx > 256) bx = 256;
  hipLaunchKernelGGL(gather3d_batched_kernel, dim3((unsigned)bx, (unsigned)nseg), dim3(256), 0, (hipStream_t)stream, g);
  OTGAN_CHECK_LAUNCH("gather3d_batched");
  return OTGAN_OK;
}

struct AdamSegs {
  const float* g[OTGAN_ADAM_MAX_SEGMENTS];
  long off[OTGAN_ADAM_MAX_SEGMENTS + 1];
};
__global__ void adam_gather_kernel(float* __restrict__ p, AdamSegs segs, float* __restrict__ v, float* __restrict__ mg,
                                   float lr, float mom1, float om1, float mom2, float om2, float c1, float c2,
                                   float* __restrict__ sh, float decay, float omd, const float* __restrict__ coef) {
  if (coef) {      // bias corrections from device memory (a captured step: the launch is replayed with a new t; trainer.py)
    c1 = coef[0];
    c2 = coef[1];
  }
  const int seg = blockIdx.y;
  const long base = segs.off[seg], n = segs.off[seg + 1] - base;
  const float* __restrict__ g = segs.g[seg];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long j = base + i;
    const float pn = adam_elem(p[j], g[i], v, mg, j, lr, mom1, om1, mom2, om2, c1, c2);
    p[j] = pn;
    if (sh) sh[j] = ema_elem(sh[j], pn, decay, omd);
  }
}
__global__ void ema_kernel(float* __restrict__ sh, const float* __restrict__ p, long n, float decay,
                           float omd) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    sh[i] = ema_elem(sh[i], p[i], decay, omd);
}

}  // namespace

extern "C" {

int otgan_weightnorm_fwd_amax_f32(const float* V, const float* g, int K, int Cout, float* w, float* wT,
                                  float* inv_norm, float* w_amax, void* stream) {
  OTGAN_CHECK_ARG(V && g && w && inv_norm && K > 0 && Cout > 0, "bad arguments");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(OTGAN_PROF_POINTWISE, 0.0, 4.0 * 3 * (double)K * Cout, s);
  // the partial sums live in the (not yet written) w buffer: nchunk <= ceil(K/64) <= K rows.
  float* partial = w;
  int nchunk = 0;
  int rc = colreduce<1>(V, nullptr, K, Cout, Cout, partial, &nchunk, s);
  if (rc) return rc;
  hipLaunchKernelGGL(colreduce_finish_kernel<1>, dim3(ceil_div(Cout, 64)), dim3(256), 0, s, partial,
                     nchunk, Cout, inv_norm);
  dim3 grid(ceil_div(Cout, 32), ceil_div(K, 32));
  hipLaunchKernelGGL(weightnorm_apply_kernel, grid, dim3(256), 0, s, V, g, inv_norm, K, Cout, w, wT, w_amax);
  OTGAN_CHECK_LAUNCH("weightnorm fwd");
  return OTGAN_OK;
}
int otgan_weightnorm_fwd_f32(const float* V, const float* g, int K, int Cout, float* w, float* wT, float* inv_norm,
                             void* stream) {
  return otgan_weightnorm_fwd_amax_f32(V, g, K, Cout, w, wT, inv_norm, nullptr, stream);
}

int otgan_weightnorm_bwd_f32(const float* V, const float* g, const float* inv_norm, const float* dw,
                             int K, int Cout, float* dV, float* dg, float* scratch, void* stream) {
  OTGAN_CHECK_ARG(V && g && inv_norm && dw && dV && dg && scratch && K > 0 && Cout > 0, "bad arguments");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(OTGAN_PROF_POINTWISE, 0.0, 4.0 * 5 * (double)K * Cout, s);
  float* partial = dV;  // dV is written only after the dots are final
  int nchunk = 0;
  int rc = colreduce<2>(dw, V, K, Cout, Cout, partial, &nchunk, s);
  if (rc) return rc;
  hipLaunchKernelGGL(wn_dot_finish_kernel, dim3(ceil_div(Cout, 64)), dim3(256), 0, s, partial, nchunk, Cout, inv_norm, scratch, dg);
  const long total = (long)K * Cout;
  hipLaunchKernelGGL(weightnorm_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, s, V, g, inv_norm, dw,
                     scratch, total, Cout, dV);
  OTGAN_CHECK_LAUNCH("weightnorm bwd");
  return OTGAN_OK;
}

int otgan_weightnorm_fwd_batched16_f32(const otgan_wn_fwd_layer* layers, int n_layers, void* stream) {
  OTGAN_CHECK_ARG(layers && n_layers > 0, "bad arguments");
  hipStream_t s = (hipStream_t)stream;
  double bytes = 0.0;
  for (int i = 0; i < n_layers; ++i) {
    const otgan_wn_fwd_layer& l = layers[i];
    OTGAN_CHECK_ARG(l.V && l.g && l.w && l.inv && l.K > 0, "layer %d: null pointer or empty", i);
    OTGAN_CHECK_ARG(aligned16(l.V) && aligned16(l.g) && aligned16(l.w) && aligned16(l.inv), "layer %d: 16-byte alignment", i);
    bytes += 4.0 * 4 * (double)l.K * 16;
  }
  ProfScope ps(OTGAN_PROF_POINTWISE, 0.0, bytes, s);
  for (int i0 = 0; i0 < n_layers; i0 += OTGAN_WN_MAX_LAYERS) {
    const int n = n_layers - i0 < OTGAN_WN_MAX_LAYERS ? n_layers - i0 : OTGAN_WN_MAX_LAYERS;
    WnFwdArgs a;
    memset(&a, 0, sizeof(a));
    for (int i = 0; i < n; ++i) a.l[i] = layers[i0 + i];
    hipLaunchKernelGGL(wn_fwd_batched_kernel, dim3(n), dim3(kWnThreads), 0, s, a);
  }
  OTGAN_CHECK_LAUNCH("weightnorm fwd (batched)");
  return OTGAN_OK;
}

int otgan_weightnorm_bwd_batched16_f32(const otgan_wn_bwd_layer* layers, int n_layers, void* stream) {
  OTGAN_CHECK_ARG(layers && n_layers > 0, "bad arguments");
  hipStream_t s = (hipStream_t)stream;
  double bytes = 0.0;
  for (int i = 0; i < n_layers; ++i) {
    const otgan_wn_bwd_layer& l = layers[i];
    OTGAN_CHECK_ARG(l.V && l.g && l.inv && l.dV && l.dg && l.Ceff > 0 && l.taps > 0, "layer %d: null pointer or empty", i);
    OTGAN_CHECK_ARG(aligned16(l.V) && aligned16(l.g) && aligned16(l.inv) && aligned16(l.dV) && aligned16(l.dg), "layer %d: 16-byte alignment", i);
    int rows = 0;
    for (int p = 0; p < 3; ++p) {
      const otgan_wn_part& q = l.part[p];
      OTGAN_CHECK_ARG(q.nrows >= 0 && (q.nrows == 0 || (q.p && aligned16(q.p) && q.rstride >= 16 && q.rstride % 4 == 0)),
                      "layer %d part %d: pointer / stride", i, p);
      rows += q.nrows;
    }
    OTGAN_CHECK_ARG(rows == l.Ceff, "layer %d: the parts hold %d rows per tap, Ceff is %d", i, rows, l.Ceff);
    bytes += 4.0 * 5 * (double)l.taps * l.Ceff * 16;
  }
  ProfScope ps(OTGAN_PROF_POINTWISE, 0.0, bytes, s);
  for (int i0 = 0; i0 < n_layers; i0 += OTGAN_WN_MAX_LAYERS) {
    const int n = n_layers - i0 < OTGAN_WN_MAX_LAYERS ? n_layers - i0 : OTGAN_WN_MAX_LAYERS;
    WnBwdArgs a;
    memset(&a, 0, sizeof(a));
    for (int i = 0; i < n; ++i) a.l[i] = layers[i0 + i];
    hipLaunchKernelGGL(wn_bwd_batched_kernel, dim3(n), dim3(kWnThreads), 0, s, a);
  }
  OTGAN_CHECK_LAUNCH("weightnorm bwd (batched)");
  return OTGAN_OK;
}

int otgan_colsum_f32(const float* a, long rows, int cols, long lda, float* out, float* scratch,
                     void* stream) {
  OTGAN_CHECK_ARG(a && out && scratch && rows > 0 && cols > 0 && lda >= cols, "bad arguments");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(OTGAN_PROF_POINTWISE, 0.0, 4.0 * (double)rows * cols, s);
  int nchunk = 0;
  int rc = colreduce<0>(a, nullptr, rows, cols, lda, scratch, &nchunk, s);
  if (rc) return rc;
  hipLaunchKernelGGL(colreduce_finish_kernel<0>, dim3(ceil_div(cols, 64)), dim3(256), 0, s, scratch,
                     nchunk, cols, out);
  OTGAN_CHECK_LAUNCH("colsum");
  return OTGAN_OK;
}


int otgan_glu_fwd_amax_f32(const float* x, long rows, int C, float* y, float* y_amax, void* stream) {
  OTGAN_CHECK_ARG(x && y && rows > 0 && C > 0, "bad arguments");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(OTGAN_PROF_POINTWISE, 0.0, 4.0 * 3 * (double)rows * C, s);
  if (C % 4 == 0 && aligned16(x) && aligned16(y)) {
    hipLaunchKernelGGL(glu_fwd4_kernel, dim3(grid_for(rows * C, 8)), dim3(256), 0, s, x, rows, C / 4, y, y_amax);
  } else {
    OTGAN_CHECK_ARG(!y_amax, "glu: an amax record needs C %% 4 == 0 and 16-byte aligned tensors");
    hipLaunchKernelGGL(glu_fwd_kernel, dim3(grid_for(rows * C)), dim3(256), 0, s, x, rows, C, y);
  }
  OTGAN_CHECK_LAUNCH("glu fwd");
  return OTGAN_OK;
}
int otgan_glu_fwd_f32(const float* x, long rows, int C, float* y, void* stream) {
  return otgan_glu_fwd_amax_f32(x, rows, C, y, nullptr, stream);
}
int otgan_glu_bwd_amax_f32(const float* x, const float* dy, long rows, int C, float* dx, float* dx_amax, void* stream) {
  OTGAN_CHECK_ARG(x && dy && dx && rows > 0 && C > 0, "bad arguments");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(OTGAN_PROF_POINTWISE, 0.0, 4.0 * 5 * (double)rows * C, s);
  if (C % 4 == 0 && aligned16(x) && aligned16(dy) && aligned16(dx)) {
    hipLaunchKernelGGL(glu_bwd4_kernel, dim3(grid_for(rows * C, 8)), dim3(256), 0, s, x, dy, rows, C / 4, dx, dx_amax);
  } else {
    OTGAN_CHECK_ARG(!dx_amax, "glu: an amax record needs C %% 4 == 0 and 16-byte aligned tensors");
    hipLaunchKernelGGL(glu_bwd_kernel, dim3(grid_for(rows * C)), dim3(256), 0, s, x, dy, rows, C, dx);
  }
  OTGAN_CHECK_LAUNCH("glu bwd");
  return OTGAN_OK;
}
int otgan_glu_bwd_colsum_f32(const float* x, const float* dy, long rows, int C, float* dx, float* dx_amax, float* colsum,
                             float* scratch, void* stream) {
  OTGAN_CHECK_ARG(x && dy && dx && colsum && scratch && rows > 0 && C > 0 && C % 4 == 0 && aligned16(x) && aligned16(dy) &&
                      aligned16(dx) && aligned16(scratch),
                  "glu bwd with column sums: C %% 4 == 0 and 16-byte aligned tensors");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(OTGAN_PROF_POINTWISE, 0.0, 4.0 * 5 * (double)rows * C, s);
  const int colblocks = ceil_div(C, 64);
  long nchunk = ceil_div(1024, colblocks);              // aim for >= 1024 workgroups
  if (nchunk > kChunks) nchunk = kChunks;
  if (nchunk > ceil_div_l(rows, 64)) nchunk = ceil_div_l(rows, 64);
  if (nchunk < 1) nchunk = 1;
  hipLaunchKernelGGL(glu_bwd4_colsum_kernel, dim3(colblocks, (int)nchunk), dim3(256), 0, s, x, dy, rows, C, dx, dx_amax, scratch);
  hipLaunchKernelGGL(colreduce_finish_kernel<0>, dim3(ceil_div(2 * C, 64)), dim3(256), 0, s, scratch, (int)nchunk, 2 * C, colsum);
  OTGAN_CHECK_LAUNCH("glu bwd (+ column sums)");
  return OTGAN_OK;
}
int otgan_glu_bwd_f32(const float* x, const float* dy, long rows, int C, float* dx, void* stream) {
  return otgan_glu_bwd_amax_f32(x, dy, rows, C, dx, nullptr, stream);
}
int otgan_tanh_fwd_f32(const float* x, long n, float* y, void* stream) {
  OTGAN_CHECK_ARG(x && y && n > 0, "bad arguments");
  hipLaunchKernelGGL(tanh_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, n, y);
  OTGAN_CHECK_LAUNCH("tanh fwd");
  return OTGAN_OK;
}
int otgan_tanh_bwd_f32(const float* y, const float* dy, long n, float* dx, void* stream) {
  OTGAN_CHECK_ARG(y && dy && dx && n > 0, "bad arguments");
  hipLaunchKernelGGL(tanh_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, y, dy, n, dx);
  OTGAN_CHECK_LAUNCH("tanh bwd");
  return OTGAN_OK;
}

int otgan_feature_head_fwd_f32(const float* x, int N, int HW, int C, float* f, float* norm,
                               void* stream) {
  OTGAN_CHECK_ARG(x && f && norm && N > 0 && HW > 0 && C > 0, "bad arguments");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(OTGAN_PROF_POINTWISE, 0.0, 4.0 * 4 * (double)N * HW * C, s);
  if (C % 4 == 0 && aligned16(x) && aligned16(f))
    hipLaunchKernelGGL(feature_head_fwd4_kernel<1024>, dim3(N), dim3(1024), 0, s, x, HW, C / 4, f, norm);
  else
    hipLaunchKernelGGL(feature_head_fwd_kernel, dim3(N), dim3(256), 0, s, x, HW, C, f, norm);
  OTGAN_CHECK_LAUNCH("feature head fwd");
  return OTGAN_OK;
}
int otgan_feature_head_bwd_amax_f32(const float* x, const float* f, const float* norm, const float* df,
                                    int N, int HW, int C, float* dx, float* dx_amax, void* stream) {
  OTGAN_CHECK_ARG(x && f && norm && df && dx && N > 0 && HW > 0 && C > 0, "bad arguments");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(OTGAN_PROF_POINTWISE, 0.0, 4.0 * 10 * (double)N * HW * C, s);
  if (C % 4 == 0 && aligned16(x) && aligned16(f) && aligned16(df) && aligned16(dx))
    hipLaunchKernelGGL(feature_head_bwd4_kernel<1024>, dim3(N), dim3(1024), 0, s, x, f, norm, df, HW, C / 4, dx, dx_amax);
  else
    hipLaunchKernelGGL(feature_head_bwd_kernel, dim3(N), dim3(256), 0, s, x, f, norm, df, HW, C, dx, dx_amax);
  OTGAN_CHECK_LAUNCH("feature head bwd");
  return OTGAN_OK;
}
int otgan_feature_head_bwd_f32(const float* x, const float* f, const float* norm, const float* df,
                               int N, int HW, int C, float* dx, void* stream) {
  return otgan_feature_head_bwd_amax_f32(x, f, norm, df, N, HW, C, dx, nullptr, stream);
}

void otgan_adam_coefficients(double mom1, double mom2, double t, float* out2) {
  // nn.py:62,67: `1. - tf.pow(mom, t)` is evaluated IN the fp32 graph (t is a float32 variable)
  out2[0] = 1.f - powf((float)mom1, (float)t);
  out2[1] = 1.f - powf((float)mom2, (float)t);
}
int otgan_adam_step_f32(float* p, const float* grad, float* v, float* mg, long n, double lr,
                        double mom1, double mom2, double t, void* stream) {
  return otgan_adam_step_coef_f32(p, grad, v, mg, n, lr, mom1, mom2, t, nullptr, stream);
}
int otgan_adam_step_coef_f32(float* p, const float* grad, float* v, float* mg, long n, double lr,
                             double mom1, double mom2, double t, const float* coef_dev, void* stream) {
  OTGAN_CHECK_ARG(p && grad && mg && n > 0 && (coef_dev || t >= 1.0) && (mom1 <= 0.0 || v), "bad arguments");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(OTGAN_PROF_POINTWISE, 0.0, 4.0 * 7 * (double)n, s);
  // nn.py:62,67: `1. - tf.pow(mom, t)` is evaluated IN the fp32 graph (t is a float32 variable),
  // whereas `(1. - mom)` (nn.py:61,66) is a Python double folded into an fp32 constant.
  float c[2] = {1.f, 1.f};
  if (!coef_dev) otgan_adam_coefficients(mom1, mom2, t, c);
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, s, p, grad, v, mg, n, (float)lr,
                     (float)mom1, (float)(1.0 - mom1), (float)mom2, (float)(1.0 - mom2), c[0], c[1], coef_dev);
  OTGAN_CHECK_LAUNCH("adam");
  return OTGAN_OK;
}
int otgan_adam_step_gather_f32(float* p, const float* const* grads, const long* offsets, int nseg, float* v, float* mg,
                               double lr, double mom1, double mom2, double t, float* ema_shadow, double ema_decay,
                               void* stream) {
  return otgan_adam_step_gather_coef_f32(p, grads, offsets, nseg, v, mg, lr, mom1, mom2, t, nullptr, ema_shadow, ema_decay, stream);
}
int otgan_adam_step_gather_coef_f32(float* p, const float* const* grads, const long* offsets, int nseg, float* v, float* mg,
                                    double lr, double mom1, double mom2, double t, const float* coef_dev, float* ema_shadow,
                                    double ema_decay, void* stream) {
  OTGAN_CHECK_ARG(p && grads && offsets && mg && nseg > 0 && nseg <= OTGAN_ADAM_MAX_SEGMENTS && (coef_dev || t >= 1.0) && (mom1 <= 0.0 || v),
                  "bad arguments (at most %d segments)", OTGAN_ADAM_MAX_SEGMENTS);
  hipStream_t s = (hipStream_t)stream;
  AdamSegs segs;
  memset(&segs, 0, sizeof(segs));
  long longest = 0;
  for (int i = 0; i < nseg; ++i) {
    OTGAN_CHECK_ARG(grads[i] && offsets[i + 1] > offsets[i], "segment %d: null gradient or empty range", i);
    segs.g[i] = grads[i];
    segs.off[i] = offsets[i];
    if (offsets[i + 1] - offsets[i] > longest) longest = offsets[i + 1] - offsets[i];
  }
  segs.off[nseg] = offsets[nseg];
  const long n = offsets[nseg] - offsets[0];
  ProfScope ps(OTGAN_PROF_POINTWISE, 0.0, 4.0 * (ema_shadow ? 9 : 7) * (double)n, s);
  float c[2] = {1.f, 1.f};
  if (!coef_dev) otgan_adam_coefficients(mom1, mom2, t, c);          // (see otgan_adam_step_f32)
  hipLaunchKernelGGL(adam_gather_kernel, dim3(grid_for(longest), nseg), dim3(256), 0, s, p, segs, v, mg, (float)lr,
                     (float)mom1, (float)(1.0 - mom1), (float)mom2, (float)(1.0 - mom2), c[0], c[1], ema_shadow,
                     (float)ema_decay, (float)(1.0 - ema_decay), coef_dev);
  OTGAN_CHECK_LAUNCH("adam (gathered gradients)");
  return OTGAN_OK;
}
int otgan_adamax_step_f32(float* p, const float* grad, float* v, float* mg, long n, double lr,
                          double mom1, double mom2, void* stream) {
  OTGAN_CHECK_ARG(p && grad && mg && n > 0 && (mom1 <= 0.0 || v), "bad arguments");
  hipLaunchKernelGGL(adamax_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, grad, v,
                     mg, n, (float)lr, (float)mom1, (float)(1.0 - mom1), (float)mom2);
  OTGAN_CHECK_LAUNCH("adamax");
  return OTGAN_OK;
}
int otgan_nesterov_step_f32(float* p, const float* grad, float* v, long n, double lr, double mom1,
                            void* stream) {
  OTGAN_CHECK_ARG(p && grad && v && n > 0, "bad arguments");
  hipLaunchKernelGGL(nesterov_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, grad,
                     v, n, (float)lr, (float)mom1, (float)(1.0 + mom1));
  OTGAN_CHECK_LAUNCH("nesterov");
  return OTGAN_OK;
}
int otgan_ema_update_f32(float* shadow, const float* p, long n, double decay, void* stream) {
  OTGAN_CHECK_ARG(shadow && p && n > 0, "bad arguments");
  hipLaunchKernelGGL(ema_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, shadow, p, n,
                     (float)decay, (float)(1.0 - decay));
  OTGAN_CHECK_LAUNCH("ema");
  return OTGAN_OK;
}

}  // extern "C"
