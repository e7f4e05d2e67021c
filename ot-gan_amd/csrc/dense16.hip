// dense16.hip -- LDS-free MFMA kernels for the DenseNet growth layers (3x3 / stride 1 / 16
// output channels, in-place concatenation buffers).
//
// Why a second conv engine: with 16 output channels an LDS-tiled implicit GEMM re-stages every
// activation through LDS to use it in a single 16-wide MFMA row -- the barrier/LDS machinery
// costs more than the math (measured 25-40 TFLOP/s, and 64-256 workgroups on the 8x8/16x16
// stages).  Here every wave streams its operands straight from global memory into MFMA
// operand registers:
//
//   v_mfma_f32_16x16x4_f32:  A[m = lane%16][k = lane/16],  B[k = lane/16][n = lane%16]
//
// The contraction index may be permuted freely as long as A and B agree, so lane (p, g) takes
// EIGHT consecutive effective channels e0 + 8g .. e0 + 8g + 7 of pixel p (two 16-byte loads;
// the four lane groups of a pixel cover one 128-byte line) and MFMA j of the step contracts
// the channels {e0 + 8g + j}.  The weight operand of lane (n, g) is then the eight consecutive
// floats wT[n][tap*Ceff + e0 + 8g ..] of the transposed weight -- again two 16-byte loads, no
// transposition, no LDS, no barrier; waves are independent and hide latency by occupancy plus
// a one-step register prefetch.
#include "dense16.h"

#include <stdlib.h>

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));

struct FwdArgs {
  const float* x;
  const int32_t* cmap;
  const float* wT;
  const float* bias;
  float* y;
  int M, H, W, logW, ldx, C, Ceff, doubled, K, ldy, coff;
};

template <int ACT>
__device__ __forceinline__ float d16_act(float v) {
  if (ACT == 1) return fmaxf(v, 0.f);
  if (ACT == 2) return v > 0.f ? v : expm1f(v);
  return v;
}

// the eight effective channels a lane owns in one 32-channel chunk
struct Group {
  int e;       // first effective channel (clamped into range)
  int c;       // its source channel
  float sg;    // sign of the whole group (contiguous groups)
  bool valid;  // e < Ceff
  bool contig; // the eight map to eight consecutive source channels with one sign
};

__device__ __forceinline__ Group d16_group(const FwdArgs& a, int chunk, int g) {
  Group r;
  const int e = chunk * 32 + 8 * g;
  r.valid = e < a.Ceff;
  r.e = r.valid ? e : 0;
  if (a.cmap) {
    const i32x4 m0 = *reinterpret_cast<const i32x4*>(a.cmap + r.e);
    const i32x4 m1 = *reinterpret_cast<const i32x4*>(a.cmap + r.e + 4);
    r.c = m0.x & 0x7fffffff;
    r.sg = m0.x < 0 ? -1.f : 1.f;
    r.contig = m0.y == m0.x + 1 && m0.z == m0.x + 2 && m0.w == m0.x + 3 && m1.x == m0.x + 4 &&
               m1.y == m0.x + 5 && m1.z == m0.x + 6 && m1.w == m0.x + 7 && (r.c & 3) == 0;
  } else {
    const bool neg = a.doubled && r.e >= a.C;
    r.c = neg ? r.e - a.C : r.e;
    r.sg = neg ? -1.f : 1.f;
    r.contig = true;
  }
  return r;
}

template <int PT>
struct Pix {
  long off[PT];  // element offset of the pixel in x
  int y[PT], x[PT];
  bool ok[PT];
};

template <int PT>
struct Stage {     // operands of one (chunk, tap) step
  f32x4 A[PT][2];
  f32x4 Wv[2];
  float sg;
};

template <int PT>
__device__ __forceinline__ void d16_load(const FwdArgs& a, const Group& gr, int tap, const Pix<PT>& px,
                                         const float* wrow, Stage<PT>& st) {
  const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  const float* wp = wrow + (long)tap * a.Ceff + gr.e;
  st.Wv[0] = gr.valid ? *reinterpret_cast<const f32x4*>(wp) : zero;
  st.Wv[1] = gr.valid ? *reinterpret_cast<const f32x4*>(wp + 4) : zero;
  st.sg = gr.contig ? gr.sg : 1.f;
  const long shift = (long)(dy * a.W + dx) * a.ldx;
#pragma unroll
  for (int t = 0; t < PT; ++t) {
    const int y2 = px.y[t] + dy, x2 = px.x[t] + dx;
    const bool ok = px.ok[t] && gr.valid && (unsigned)y2 < (unsigned)a.H && (unsigned)x2 < (unsigned)a.W;
    const float* xp = a.x + px.off[t] + shift;
    if (gr.contig) {
      st.A[t][0] = ok ? *reinterpret_cast<const f32x4*>(xp + gr.c) : zero;
      st.A[t][1] = ok ? *reinterpret_cast<const f32x4*>(xp + gr.c + 4) : zero;
    } else {
      // irregular channel map (list elements narrower than 8 channels): per-channel gathers
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int cm = a.cmap[gr.e + j];
        float v = ok ? xp[cm & 0x7fffffff] : 0.f;
        if (cm < 0) v = -v;
        st.A[t][j >> 2][j & 3] = v;
      }
    }
  }
}

template <int PT, int ACT, bool SIGNED>
__device__ __forceinline__ void d16_compute(const Stage<PT>& st, f32x4 (&acc)[PT]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float w = st.Wv[j >> 2][j & 3];
#pragma unroll
    for (int t = 0; t < PT; ++t) {
      float v = st.A[t][j >> 2][j & 3];
      if (SIGNED) v *= st.sg;
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(d16_act<ACT>(v), w, acc[t], 0, 0, 0);
    }
  }
}

// block = 4 independent waves, wave = PT tiles of 16 consecutive pixels x 16 output channels
template <int PT, int ACT, bool SIGNED>
__global__ __launch_bounds__(256) void dense16_fwd_kernel(FwdArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p = lane & 15, g = lane >> 4;
  const int tile0 = (blockIdx.x * 4 + wave) * PT;
  if (tile0 * 16 >= a.M) return;
  Pix<PT> px;
#pragma unroll
  for (int t = 0; t < PT; ++t) {
    const int m = (tile0 + t) * 16 + p;
    px.ok[t] = m < a.M;
    const int mm = px.ok[t] ? m : 0;
    px.x[t] = mm & (a.W - 1);
    px.y[t] = (mm >> a.logW) & (a.H - 1);
    px.off[t] = (long)mm * a.ldx;
  }
  f32x4 acc[PT];
#pragma unroll
  for (int t = 0; t < PT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* wrow = a.wT + (long)p * a.K;
  const int nchunk = (a.Ceff + 31) >> 5;
  const int S = nchunk * 9;

  // load stream position (chunk, tap) runs one step ahead of the compute stream
  int lchunk = 0, ltap = 0;
  Group gr = d16_group(a, 0, g);
  Group grn = d16_group(a, nchunk > 1 ? 1 : 0, g);
  auto advance = [&]() {
    if (++ltap == 9) {
      ltap = 0;
      ++lchunk;
      gr = grn;
      grn = d16_group(a, lchunk + 1 < nchunk ? lchunk + 1 : lchunk, g);
    }
  };
  Stage<PT> s0, s1;
  d16_load<PT>(a, gr, ltap, px, wrow, s0);
  advance();
  for (int s = 0; s < S; s += 2) {
    if (s + 1 < S) {
      d16_load<PT>(a, gr, ltap, px, wrow, s1);
      advance();
    }
    d16_compute<PT, ACT, SIGNED>(s0, acc);
    if (s + 2 < S) {
      d16_load<PT>(a, gr, ltap, px, wrow, s0);
      advance();
    }
    if (s + 1 < S) d16_compute<PT, ACT, SIGNED>(s1, acc);
  }
  // D[m = 4g + r][n = p]
  const float b = a.bias ? a.bias[p] : 0.f;
#pragma unroll
  for (int t = 0; t < PT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long m = (long)(tile0 + t) * 16 + 4 * g + r;
      if (m < a.M) a.y[m * a.ldy + a.coff + p] = acc[t][r] + b;
    }
}

template <int PT>
void launch_fwd_pt(const FwdArgs& a, int act, bool sgn, int blocks, hipStream_t s) {
  const dim3 grid(blocks), blk(256);
  if (act == 1 && sgn) hipLaunchKernelGGL((dense16_fwd_kernel<PT, 1, true>), grid, blk, 0, s, a);
  else if (act == 1) hipLaunchKernelGGL((dense16_fwd_kernel<PT, 1, false>), grid, blk, 0, s, a);
  else if (act == 2 && sgn) hipLaunchKernelGGL((dense16_fwd_kernel<PT, 2, true>), grid, blk, 0, s, a);
  else if (act == 2) hipLaunchKernelGGL((dense16_fwd_kernel<PT, 2, false>), grid, blk, 0, s, a);
  else if (sgn) hipLaunchKernelGGL((dense16_fwd_kernel<PT, 0, true>), grid, blk, 0, s, a);
  else hipLaunchKernelGGL((dense16_fwd_kernel<PT, 0, false>), grid, blk, 0, s, a);
}

int ilog2i(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

}  // namespace

bool dense16_enabled() {
  static const bool on = [] {
    const char* e = getenv("OTGAN_DISABLE_DENSE16");
    return !(e && e[0] == '1');
  }();
  return on;
}

int dense16_fwd(const Dense16Geo& g, const float* x, const float* wT, const float* bias, float* y,
                int ldy, int coff, hipStream_t s) {
  FwdArgs a;
  a.x = x; a.cmap = g.cmap; a.wT = wT; a.bias = bias; a.y = y;
  a.M = g.N * g.H * g.W;
  a.H = g.H; a.W = g.W; a.logW = ilog2i(g.W);
  a.ldx = g.ldx; a.C = g.C; a.Ceff = g.Ceff; a.doubled = g.doubled;
  a.K = 9 * g.Ceff; a.ldy = ldy; a.coff = coff;
  const int tiles = (a.M + 15) / 16;
  const bool sgn = g.doubled || g.cmap != nullptr;
  // enough waves to fill 256 CUs x 4 SIMDs a few times over; fewer, fatter waves when there are plenty
  if (tiles >= 8192) launch_fwd_pt<4>(a, g.act, sgn, (tiles + 15) / 16, s);
  else if (tiles >= 4096) launch_fwd_pt<2>(a, g.act, sgn, (tiles + 7) / 8, s);
  else launch_fwd_pt<1>(a, g.act, sgn, (tiles + 3) / 4, s);
  return OTGAN_OK;
}
