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
  int accumulate;   // y += instead of y = (the growth layers of a split dense block add onto the block-input convolution's result)
  float* amax;      // amax record of the values written (common.h), or null
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
  if (tile0 * 16 >= a.M) {
    if (a.amax) amax_commit(a.amax, 0u);   // (the workgroup's barrier in there counts every wave once)
    return;
  }
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
  unsigned omax = 0u;
#pragma unroll
  for (int t = 0; t < PT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long m = (long)(tile0 + t) * 16 + 4 * g + r;
      if (m < a.M) {
        float* yp = a.y + m * a.ldy + a.coff + p;
        const float o = acc[t][r] + b + (a.accumulate ? *yp : 0.f);
        *yp = o;
        const unsigned ob = amax_bits(o);
        omax = ob > omax ? ob : omax;
      }
    }
  if (a.amax) amax_commit(a.amax, omax);
}


// =======================================================================================
// Forward, version 2: haloed spatial tile in LDS.
//
// The streaming kernel above re-fetches every activation nine times (once per tap) from L2:
// at 8 FLOP per fetched byte that path saturates near 57 TFLOP/s.  Here a block owns TR full
// image rows (64*PT pixels); per 32-channel slice it stages the (TR+2)-row halo tile ONCE
// (1.25x instead of 9x), with the activation and the CReLU sign applied on the way in, and all
// nine taps read their MFMA operands from LDS.
//
// LDS layout: pixel-major, 40 floats per pixel (32 channels + 8 pad): quad s of a pixel holds
// channels 4s..4s+3; lane (p, g) reads quads g and g+4 with two ds_read_b128.  Pixel stride
// 10 quads (2 * odd) plus the one-quad lane-group offset is conflict-free for the hardware's
// b128 lane grouping (brute-forced over all four groups); stores put the 8 quads of a pixel in
// 8 consecutive lanes = one 128-byte row of banks.  K permutation: MFMA (h, i) contracts
// channel 16h + 4g + i in k-slot g, so lane (n, g) takes its weights as the float4s
// wT[n][tap*Ceff + c0 + 16h + 4g ..] straight from global memory.
// =======================================================================================
struct FwdLdsArgs {
  const float* x;
  const int32_t* cmap;
  const float* wT;
  const float* bias;
  float* y;
  int N, H, W, logW, ldx, C, Ceff, doubled, K, ldy, coff;
  int TR, RS;  // tile rows, LDS row stride in pixels
  int accumulate;
  float* amax;  // amax record of the values written (common.h), or null
};

constexpr int kPixQuads = 10;  // LDS quads (16 B) per pixel

struct QuadMap {
  int c;
  float sg;
  bool valid, contig;
  int e;
};

__device__ __forceinline__ QuadMap d16_quad(const FwdLdsArgs& a, int e, i32x4 cm) {
  QuadMap r;
  r.valid = e < a.Ceff;
  r.e = r.valid ? e : 0;
  if (a.cmap) {
    r.c = cm.x & 0x7fffffff;
    r.sg = cm.x < 0 ? -1.f : 1.f;
    r.contig = cm.y == cm.x + 1 && cm.z == cm.x + 2 && cm.w == cm.x + 3 && (r.c & 3) == 0;
  } else {
    const bool neg = a.doubled && r.e >= a.C;
    r.c = neg ? r.e - a.C : r.e;
    r.sg = neg ? -1.f : 1.f;
    r.contig = true;
  }
  return r;
}

template <int PT, int ACT, bool W8>
__global__ __launch_bounds__(256, 2) void dense16_fwd_lds_kernel(FwdLdsArgs a) {
  extern __shared__ f32x4 smem4[];
  constexpr int NITMAX = 2 * PT + 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p = lane & 15, g = lane >> 4;
  const int tiles_per_img = a.H / a.TR;
  const int n = blockIdx.x / tiles_per_img;
  const int r0 = (blockIdx.x - n * tiles_per_img) * a.TR;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  const int lds_quads = (a.TR + 2) * a.RS * kPixQuads;
  for (int i = tid; i < lds_quads; i += 256) smem4[i] = zero;

  // staging: item = (pixel of the (TR+2) x W row band, quad); 8 consecutive lanes = one pixel
  const int slot = tid & 7;
  const int total = (a.TR + 2) * a.W * 8;
  const long img_base = (long)n * a.H * a.W;
  f32x4 R[NITMAX];
  auto stage_load = [&](const QuadMap& q) {
#pragma unroll
    for (int it = 0; it < NITMAX; ++it) {
      const int i = it * 256 + tid;
      const int px = i >> 3;
      const int row = px >> a.logW, col = px & (a.W - 1);
      const int ir = r0 - 1 + row;
      const bool ok = i < total && q.valid && (unsigned)ir < (unsigned)a.H;
      const float* xp = a.x + (img_base + (long)ir * a.W + col) * a.ldx;
      if (q.contig) {
        R[it] = ok ? *reinterpret_cast<const f32x4*>(xp + q.c) : zero;
      } else {
        f32x4 v = zero;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int cm = a.cmap[q.e + j];
          float t = ok ? xp[cm & 0x7fffffff] : 0.f;
          v[j] = cm < 0 ? -t : t;
        }
        R[it] = v;
      }
    }
  };
  auto stage_store = [&](float sg) {
#pragma unroll
    for (int it = 0; it < NITMAX; ++it) {
      const int i = it * 256 + tid;
      if (i < total) {
        const int px = i >> 3;
        const int row = px >> a.logW, col = px & (a.W - 1);
        f32x4 v = R[it];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = d16_act<ACT>(v[j] * sg);
        smem4[(row * a.RS + col + 1) * kPixQuads + slot] = v;
      }
    }
  };
  auto load_cm = [&](int chunk) {
    const int e = chunk * 32 + 4 * slot;
    i32x4 cm = {0, 0, 0, 0};
    if (a.cmap && e < a.Ceff) cm = *reinterpret_cast<const i32x4*>(a.cmap + e);
    return cm;
  };

  // fragment addressing: M-tile = 16 consecutive pixels of the row band
  int ab[PT];
#pragma unroll
  for (int t = 0; t < PT; ++t) {
    const int q0 = (wave * PT + t) * 16;
    int rr, cc;
    if (W8) {
      rr = (q0 >> 3) + (p >> 3);
      cc = p & 7;
    } else {
      rr = q0 >> a.logW;
      cc = (q0 & (a.W - 1)) + p;
    }
    ab[t] = ((rr + 1) * a.RS + cc + 1) * kPixQuads + g;
  }
  f32x4 acc[PT];
#pragma unroll
  for (int t = 0; t < PT; ++t) acc[t] = zero;

  const int nchunk = (a.Ceff + 31) >> 5;
  const float* wlane = a.wT + (long)p * a.K + 4 * g;   // + tap*Ceff + chunk*32 + 16h
  auto load_w = [&](int chunk, int tap, f32x4 (&Wv)[2]) {
    const int e0 = chunk * 32 + 4 * g;
    const float* wp = wlane + (long)tap * a.Ceff + chunk * 32;
    Wv[0] = (e0 < a.Ceff) ? *reinterpret_cast<const f32x4*>(wp) : zero;
    Wv[1] = (e0 + 16 < a.Ceff) ? *reinterpret_cast<const f32x4*>(wp + 16) : zero;
  };

  QuadMap q = d16_quad(a, 4 * slot, load_cm(0));
  i32x4 cm_next = load_cm(1);
  stage_load(q);
  f32x4 Wa[2], Wb[2];
  load_w(0, 0, Wa);
  __syncthreads();  // zero fill complete
  stage_store(q.contig ? q.sg : 1.f);
  __syncthreads();

  for (int c = 0; c < nchunk; ++c) {
    const bool more = c + 1 < nchunk;
    if (more) {
      q = d16_quad(a, (c + 1) * 32 + 4 * slot, cm_next);
      cm_next = load_cm(c + 2);
      stage_load(q);  // lands while the nine taps below run
    }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      f32x4 (&Wc)[2] = (tap & 1) ? Wb : Wa;
      f32x4 (&Wn)[2] = (tap & 1) ? Wa : Wb;
      if (tap < 8) load_w(c, tap + 1, Wn);
      else if (more) load_w(c + 1, 0, Wn);
      const int dy = tap / 3 - 1, dx = tap % 3 - 1;
      const int sh = (dy * a.RS + dx) * kPixQuads;
      f32x4 A[PT][2];
#pragma unroll
      for (int t = 0; t < PT; ++t) {
        A[t][0] = smem4[ab[t] + sh];
        A[t][1] = smem4[ab[t] + sh + 4];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float w = Wc[j >> 2][j & 3];
#pragma unroll
        for (int t = 0; t < PT; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[t][j >> 2][j & 3], w, acc[t], 0, 0, 0);
      }
    }
    if (more) {  // nine taps: the prefetch of (c+1, tap 0) landed in the odd buffer
      Wa[0] = Wb[0];
      Wa[1] = Wb[1];
    }
    __syncthreads();
    if (more) {
      stage_store(q.contig ? q.sg : 1.f);
      __syncthreads();
    }
  }
  const float b = a.bias ? a.bias[p] : 0.f;
  const long m0 = (img_base + (long)r0 * a.W);
  unsigned omax = 0u;
#pragma unroll
  for (int t = 0; t < PT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long m = m0 + (wave * PT + t) * 16 + 4 * g + r;
      float* yp = a.y + m * a.ldy + a.coff + p;
      const float o = acc[t][r] + b + (a.accumulate ? *yp : 0.f);
      *yp = o;
      const unsigned ob = amax_bits(o);
      omax = ob > omax ? ob : omax;
    }
  if (a.amax) amax_commit(a.amax, omax);
}

// =======================================================================================
// Forward, version 3: the same halo tile, on the bf16 matrix pipe with split-precision operands.
// Activations are split into three bf16 pieces (x = hi + mid + lo) on the way into LDS, weights in
// registers after the load; six v_mfma_f32_16x16x32_bf16 (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi,
// mid*mid) replace eight fp32 MFMAs at 2.7x fewer matrix-pipe cycles, products fp32-exact.
// The 32-wide contraction of one MFMA is (2 taps) x (16 effective channels): k-groups 0,1 of the
// lanes take channels 0-7 / 8-15 of tap 2t, k-groups 2,3 the same channels of tap 2t+1 (the fifth
// pair is half empty).  With 16 channels per chunk a pixel is exactly 32 bytes per piece and the
// fragment reads (lane = pixel, 16 bytes at offset 16*(g&1)) are conflict-free without padding.
// =======================================================================================
typedef unsigned short u16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void d16_split2(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  f32x2_t v = {x0, x1};
  const bf16x2_t hb = __builtin_convertvector(v, bf16x2_t);
  v -= __builtin_convertvector(hb, f32x2_t);
  const bf16x2_t mb = __builtin_convertvector(v, bf16x2_t);
  v -= __builtin_convertvector(mb, f32x2_t);
  const bf16x2_t lb = __builtin_convertvector(v, bf16x2_t);
  h = __builtin_bit_cast(unsigned, hb);
  m = __builtin_bit_cast(unsigned, mb);
  l = __builtin_bit_cast(unsigned, lb);
}

template <int PT, int ACT, bool W8>
__global__ __launch_bounds__(256, 2) void dense16_fwd_x3_kernel(FwdLdsArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
  constexpr int NITMAX = PT + 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p = lane & 15, g = lane >> 4;
  const int tiles_per_img = a.H / a.TR;
  const int n = blockIdx.x / tiles_per_img;
  const int r0 = (blockIdx.x - n * tiles_per_img) * a.TR;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  const int TILE = (a.TR + 2) * a.RS * 32;                  // bytes of one piece plane
  for (int i = tid; i < 3 * TILE / 16; i += 256) reinterpret_cast<u32x4*>(smem3)[i] = u32x4{0u, 0u, 0u, 0u};

  // staging: item = (pixel of the (TR+2) x W row band, quad of 4 channels); 4 lanes = one pixel
  const int slot = tid & 3;
  const int total = (a.TR + 2) * a.W * 4;
  const long img_base = (long)n * a.H * a.W;
  f32x4 R[NITMAX];
  auto stage_load = [&](const QuadMap& q) {
#pragma unroll
    for (int it = 0; it < NITMAX; ++it) {
      const int i = it * 256 + tid;
      const int px = i >> 2;
      const int row = px >> a.logW, col = px & (a.W - 1);
      const int ir = r0 - 1 + row;
      const bool ok = i < total && q.valid && (unsigned)ir < (unsigned)a.H;
      const float* xp = a.x + (img_base + (long)ir * a.W + col) * a.ldx;
      if (q.contig) {
        R[it] = ok ? *reinterpret_cast<const f32x4*>(xp + q.c) : zero;
      } else {
        f32x4 v = zero;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int cm = a.cmap[q.e + j];
          float t = ok ? xp[cm & 0x7fffffff] : 0.f;
          v[j] = cm < 0 ? -t : t;
        }
        R[it] = v;
      }
    }
  };
  auto stage_store = [&](float sg) {
#pragma unroll
    for (int it = 0; it < NITMAX; ++it) {
      const int i = it * 256 + tid;
      if (i < total) {
        const int px = i >> 2;
        const int row = px >> a.logW, col = px & (a.W - 1);
        f32x4 v = R[it];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = d16_act<ACT>(v[j] * sg);
        unsigned h0, m0, l0, h1, m1, l1;
        d16_split2(v[0], v[1], h0, m0, l0);
        d16_split2(v[2], v[3], h1, m1, l1);
        unsigned char* dst = smem3 + (row * a.RS + col + 1) * 32 + slot * 8;
        *reinterpret_cast<u32x2*>(dst) = u32x2{h0, h1};
        *reinterpret_cast<u32x2*>(dst + TILE) = u32x2{m0, m1};
        *reinterpret_cast<u32x2*>(dst + 2 * TILE) = u32x2{l0, l1};
      }
    }
  };
  auto load_cm = [&](int chunk) {
    const int e = chunk * 16 + 4 * slot;
    i32x4 cm = {0, 0, 0, 0};
    if (a.cmap && e < a.Ceff) cm = *reinterpret_cast<const i32x4*>(a.cmap + e);
    return cm;
  };

  // fragment addressing (bytes): M-tile = 16 consecutive pixels of the row band
  int ab[PT];
#pragma unroll
  for (int t = 0; t < PT; ++t) {
    const int q0 = (wave * PT + t) * 16;
    int rr, cc;
    if (W8) {
      rr = (q0 >> 3) + (p >> 3);
      cc = p & 7;
    } else {
      rr = q0 >> a.logW;
      cc = (q0 & (a.W - 1)) + p;
    }
    ab[t] = ((rr + 1) * a.RS + cc + 1) * 32 + 16 * (g & 1);
  }
  f32x4 acc[PT];
#pragma unroll
  for (int t = 0; t < PT; ++t) acc[t] = zero;

  const int nchunk = (a.Ceff + 15) >> 4;
  const int hiTap = g >> 1;                                  // 0: first tap of a pair, 1: second
  const float* wlane = a.wT + (long)p * a.K + 8 * (g & 1);   // + tap*Ceff + chunk*16
  // weights of one (chunk, tap pair) for this lane: 8 consecutive effective channels of its tap
  auto load_w = [&](int chunk, int tp, f32x4 (&Wv)[2]) {
    const int tap = 2 * tp + hiTap;
    const int e0 = chunk * 16 + 8 * (g & 1);
    const bool ok = tap < 9 && e0 < a.Ceff;
    const float* wp = wlane + (long)(ok ? tap : 0) * a.Ceff + chunk * 16;
    Wv[0] = ok ? *reinterpret_cast<const f32x4*>(wp) : zero;
    Wv[1] = ok ? *reinterpret_cast<const f32x4*>(wp + 4) : zero;
  };

  QuadMap q = d16_quad(a, 4 * slot, load_cm(0));
  i32x4 cm_next = load_cm(1);
  stage_load(q);
  f32x4 Wa[2], Wb[2];
  load_w(0, 0, Wa);
  __syncthreads();  // zero fill complete
  stage_store(q.contig ? q.sg : 1.f);
  __syncthreads();

  for (int c = 0; c < nchunk; ++c) {
    const bool more = c + 1 < nchunk;
    if (more) {
      q = d16_quad(a, (c + 1) * 16 + 4 * slot, cm_next);
      cm_next = load_cm(c + 2);
      stage_load(q);
    }
#pragma unroll
    for (int tp = 0; tp < 5; ++tp) {
      f32x4 (&Wc)[2] = (tp & 1) ? Wb : Wa;
      f32x4 (&Wn)[2] = (tp & 1) ? Wa : Wb;
      if (tp < 4) load_w(c, tp + 1, Wn);
      else if (more) load_w(c + 1, 0, Wn);
      // weight pieces
      unsigned wh[4], wm[4], wl[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) d16_split2(Wc[j >> 1][2 * (j & 1)], Wc[j >> 1][2 * (j & 1) + 1], wh[j], wm[j], wl[j]);
      const bf16x8 Bh = __builtin_bit_cast(bf16x8, u32x4{wh[0], wh[1], wh[2], wh[3]});
      const bf16x8 Bm = __builtin_bit_cast(bf16x8, u32x4{wm[0], wm[1], wm[2], wm[3]});
      const bf16x8 Bl = __builtin_bit_cast(bf16x8, u32x4{wl[0], wl[1], wl[2], wl[3]});
      // this lane's tap of the pair (the ninth tap has no partner: its upper k-groups re-read tap 8 against zero weights)
      const int t0 = 2 * tp, t1 = (2 * tp + 1 < 9) ? 2 * tp + 1 : 8;
      const int sh0 = ((t0 / 3 - 1) * a.RS + (t0 % 3 - 1)) * 32;
      const int sh1 = ((t1 / 3 - 1) * a.RS + (t1 % 3 - 1)) * 32;
      const int sh = hiTap ? sh1 : sh0;
#pragma unroll
      for (int t = 0; t < PT; ++t) {
        const unsigned char* ap = smem3 + ab[t] + sh;
        const bf16x8 Ah = *reinterpret_cast<const bf16x8*>(ap);
        const bf16x8 Am = *reinterpret_cast<const bf16x8*>(ap + TILE);
        const bf16x8 Al = *reinterpret_cast<const bf16x8*>(ap + 2 * TILE);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Al, Bh, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, Bl, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Am, Bm, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Am, Bh, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, Bm, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, Bh, acc[t], 0, 0, 0);
      }
    }
    if (more) {  // five pairs: the prefetch of (c+1, pair 0) landed in the odd buffer
      Wa[0] = Wb[0];
      Wa[1] = Wb[1];
    }
    __syncthreads();
    if (more) {
      stage_store(q.contig ? q.sg : 1.f);
      __syncthreads();
    }
  }
  const float b = a.bias ? a.bias[p] : 0.f;
  const long m0 = (img_base + (long)r0 * a.W);
  unsigned omax = 0u;
#pragma unroll
  for (int t = 0; t < PT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long m = m0 + (wave * PT + t) * 16 + 4 * g + r;
      float* yp = a.y + m * a.ldy + a.coff + p;
      const float o = acc[t][r] + b + (a.accumulate ? *yp : 0.f);
      *yp = o;
      const unsigned ob = amax_bits(o);
      omax = ob > omax ? ob : omax;
    }
  if (a.amax) amax_commit(a.amax, omax);
}

// =======================================================================================
// Growth-layer forward on TWO SCALED fp16 PIECES (round 4): the chains of a split dense block.
//
// What bound dense16_fwd_x3_kernel (rocprofv3 SQ counters, profiles/r04_pmc_sq_dense_before.txt): 6 VALU instructions per
// MFMA (every lane split its weights and every staged activation into three bf16 pieces, once per chunk and tap pair),
// waves parked at barriers / s_waitcnt for half of their cycles (two barriers per 16-channel chunk, two workgroups per
// compute unit) -- MFMA busy 0.26, LDS issue stalls 2 %.  Here:
//   * x * 2^sx = hi + lo in fp16 (22 bits; the arithmetic of the Winograd-domain GEMMs, gemm_x3.h): THREE
//     v_mfma_f32_16x16x32_f16 per product (hi*hi, hi*lo, lo*hi) instead of six, two LDS planes instead of three; sx from
//     the amax records of the slices read (one record per finished growth slice, written by this kernel's epilogue in
//     the layer that finished it, plus the record of the wide convolutions' sums);
//   * the weights arrive PRE-SPLIT in MFMA fragment order (dense16_h2_prepare: once per weight update, one launch per
//     block), a lane's operand of a (slice, sign, tap pair) is two 16-byte loads, no VALU;
//   * a K step is one 16-channel source SLICE with both CReLU signs (32 effective channels): one global load, one split,
//     relu(x) and relu(-x) from the same two pieces by a sign mask (fp16 negation is exact) -- half the staging work and
//     half the barriers per effective channel;
//   * the weights of the slice ride along through LDS (20 KB behind the four planes of (TR + 2) x (W + 2) pixels x 32
//     bytes: 64 KB per workgroup at 32 x 32, two workgroups per compute unit): no memory wait inside the matrix loop.
// Only for the CReLU chains of equal 16-channel list elements (otgan_conv_desc::list_width == 16); everything else keeps
// the kernels above.
// =======================================================================================
typedef _Float16 d16_h2 __attribute__((ext_vector_type(2)));
typedef _Float16 d16_h8 __attribute__((ext_vector_type(8)));
typedef float d16_f2 __attribute__((ext_vector_type(2)));
constexpr int kH2HdrBytes = 64;                       // int exponent of the weights' scale, padding
constexpr int kH2SliceU16 = 2 * 5 * 2 * 64 * 8;       // [sign][tap pair][piece][lane][8] fp16 per source slice

struct FwdH2Args {
  const float* x;        // first channel of the first slice; [N, H, W, ldx]
  const unsigned char* wq;   // prepared weights (dense16_h2_prepare)
  float* y;
  const float* rec;      // amax records of the input: nrec consecutive records (512 floats apart)
  int nrec;
  int nsl;               // source slices of 16 channels
  int N, H, W, logW, ldx, ldy, coff, TR, RS;
  float* amax;           // amax record of the sums written, or null
};

struct H2PrepArgs {
  const float* wT[16];   // [16][9 * 32 * nsl]
  unsigned char* out[16];
  int nsl[16];
};
// one workgroup per layer: largest magnitude -> exponent -> the two fp16 pieces of w * 2^(14 - e) in fragment order
__global__ __launch_bounds__(1024) void dense16_h2_prep_kernel(H2PrepArgs a) {
  __shared__ float red[16];
  const int L = blockIdx.x, tid = threadIdx.x;
  const float* w = a.wT[L];
  const int nsl = a.nsl[L], Ceff = 32 * nsl, K = 9 * Ceff, total = 16 * K;
  float m = 0.f;
  bool bad = false;
  for (int i = tid; i < total; i += 1024) {
    const float v = fabsf(w[i]);
    bad = bad || !(v <= 3.0e38f);
    m = fmaxf(m, v);
  }
  if (bad) m = __builtin_nanf("");
  for (int o = 32; o; o >>= 1) {
    const float t = __shfl_xor(m, o);
    m = (t != t || m != m) ? __builtin_nanf("") : fmaxf(m, t);
  }
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  m = red[0];
  for (int i = 1; i < 16; ++i) m = (red[i] != red[i] || m != m) ? __builtin_nanf("") : fmaxf(m, red[i]);
  int e = 0;
  if (m > 0.f) e = __builtin_amdgcn_frexp_expf(m);
  const float sc = (m == m) ? __builtin_ldexpf(1.f, 14 - e) : m;
  if (tid == 0) *reinterpret_cast<int*>(a.out[L]) = e;
  unsigned short* q = reinterpret_cast<unsigned short*>(a.out[L] + kH2HdrBytes);
  const int items = nsl * 2 * 5 * 64;               // (slice, sign, tap pair, lane): 8 k values each
  for (int it = tid; it < items; it += 1024) {
    const int lane = it & 63, tp = (it >> 6) % 5, sign = (it / 320) & 1, sl = it / 640;
    const int n = lane & 15, g = lane >> 4;
    const int tap = 2 * tp + (g >> 1);
    unsigned short* dh = q + (long)sl * kH2SliceU16 + (((sign * 5 + tp) * 2 + 0) * 64 + lane) * 8;
    unsigned short* dl = dh + 64 * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = 0.f;
      if (tap < 9) v = w[(long)n * K + tap * Ceff + 32 * sl + 16 * sign + 8 * (g & 1) + j] * sc;
      const _Float16 h = (_Float16)v;
      const _Float16 l = (_Float16)(v - (float)h);
      dh[j] = __builtin_bit_cast(unsigned short, h);
      dl[j] = __builtin_bit_cast(unsigned short, l);
    }
  }
}

template <int PT, int WW>
__global__ __launch_bounds__(256, 2) void dense16_fwd_h2_kernel(FwdH2Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smemh[];
  __shared__ float s_sc[2];
  constexpr int NIT = PT + 1;
  constexpr int WBYTES = kH2SliceU16 * 2;                    // prepared weights of one slice: 20 steps x 1 KiB
  // the image width is a template parameter: tile rows, LDS row stride and plane size are then compile-time constants and the
  // plane / piece / tile offsets of the fragment reads fold into the ds_read immediates (one address add per tap pair
  // instead of one per read: 100 -> 10 per slice)
  constexpr bool W8 = WW == 8;
  constexpr int TR = 64 * PT / WW, RS = WW == 8 ? 16 : WW + 2, LOGW = WW == 8 ? 3 : WW == 16 ? 4 : 5;
  constexpr int PLANE = (TR + 2) * RS * 32;                 // bytes of one (sign, piece) plane
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p = lane & 15, g = lane >> 4;
  const int tiles_per_img = a.H / TR;
  const int n = blockIdx.x / tiles_per_img;
  const int r0 = (blockIdx.x - n * tiles_per_img) * TR;
  unsigned char* const smw = smemh + 4 * PLANE;             // the slice's weights behind the four planes
  for (int i = tid; i < 4 * PLANE / 16; i += 256) reinterpret_cast<u32x4*>(smemh)[i] = u32x4{0u, 0u, 0u, 0u};
  // input scale: the maximum of the records' sub-slots (bit patterns of non-negative floats: unsigned order)
  if (wave == 0) {
    unsigned mb = 0u;
    for (int i = lane; i < 16 * a.nrec; i += 64) {
      const unsigned v = reinterpret_cast<const unsigned*>(a.rec)[(long)(i >> 4) * (kAmaxSub * kAmaxSubStride) + (i & 15) * kAmaxSubStride];
      mb = v > mb ? v : mb;
    }
    for (int o = 32; o; o >>= 1) {
      const unsigned t = __shfl_xor(mb, o);
      mb = t > mb ? t : mb;
    }
    if (lane == 0) {
      const float amax = __uint_as_float(mb);
      int e = 0;
      if (amax > 0.f) e = __builtin_amdgcn_frexp_expf(amax);
      const int ew = *reinterpret_cast<const int*>(a.wq);
      s_sc[0] = (amax <= 3.0e38f) ? __builtin_ldexpf(1.f, 14 - e) : __builtin_nanf("");
      s_sc[1] = (amax <= 3.0e38f) ? __builtin_ldexpf(1.f, e + ew - 28) : __builtin_nanf("");   // (a NaN record stays loud)
    }
  }
  const int slot = tid & 3;
  const int total = (TR + 2) * WW * 4;
  const long img_base = (long)n * a.H * WW;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  // staging items of this thread: (pixel of the (TR + 2) x W band, quad of 4 channels); source offset (floats, slice 0;
  // negative: outside the image or the band) and LDS byte offset, computed once
  long xoff[NIT];
  int loff[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = it * 256 + tid;
    const int px = i >> 2;
    const int row = px >> LOGW, col = px & (WW - 1);
    const int ir = r0 - 1 + row;
    const bool ok = i < total && (unsigned)ir < (unsigned)a.H;
    xoff[it] = ok ? (img_base + (long)ir * WW + col) * a.ldx + 4 * slot : -1;
    loff[it] = i < total ? (row * RS + col + 1) * 32 + slot * 8 : -1;
  }
  f32x4 R[NIT];
  u32x4 WR[5];
  const unsigned char* wsrc = a.wq + kH2HdrBytes + tid * 16;
  auto stage_load = [&](int sl) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) R[it] = xoff[it] >= 0 ? *reinterpret_cast<const f32x4*>(a.x + xoff[it] + 16 * sl) : zero;
#pragma unroll
    for (int q = 0; q < 5; ++q) WR[q] = *reinterpret_cast<const u32x4*>(wsrc + (long)sl * WBYTES + q * 4096);
  };
  auto stage_store = [&](float sx) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      if (loff[it] >= 0) {
        unsigned wd[4][2];       // [plane: +hi, +lo, -hi, -lo][element pair]
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const d16_f2 v = d16_f2{R[it][2 * h], R[it][2 * h + 1]} * sx;
          const d16_h2 hi = __builtin_convertvector(v, d16_h2);
          const d16_h2 lo = __builtin_convertvector(v - __builtin_convertvector(hi, d16_f2), d16_h2);
          const d16_h2 z = {(_Float16)0.f, (_Float16)0.f};
          // relu(x) = hi+ + lo+, relu(-x) = hi- + lo- from the same two pieces (fp16 negation is exact): the sign of hi
          // decides per element (hi = 0: |x 2^sx| < 2^-25, both sides take the sub-ulp lo piece or drop it -- 2^-39 of amax)
          typedef short d16_s2 __attribute__((ext_vector_type(2)));
          const unsigned neg = __builtin_bit_cast(unsigned, (d16_s2)(__builtin_bit_cast(d16_s2, hi) >> 15));   // 0xffff per negative half
          const unsigned lb = __builtin_bit_cast(unsigned, lo);
          wd[0][h] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(hi, z));
          wd[1][h] = lb & ~neg;
          wd[2][h] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(-hi, z));
          wd[3][h] = (lb ^ 0x80008000u) & neg;
        }
        unsigned char* dst = smemh + loff[it];
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<u32x2*>(dst + q * PLANE) = u32x2{wd[q][0], wd[q][1]};
      }
    }
#pragma unroll
    for (int q = 0; q < 5; ++q) *reinterpret_cast<u32x4*>(smw + q * 4096 + tid * 16) = WR[q];
  };
  int ab[PT];
#pragma unroll
  for (int t = 0; t < PT; ++t) {
    const int q0 = (wave * PT + t) * 16;
    int rr, cc;
    if (W8) {
      rr = (q0 >> 3) + (p >> 3);
      cc = p & 7;
    } else {
      rr = q0 >> LOGW;
      cc = (q0 & (WW - 1)) + p;
    }
    ab[t] = ((rr + 1) * RS + cc + 1) * 32 + 16 * (g & 1);
  }
  f32x4 acc[PT];
#pragma unroll
  for (int t = 0; t < PT; ++t) acc[t] = zero;
  const int hiTap = g >> 1;
  // Weights of a slice: twenty 1 KiB steps (sign, tap pair, piece), each the 64 lanes' 16-byte MFMA operands in lane order.
  // They travel with the activations: loaded (coalesced, L2 hits) while the previous slice is multiplied, stored into LDS
  // behind the planes, read back as two ds_read_b128 per step -- the matrix loop waits on LDS only, never on memory
  // (a first version read them from global memory inside the loop: ten exposed L2 latencies per slice, 73 us; a register
  // ring three steps ahead: 59 us, but its loads queue behind the next slice's activation loads in vmcnt order).
  stage_load(0);
  __syncthreads();               // zero fill and scales complete
  const float sx = s_sc[0];
  stage_store(sx);
  __syncthreads();
  const long m0 = (img_base + (long)r0 * WW);
  float yv[PT][4];
  for (int sl = 0; sl < a.nsl; ++sl) {
    const bool more = sl + 1 < a.nsl;
    if (more) {
      stage_load(sl + 1);
    } else {
      // the sums this workgroup adds onto: fetched under the last slice's matrix work
#pragma unroll
      for (int t = 0; t < PT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) yv[t][r] = a.y[(m0 + (wave * PT + t) * 16 + 4 * g + r) * a.ldy + a.coff + p];
    }
#pragma unroll
    for (int st = 0; st < 10; ++st) {
      const int sign = st / 5, tp = st % 5;
      const unsigned char* plane = smemh + 2 * sign * PLANE;
      const d16_h8 Bh = *reinterpret_cast<const d16_h8*>(smw + st * 2048 + lane * 16);
      const d16_h8 Bl = *reinterpret_cast<const d16_h8*>(smw + st * 2048 + 1024 + lane * 16);
      const int t0 = 2 * tp, t1 = (2 * tp + 1 < 9) ? 2 * tp + 1 : 8;    // (the ninth tap's partner: tap 8 against zero weights)
      const int sh0 = ((t0 / 3 - 1) * RS + (t0 % 3 - 1)) * 32;
      const int sh1 = ((t1 / 3 - 1) * RS + (t1 % 3 - 1)) * 32;
      const int sh = hiTap ? sh1 : sh0;
      d16_h8 Ah[PT], Al[PT];
#pragma unroll
      for (int t = 0; t < PT; ++t) {
        const unsigned char* ap = plane + ab[t] + sh;
        Ah[t] = *reinterpret_cast<const d16_h8*>(ap);
        Al[t] = *reinterpret_cast<const d16_h8*>(ap + PLANE);
      }
      // term-major: consecutive matrix instructions never share an accumulator (smallest terms first)
#pragma unroll
      for (int t = 0; t < PT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al[t], Bh, acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < PT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[t], Bl, acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < PT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[t], Bh, acc[t], 0, 0, 0);
    }
    __syncthreads();
    if (more) {
      stage_store(sx);
      __syncthreads();
    }
  }
  const float so = s_sc[1];
  unsigned omax = 0u;
#pragma unroll
  for (int t = 0; t < PT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long m = m0 + (wave * PT + t) * 16 + 4 * g + r;
      const float o = fmaf(acc[t][r], so, yv[t][r]);          // the chains add onto the wide convolutions' sums
      a.y[m * a.ldy + a.coff + p] = o;
      const unsigned ob = amax_bits(o);
      omax = ob > omax ? ob : omax;
    }
  if (a.amax) amax_commit(a.amax, omax);
}

// ---------------------------------------------------------------------------------------------------------------------
// One launch per chain (round 6).  Where a workgroup of dense16_fwd_h2_kernel covers a WHOLE image (8 x 8
// with PT = 1, 16 x 16 with PT = 4), layer j + 1 of a group's chain reads, for its image, only what layer j's workgroup OF THE
// SAME IMAGE wrote: the chain has no dependence between workgroups, and one workgroup can walk all the layers of its image --
// the launches of nslices - 1 kernels that ran 9 us (8 x 8) / 15 us (16 x 16) each at MFMA busy 0.05 - 0.15 become one.
// What changes in the arithmetic: the scale of a layer's fp16 operand pieces.  The per-layer kernel takes it from the GLOBAL amax
// records of the input slices; the records of the slices this launch produces are still being accumulated by the other
// workgroups when the next layer starts, so a workgroup bounds those slices by the largest magnitude IT wrote (its image is all
// it reads) together with the records that were final before the launch (the wide convolutions' sums, slice 0).  A valid
// bound, per image instead of per batch: deterministic, never looser than the global one, results within an fp16-piece
// rounding (2^-22 of the operand) of the per-layer kernels'.  The global records of the produced slices are still written
// (later consumers: the wide convolutions behind the group, the weight gradients).
struct ChainH2Args {
  float* buf;                 // first channel of the group's first slice; [N, H, W, ld]
  const unsigned char* wq[16];    // prepared weights of chain layers 1 .. nslices - 1
  float* rec;                 // records: [0] the wide convolutions' sums, [1 + c] slice c (c >= 1 written here)
  int nslices;
  int N, H, W, ld;
};
template <int PT, int WW>
__global__ __launch_bounds__(256, 2) void dense16_chain_fwd_h2_kernel(ChainH2Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smemh[];
  __shared__ float s_sc[2];
  __shared__ unsigned s_run[2];                              // [0] bits of the largest magnitude this workgroup wrote; [1] base records
  constexpr int NIT = PT + 1;
  constexpr int WBYTES = kH2SliceU16 * 2;
  constexpr bool W8 = WW == 8;
  constexpr int TR = 64 * PT / WW, RS = WW == 8 ? 16 : WW + 2, LOGW = WW == 8 ? 3 : WW == 16 ? 4 : 5;
  constexpr int PLANE = (TR + 2) * RS * 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p = lane & 15, g = lane >> 4;
  const int n = blockIdx.x;                                  // one workgroup per image (TR == H)
  unsigned char* const smw = smemh + 4 * PLANE;
  for (int i = tid; i < 4 * PLANE / 16; i += 256) reinterpret_cast<u32x4*>(smemh)[i] = u32x4{0u, 0u, 0u, 0u};
  if (wave == 0) {                                           // records final before the launch: the wide sums and slice 0
    unsigned mb = 0u;
    for (int i = lane; i < 16 * 2; i += 64) {
      const unsigned v = reinterpret_cast<const unsigned*>(a.rec)[(long)(i >> 4) * (kAmaxSub * kAmaxSubStride) + (i & 15) * kAmaxSubStride];
      mb = v > mb ? v : mb;
    }
    for (int o = 32; o; o >>= 1) {
      const unsigned t = __shfl_xor(mb, o);
      mb = t > mb ? t : mb;
    }
    if (lane == 0) { s_run[0] = 0u; s_run[1] = mb; }
  }
  const int slot = tid & 3;
  const int total = (TR + 2) * WW * 4;
  const long img_base = (long)n * a.H * WW;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  long xoff[NIT];
  int loff[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = it * 256 + tid;
    const int px = i >> 2;
    const int row = px >> LOGW, col = px & (WW - 1);
    const int ir = row - 1;                                  // (r0 = 0: the band is the image with one halo row above and below)
    const bool ok = i < total && (unsigned)ir < (unsigned)a.H;
    xoff[it] = ok ? (img_base + (long)ir * WW + col) * a.ld + 4 * slot : -1;
    loff[it] = i < total ? (row * RS + col + 1) * 32 + slot * 8 : -1;
  }
  int ab[PT];
#pragma unroll
  for (int t = 0; t < PT; ++t) {
    const int q0 = (wave * PT + t) * 16;
    int rr, cc;
    if (W8) {
      rr = (q0 >> 3) + (p >> 3);
      cc = p & 7;
    } else {
      rr = q0 >> LOGW;
      cc = (q0 & (WW - 1)) + p;
    }
    ab[t] = ((rr + 1) * RS + cc + 1) * 32 + 16 * (g & 1);
  }
  const int hiTap = g >> 1;
  const long m0 = img_base;
  const float* const x = a.buf;
  for (int j = 1; j < a.nslices; ++j) {
    const unsigned char* const wq = a.wq[j - 1];
    __syncthreads();               // the zero fill (first trip) / the previous layer's LDS reads and s_run update are complete
    if (tid == 0) {
      const unsigned mb = s_run[0] > s_run[1] ? s_run[0] : s_run[1];
      const float amax = __uint_as_float(mb);
      int e = 0;
      if (amax > 0.f) e = __builtin_amdgcn_frexp_expf(amax);
      const int ew = *reinterpret_cast<const int*>(wq);
      s_sc[0] = (amax <= 3.0e38f) ? __builtin_ldexpf(1.f, 14 - e) : __builtin_nanf("");
      s_sc[1] = (amax <= 3.0e38f) ? __builtin_ldexpf(1.f, e + ew - 28) : __builtin_nanf("");
    }
    f32x4 R[NIT];
    u32x4 WR[5];
    const unsigned char* wsrc = wq + kH2HdrBytes + tid * 16;
    auto stage_load = [&](int sl) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) R[it] = xoff[it] >= 0 ? *reinterpret_cast<const f32x4*>(x + xoff[it] + 16 * sl) : zero;
#pragma unroll
      for (int q = 0; q < 5; ++q) WR[q] = *reinterpret_cast<const u32x4*>(wsrc + (long)sl * WBYTES + q * 4096);
    };
    auto stage_store = [&](float sx) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        if (loff[it] >= 0) {
          unsigned wd[4][2];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const d16_f2 v = d16_f2{R[it][2 * h], R[it][2 * h + 1]} * sx;
            const d16_h2 hi = __builtin_convertvector(v, d16_h2);
            const d16_h2 lo = __builtin_convertvector(v - __builtin_convertvector(hi, d16_f2), d16_h2);
            const d16_h2 z = {(_Float16)0.f, (_Float16)0.f};
            typedef short d16_s2 __attribute__((ext_vector_type(2)));
            const unsigned neg = __builtin_bit_cast(unsigned, (d16_s2)(__builtin_bit_cast(d16_s2, hi) >> 15));
            const unsigned lb = __builtin_bit_cast(unsigned, lo);
            wd[0][h] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(hi, z));
            wd[1][h] = lb & ~neg;
            wd[2][h] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(-hi, z));
            wd[3][h] = (lb ^ 0x80008000u) & neg;
          }
          unsigned char* dst = smemh + loff[it];
#pragma unroll
          for (int q = 0; q < 4; ++q) *reinterpret_cast<u32x2*>(dst + q * PLANE) = u32x2{wd[q][0], wd[q][1]};
        }
      }
#pragma unroll
      for (int q = 0; q < 5; ++q) *reinterpret_cast<u32x4*>(smw + q * 4096 + tid * 16) = WR[q];
    };
    f32x4 acc[PT];
#pragma unroll
    for (int t = 0; t < PT; ++t) acc[t] = zero;
    stage_load(0);
    __syncthreads();               // scales complete
    const float sx = s_sc[0];
    stage_store(sx);
    __syncthreads();
    const int coff = 16 * j;
    float yv[PT][4];
    for (int sl = 0; sl < j; ++sl) {
      const bool more = sl + 1 < j;
      if (more) {
        stage_load(sl + 1);
      } else {
#pragma unroll
        for (int t = 0; t < PT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) yv[t][r] = a.buf[(m0 + (wave * PT + t) * 16 + 4 * g + r) * a.ld + coff + p];
      }
#pragma unroll
      for (int st = 0; st < 10; ++st) {
        const int sign = st / 5, tp = st % 5;
        const unsigned char* plane = smemh + 2 * sign * PLANE;
        const d16_h8 Bh = *reinterpret_cast<const d16_h8*>(smw + st * 2048 + lane * 16);
        const d16_h8 Bl = *reinterpret_cast<const d16_h8*>(smw + st * 2048 + 1024 + lane * 16);
        const int t0 = 2 * tp, t1 = (2 * tp + 1 < 9) ? 2 * tp + 1 : 8;
        const int sh0 = ((t0 / 3 - 1) * RS + (t0 % 3 - 1)) * 32;
        const int sh1 = ((t1 / 3 - 1) * RS + (t1 % 3 - 1)) * 32;
        const int sh = hiTap ? sh1 : sh0;
        d16_h8 Ah[PT], Al[PT];
#pragma unroll
        for (int t = 0; t < PT; ++t) {
          const unsigned char* ap = plane + ab[t] + sh;
          Ah[t] = *reinterpret_cast<const d16_h8*>(ap);
          Al[t] = *reinterpret_cast<const d16_h8*>(ap + PLANE);
        }
#pragma unroll
        for (int t = 0; t < PT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al[t], Bh, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < PT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[t], Bl, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < PT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[t], Bh, acc[t], 0, 0, 0);
      }
      __syncthreads();
      if (more) {
        stage_store(sx);
        __syncthreads();
      }
    }
    const float so = s_sc[1];
    unsigned omax = 0u;
#pragma unroll
    for (int t = 0; t < PT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long m = m0 + (wave * PT + t) * 16 + 4 * g + r;
        const float o = fmaf(acc[t][r], so, yv[t][r]);
        a.buf[m * a.ld + coff + p] = o;
        const unsigned ob = amax_bits(o);
        omax = ob > omax ? ob : omax;
      }
    // the workgroup's own maximum bounds what the next layers read of this slice; the global record for the later consumers
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned w = (unsigned)__shfl_xor((int)omax, o, 64);
      omax = w > omax ? w : omax;
    }
    if (lane == 0) atomicMax(&s_run[0], omax);
    amax_commit(a.rec + (size_t)(1 + j) * (kAmaxSub * kAmaxSubStride), omax);
    // (the slice just written is read back by other waves of THIS workgroup in the next trip: the barrier at the top of the loop
    // orders the stores before those loads -- workgroup scope, the vector cache is shared by the workgroup's waves; an
    // agent-scope release here would write the whole L2 back once per layer and workgroup)
  }
}

// =======================================================================================
// Input gradient of the chains BY SLICE on two scaled fp16 pieces (round 4).
//
// The per-layer input gradient (dense16_dgrad_kernel) makes layer k add its share into every slice of its chain: a
// 16-byte read-modify-write of the gradient buffer per (layer, earlier slice) pair -- 28 slice passes per half block, and
// that traffic, not the matrix pipe, is what the kernel runs against.  Transposed: the gradient of ONE slice c gathers
// from all later layers of its group in one launch,
//     dG_c[q] += [x_c > 0] G+ - [x_c < 0] G-,   G+-[q][cc] = sum_{k > c} sum_{tap, n} G_k[q + d(tap)][n] w_k[8 - tap][e+-(c, cc)][n]
// -- the forward kernel's loop with the roles turned: the K steps run over the gradient slices of the later layers
// (two LDS planes: no activation, one sign), each step feeds two weight operands (the + and - effective channels of
// slice c), and the slice's gradient is read and written ONCE.  Slices are processed last to first, so every source
// slice is final when it is read.  Weights per (c, k) pair prepared like the forward ones (flipped taps), ONE scale
// exponent for the whole block (the K loop sums over layers).
// =======================================================================================
struct BwdH2Args {
  const float* g;        // gradient buffer at the first channel of slice c + 1; [N, H, W, ldg]
  const unsigned char* wq;   // prepared weights of output slice c: header + nsl blocks (source layers c + 1 ...)
  const float* x;        // forward buffer at the first channel of slice c
  float* dx;             // gradient buffer at the first channel of slice c
  const float* rec0;     // amax records bounding the source slices: nrec0 + nrec1 records in two ranges
  const float* rec1;
  int nrec0, nrec1;
  int nsl;
  int N, H, W, logW, ldg, ldx, TR, RS;
  float* amax;           // amax record of the sums written, or null
};

struct H2BwdPrepArgs {
  const float* w[64];         // HWIO weights of the source layer's chain: [9][32 * nch][16]
  const unsigned char* fwd[64];   // the same layer's FORWARD prepared buffer (its header holds the layer's exponent)
  unsigned char* out[64];     // this pair's 20 KB block
  unsigned char* hdr[64];     // header of the output slice's buffer (written by pair 0 of the slice: flag)
  int nch[64];                // slices of the source layer's chain
  int cidx[64];               // index of slice c inside that chain
  int first[64];
  const unsigned char* allfwd[16];   // forward buffers of every layer with a chain (block-wide exponent)
  int nall;
};
__global__ __launch_bounds__(256) void dense16_h2_bwd_prep_kernel(H2BwdPrepArgs a) {
  const int P = blockIdx.x, tid = threadIdx.x;
  int e = -1000;
  for (int i = 0; i < a.nall; ++i) {
    const int ei = *reinterpret_cast<const int*>(a.allfwd[i]);
    e = ei > e ? ei : e;
  }
  if (a.first[P] && tid == 0) *reinterpret_cast<int*>(a.hdr[P]) = e;
  const float sc = __builtin_ldexpf(1.f, 14 - e);
  const float* w = a.w[P];
  const int Ceff = 32 * a.nch[P];
  unsigned short* q = reinterpret_cast<unsigned short*>(a.out[P]);
  for (int it = tid; it < 2 * 5 * 64; it += 256) {               // (sign, tap pair, lane)
    const int lane = it & 63, tp = (it >> 6) % 5, sign = it / 320;
    const int cc = lane & 15, g = lane >> 4;
    const int tapp = 2 * tp + (g >> 1);
    unsigned short* dh = q + (((sign * 5 + tp) * 2 + 0) * 64 + lane) * 8;
    unsigned short* dl = dh + 64 * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = 0.f;
      if (tapp < 9) v = w[((long)(8 - tapp) * Ceff + 32 * a.cidx[P] + 16 * sign + cc) * 16 + 8 * (g & 1) + j] * sc;
      const _Float16 h = (_Float16)v;
      const _Float16 l = (_Float16)(v - (float)h);
      dh[j] = __builtin_bit_cast(unsigned short, h);
      dl[j] = __builtin_bit_cast(unsigned short, l);
    }
  }
}

template <int PT, int WW>
__global__ __launch_bounds__(256, 2) void dense16_bwd_h2_kernel(BwdH2Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smemh[];
  __shared__ float s_sc[2];
  constexpr int NIT = PT + 1;
  constexpr int WBYTES = kH2SliceU16 * 2;
  // the image width is a template parameter: tile rows, LDS row stride and plane size are then compile-time constants and the
  // plane / piece / tile offsets of the fragment reads fold into the ds_read immediates (one address add per tap pair
  // instead of one per read: 100 -> 10 per slice)
  constexpr bool W8 = WW == 8;
  constexpr int TR = 64 * PT / WW, RS = WW == 8 ? 16 : WW + 2, LOGW = WW == 8 ? 3 : WW == 16 ? 4 : 5;
  constexpr int PLANE = (TR + 2) * RS * 32;                 // bytes of one (sign, piece) plane
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p = lane & 15, g = lane >> 4;
  const int tiles_per_img = a.H / TR;
  const int n = blockIdx.x / tiles_per_img;
  const int r0 = (blockIdx.x - n * tiles_per_img) * TR;
  unsigned char* const smw = smemh + 2 * PLANE;
  for (int i = tid; i < 2 * PLANE / 16; i += 256) reinterpret_cast<u32x4*>(smemh)[i] = u32x4{0u, 0u, 0u, 0u};
  if (wave == 0) {
    unsigned mb = 0u;
    const int nr = a.nrec0 + a.nrec1;
    for (int i = lane; i < 16 * nr; i += 64) {
      const int r = i >> 4;
      const float* base = r < a.nrec0 ? a.rec0 + (long)r * (kAmaxSub * kAmaxSubStride) : a.rec1 + (long)(r - a.nrec0) * (kAmaxSub * kAmaxSubStride);
      const unsigned v = reinterpret_cast<const unsigned*>(base)[(i & 15) * kAmaxSubStride];
      mb = v > mb ? v : mb;
    }
    for (int o = 32; o; o >>= 1) {
      const unsigned t = __shfl_xor(mb, o);
      mb = t > mb ? t : mb;
    }
    if (lane == 0) {
      const float amax = __uint_as_float(mb);
      int e = 0;
      if (amax > 0.f) e = __builtin_amdgcn_frexp_expf(amax);
      const int ew = *reinterpret_cast<const int*>(a.wq);
      s_sc[0] = (amax <= 3.0e38f) ? __builtin_ldexpf(1.f, 14 - e) : __builtin_nanf("");
      s_sc[1] = (amax <= 3.0e38f) ? __builtin_ldexpf(1.f, e + ew - 28) : __builtin_nanf("");
    }
  }
  const int slot = tid & 3;
  const int total = (TR + 2) * WW * 4;
  const long img_base = (long)n * a.H * WW;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  long goff[NIT];
  int loff[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = it * 256 + tid;
    const int px = i >> 2;
    const int row = px >> LOGW, col = px & (WW - 1);
    const int ir = r0 - 1 + row;
    const bool ok = i < total && (unsigned)ir < (unsigned)a.H;
    goff[it] = ok ? (img_base + (long)ir * WW + col) * a.ldg + 4 * slot : -1;
    loff[it] = i < total ? (row * RS + col + 1) * 32 + slot * 8 : -1;
  }
  f32x4 R[NIT];
  u32x4 WR[5];
  const unsigned char* wsrc = a.wq + kH2HdrBytes + tid * 16;
  auto stage_load = [&](int sl) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) R[it] = goff[it] >= 0 ? *reinterpret_cast<const f32x4*>(a.g + goff[it] + 16 * sl) : zero;
#pragma unroll
    for (int q = 0; q < 5; ++q) WR[q] = *reinterpret_cast<const u32x4*>(wsrc + (long)sl * WBYTES + q * 4096);
  };
  auto stage_store = [&](float sd) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      if (loff[it] >= 0) {
        unsigned wd[2][2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const d16_f2 v = d16_f2{R[it][2 * h], R[it][2 * h + 1]} * sd;
          const d16_h2 hi = __builtin_convertvector(v, d16_h2);
          const d16_h2 lo = __builtin_convertvector(v - __builtin_convertvector(hi, d16_f2), d16_h2);
          wd[0][h] = __builtin_bit_cast(unsigned, hi);
          wd[1][h] = __builtin_bit_cast(unsigned, lo);
        }
        unsigned char* dst = smemh + loff[it];
        *reinterpret_cast<u32x2*>(dst) = u32x2{wd[0][0], wd[0][1]};
        *reinterpret_cast<u32x2*>(dst + PLANE) = u32x2{wd[1][0], wd[1][1]};
      }
    }
#pragma unroll
    for (int q = 0; q < 5; ++q) *reinterpret_cast<u32x4*>(smw + q * 4096 + tid * 16) = WR[q];
  };
  int ab[PT];
#pragma unroll
  for (int t = 0; t < PT; ++t) {
    const int q0 = (wave * PT + t) * 16;
    int rr, cc;
    if (W8) {
      rr = (q0 >> 3) + (p >> 3);
      cc = p & 7;
    } else {
      rr = q0 >> LOGW;
      cc = (q0 & (WW - 1)) + p;
    }
    ab[t] = ((rr + 1) * RS + cc + 1) * 32 + 16 * (g & 1);
  }
  f32x4 acc[2][PT];
#pragma unroll
  for (int t = 0; t < PT; ++t) acc[0][t] = acc[1][t] = zero;
  const int hiTap = g >> 1;
  stage_load(0);
  __syncthreads();
  const float sd = s_sc[0];
  stage_store(sd);
  __syncthreads();
  const long m0 = (img_base + (long)r0 * WW);
  float xv[PT][4], old[PT][4];
  for (int sl = 0; sl < a.nsl; ++sl) {
    const bool more = sl + 1 < a.nsl;
    if (more) {
      stage_load(sl + 1);
    } else if (PT < 4) {       // (PT = 4: 32 more live registers would spill; loaded in the epilogue)
#pragma unroll
      for (int t = 0; t < PT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const long m = m0 + (wave * PT + t) * 16 + 4 * g + r;
          xv[t][r] = a.x[m * a.ldx + p];
          old[t][r] = a.dx[m * a.ldg + p];
        }
    }
#pragma unroll
    for (int tp = 0; tp < 5; ++tp) {
      const d16_h8 Bhp = *reinterpret_cast<const d16_h8*>(smw + tp * 2048 + lane * 16);
      const d16_h8 Blp = *reinterpret_cast<const d16_h8*>(smw + tp * 2048 + 1024 + lane * 16);
      const d16_h8 Bhn = *reinterpret_cast<const d16_h8*>(smw + (5 + tp) * 2048 + lane * 16);
      const d16_h8 Bln = *reinterpret_cast<const d16_h8*>(smw + (5 + tp) * 2048 + 1024 + lane * 16);
      const int t0 = 2 * tp, t1 = (2 * tp + 1 < 9) ? 2 * tp + 1 : 8;
      const int sh0 = ((t0 / 3 - 1) * RS + (t0 % 3 - 1)) * 32;
      const int sh1 = ((t1 / 3 - 1) * RS + (t1 % 3 - 1)) * 32;
      const int sh = hiTap ? sh1 : sh0;
      d16_h8 Ah[PT], Al[PT];
#pragma unroll
      for (int t = 0; t < PT; ++t) {
        const unsigned char* ap = smemh + ab[t] + sh;
        Ah[t] = *reinterpret_cast<const d16_h8*>(ap);
        Al[t] = *reinterpret_cast<const d16_h8*>(ap + PLANE);
      }
#pragma unroll
      for (int t = 0; t < PT; ++t) {
        acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al[t], Bhp, acc[0][t], 0, 0, 0);
        acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al[t], Bhn, acc[1][t], 0, 0, 0);
      }
#pragma unroll
      for (int t = 0; t < PT; ++t) {
        acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[t], Blp, acc[0][t], 0, 0, 0);
        acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[t], Bln, acc[1][t], 0, 0, 0);
      }
#pragma unroll
      for (int t = 0; t < PT; ++t) {
        acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[t], Bhp, acc[0][t], 0, 0, 0);
        acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[t], Bhn, acc[1][t], 0, 0, 0);
      }
    }
    __syncthreads();
    if (more) {
      stage_store(sd);
      __syncthreads();
    }
  }
  const float so = s_sc[1];
  unsigned omax = 0u;
  if (PT >= 4) {
#pragma unroll
    for (int t = 0; t < PT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long m = m0 + (wave * PT + t) * 16 + 4 * g + r;
        xv[t][r] = a.x[m * a.ldx + p];
        old[t][r] = a.dx[m * a.ldg + p];
      }
  }
#pragma unroll
  for (int t = 0; t < PT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long m = m0 + (wave * PT + t) * 16 + 4 * g + r;
      const float xq = xv[t][r];
      float o = old[t][r];
      // (a NaN scale must stay loud whatever the sign of x: add the sums, masked by a factor)
      o = fmaf(acc[0][t][r] * so, xq > 0.f ? 1.f : 0.f, o);
      o = fmaf(acc[1][t][r] * so, xq < 0.f ? -1.f : 0.f, o);
      a.dx[m * a.ldg + p] = o;
      const unsigned ob = amax_bits(o);
      omax = ob > omax ? ob : omax;
    }
  if (a.amax) amax_commit(a.amax, omax);
}

// One launch per chain, input gradient (round 6): the slices of a group last to first inside one workgroup per image, as in
// dense16_chain_fwd_h2_kernel -- slice c gathers from the gradient slices c + 1 .. of ITS image, all of them written by this
// workgroup in the trips before (or final before the launch: the last slice), so the chain needs no other workgroup.  The
// source slices are bounded by the records that were final before the launch (rec0: the incoming gradient and the wide
// convolutions' shares; the last slice's record) and by the largest magnitude this workgroup has written so far.
struct ChainBwdH2Args {
  float* g;                    // gradient buffer at the first channel of the group's first slice; [N, H, W, ldg]
  const float* x;              // forward buffer at the same channel; [N, H, W, ldx]
  const unsigned char* wq[16]; // prepared weights of output slices 0 .. nslices - 2
  const float* rec0;           // one record
  float* slice_rec;            // [nslices] records: [c] written here for c <= nslices - 2, [nslices - 1] read
  int nslices;
  int N, H, W, ldg, ldx;
};
template <int PT, int WW>
__global__ __launch_bounds__(256, 2) void dense16_chain_bwd_h2_kernel(ChainBwdH2Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smemh[];
  __shared__ float s_sc[2];
  __shared__ unsigned s_run[2];
  constexpr int NIT = PT + 1;
  constexpr int WBYTES = kH2SliceU16 * 2;
  constexpr bool W8 = WW == 8;
  constexpr int TR = 64 * PT / WW, RS = WW == 8 ? 16 : WW + 2, LOGW = WW == 8 ? 3 : WW == 16 ? 4 : 5;
  constexpr int PLANE = (TR + 2) * RS * 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p = lane & 15, g = lane >> 4;
  const int n = blockIdx.x;
  unsigned char* const smw = smemh + 2 * PLANE;
  for (int i = tid; i < 2 * PLANE / 16; i += 256) reinterpret_cast<u32x4*>(smemh)[i] = u32x4{0u, 0u, 0u, 0u};
  if (wave == 0) {
    unsigned mb = 0u;
    for (int i = lane; i < 16 * 2; i += 64) {
      const float* base = (i >> 4) == 0 ? a.rec0 : a.slice_rec + (long)(a.nslices - 1) * (kAmaxSub * kAmaxSubStride);
      const unsigned v = reinterpret_cast<const unsigned*>(base)[(i & 15) * kAmaxSubStride];
      mb = v > mb ? v : mb;
    }
    for (int o = 32; o; o >>= 1) {
      const unsigned t = __shfl_xor(mb, o);
      mb = t > mb ? t : mb;
    }
    if (lane == 0) { s_run[0] = 0u; s_run[1] = mb; }
  }
  const int slot = tid & 3;
  const int total = (TR + 2) * WW * 4;
  const long img_base = (long)n * a.H * WW;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  long goff[NIT];
  int loff[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = it * 256 + tid;
    const int px = i >> 2;
    const int row = px >> LOGW, col = px & (WW - 1);
    const int ir = row - 1;
    const bool ok = i < total && (unsigned)ir < (unsigned)a.H;
    goff[it] = ok ? (img_base + (long)ir * WW + col) * a.ldg + 4 * slot : -1;
    loff[it] = i < total ? (row * RS + col + 1) * 32 + slot * 8 : -1;
  }
  int ab[PT];
#pragma unroll
  for (int t = 0; t < PT; ++t) {
    const int q0 = (wave * PT + t) * 16;
    int rr, cc;
    if (W8) {
      rr = (q0 >> 3) + (p >> 3);
      cc = p & 7;
    } else {
      rr = q0 >> LOGW;
      cc = (q0 & (WW - 1)) + p;
    }
    ab[t] = ((rr + 1) * RS + cc + 1) * 32 + 16 * (g & 1);
  }
  const int hiTap = g >> 1;
  const long m0 = img_base;
  for (int c = a.nslices - 2; c >= 0; --c) {
    const int nsl = a.nslices - 1 - c;
    const unsigned char* const wq = a.wq[c];
    const float* const gsrc = a.g + 16 * (c + 1);
    float* const dx = a.g + 16 * c;
    const float* const xc = a.x + 16 * c;
    __syncthreads();
    if (tid == 0) {
      const unsigned mb = s_run[0] > s_run[1] ? s_run[0] : s_run[1];
      const float amax = __uint_as_float(mb);
      int e = 0;
      if (amax > 0.f) e = __builtin_amdgcn_frexp_expf(amax);
      const int ew = *reinterpret_cast<const int*>(wq);
      s_sc[0] = (amax <= 3.0e38f) ? __builtin_ldexpf(1.f, 14 - e) : __builtin_nanf("");
      s_sc[1] = (amax <= 3.0e38f) ? __builtin_ldexpf(1.f, e + ew - 28) : __builtin_nanf("");
    }
    f32x4 R[NIT];
    u32x4 WR[5];
    const unsigned char* wsrc = wq + kH2HdrBytes + tid * 16;
    auto stage_load = [&](int sl) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) R[it] = goff[it] >= 0 ? *reinterpret_cast<const f32x4*>(gsrc + goff[it] + 16 * sl) : zero;
#pragma unroll
      for (int q = 0; q < 5; ++q) WR[q] = *reinterpret_cast<const u32x4*>(wsrc + (long)sl * WBYTES + q * 4096);
    };
    auto stage_store = [&](float sd) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        if (loff[it] >= 0) {
          unsigned wd[2][2];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const d16_f2 v = d16_f2{R[it][2 * h], R[it][2 * h + 1]} * sd;
            const d16_h2 hi = __builtin_convertvector(v, d16_h2);
            const d16_h2 lo = __builtin_convertvector(v - __builtin_convertvector(hi, d16_f2), d16_h2);
            wd[0][h] = __builtin_bit_cast(unsigned, hi);
            wd[1][h] = __builtin_bit_cast(unsigned, lo);
          }
          unsigned char* dst = smemh + loff[it];
          *reinterpret_cast<u32x2*>(dst) = u32x2{wd[0][0], wd[0][1]};
          *reinterpret_cast<u32x2*>(dst + PLANE) = u32x2{wd[1][0], wd[1][1]};
        }
      }
#pragma unroll
      for (int q = 0; q < 5; ++q) *reinterpret_cast<u32x4*>(smw + q * 4096 + tid * 16) = WR[q];
    };
    f32x4 acc[2][PT];
#pragma unroll
    for (int t = 0; t < PT; ++t) acc[0][t] = acc[1][t] = zero;
    stage_load(0);
    __syncthreads();
    const float sd = s_sc[0];
    stage_store(sd);
    __syncthreads();
    for (int sl = 0; sl < nsl; ++sl) {
      const bool more = sl + 1 < nsl;
      if (more) stage_load(sl + 1);
#pragma unroll
      for (int tp = 0; tp < 5; ++tp) {
        const d16_h8 Bhp = *reinterpret_cast<const d16_h8*>(smw + tp * 2048 + lane * 16);
        const d16_h8 Blp = *reinterpret_cast<const d16_h8*>(smw + tp * 2048 + 1024 + lane * 16);
        const d16_h8 Bhn = *reinterpret_cast<const d16_h8*>(smw + (5 + tp) * 2048 + lane * 16);
        const d16_h8 Bln = *reinterpret_cast<const d16_h8*>(smw + (5 + tp) * 2048 + 1024 + lane * 16);
        const int t0 = 2 * tp, t1 = (2 * tp + 1 < 9) ? 2 * tp + 1 : 8;
        const int sh0 = ((t0 / 3 - 1) * RS + (t0 % 3 - 1)) * 32;
        const int sh1 = ((t1 / 3 - 1) * RS + (t1 % 3 - 1)) * 32;
        const int sh = hiTap ? sh1 : sh0;
        d16_h8 Ah[PT], Al[PT];
#pragma unroll
        for (int t = 0; t < PT; ++t) {
          const unsigned char* ap = smemh + ab[t] + sh;
          Ah[t] = *reinterpret_cast<const d16_h8*>(ap);
          Al[t] = *reinterpret_cast<const d16_h8*>(ap + PLANE);
        }
#pragma unroll
        for (int t = 0; t < PT; ++t) {
          acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al[t], Bhp, acc[0][t], 0, 0, 0);
          acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al[t], Bhn, acc[1][t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < PT; ++t) {
          acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[t], Blp, acc[0][t], 0, 0, 0);
          acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[t], Bln, acc[1][t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < PT; ++t) {
          acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[t], Bhp, acc[0][t], 0, 0, 0);
          acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[t], Bhn, acc[1][t], 0, 0, 0);
        }
      }
      __syncthreads();
      if (more) {
        stage_store(sd);
        __syncthreads();
      }
    }
    const float so = s_sc[1];
    unsigned omax = 0u;
#pragma unroll
    for (int t = 0; t < PT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long m = m0 + (wave * PT + t) * 16 + 4 * g + r;
        const float xq = xc[m * a.ldx + p];
        float o = dx[m * a.ldg + p];
        o = fmaf(acc[0][t][r] * so, xq > 0.f ? 1.f : 0.f, o);
        o = fmaf(acc[1][t][r] * so, xq < 0.f ? -1.f : 0.f, o);
        dx[m * a.ldg + p] = o;
        const unsigned ob = amax_bits(o);
        omax = ob > omax ? ob : omax;
      }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned w = (unsigned)__shfl_xor((int)omax, o, 64);
      omax = w > omax ? w : omax;
    }
    if (lane == 0) atomicMax(&s_run[0], omax);
    amax_commit(a.slice_rec + (size_t)c * (kAmaxSub * kAmaxSubStride), omax);
  }
}

// =======================================================================================
// Weight gradient:  dW[tap][e][n] = sum_q act(x)[q][e] * dy[q - tap][n]
//
// MFMA roles: M = 16 effective channels, N = the 16 output channels, K = pixels.  A block owns
// one 32-channel slice (blockIdx.y) and a range of pixel tiles (blockIdx.x); per tile it stages
// act(x) (64*PT pixels x 32 channels, no halo) and the dy rows with a one-pixel halo, and every
// staged activation fragment (one ds_read_b32) feeds nine MFMAs -- one per tap, against the nine
// shifted dy fragments, which are shared by the two channel tiles: 11 LDS reads per 18 MFMAs.
// k-slot g of a step holds pixel 4s + {0,2,1,3}[g]: the two pixels read by one 32-lane LDS
// group are 2 apart, which puts them 16 banks apart for both pixel strides (40 and 24 floats).
// The four waves take different pixels; their accumulators are summed through LDS and written
// as one slab per pixel split (deterministic, reduced by the caller).
// =======================================================================================
struct WgArgs {
  const float* x;
  const int32_t* cmap;
  const float* dy;
  float* slabs;
  int N, H, W, logW, ldx, C, Ceff, doubled, ldy;
  int TR, RS, tiles, tiles_per_split;
  long slab_elems;
  // H2 (round 4): amax records of x (the slices read) and of dy -- operands as two scaled fp16 pieces
  const float* x_rec;
  const float* dy_rec;
  int x_nrec, dy_nrec;
  int xmap, nsplit, nchunk;   // 1: one-dimensional grid, (split, chunk) from the XCD-aware map in the kernel
};

constexpr int kDyStride = 24;   // floats per pixel of the dy halo tile (16 + 8 pad)
constexpr int kAStride = 40;    // floats per pixel of the activation tile

// H2 (round 4): the same kernel on the fp16 matrix pipe.  The fp32 form is bound by v_mfma_f32_16x16x4_f32 -- 18 of them
// (32 clocks each) per four pixels, MFMA busy 0.52 of a pipe that peaks at 157 TFLOP/s: 128 us per launch at 32 x 32 x 256
// images, 2.9 ms of a DenseNet step.  Here the staging pass writes every activation and every dy value ONCE as a packed
// {hi, lo} pair of scaled fp16 pieces into the same LDS slots (x 2^(14 - e) from the tensors' amax records), the inner loop
// takes runs of four consecutive pixels per lane (k-group g of a 16-pixel step = pixels 4g .. 4g + 3: one image row), picks
// the hi and the lo halves apart with v_perm_b32, and feeds v_mfma_f32_16x16x16_f16: three per product (hi hi, hi lo,
// lo hi: 22 bits), 54 of 8 clocks per 16 pixels instead of 72 of 32.  The three taps of a filter row share one six-pixel
// window of dy reads.  Same accumulator layout, same reduction and slab write-out (times 2^(ex + edy - 28), exact).
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned wg_pack_pieces(float v) {      // {hi = fp16(v), lo = fp16(v - hi)} in one dword
  const _Float16 h = (_Float16)v;
  const _Float16 l = (_Float16)(v - (float)h);
  return (unsigned)__builtin_bit_cast(unsigned short, h) | ((unsigned)__builtin_bit_cast(unsigned short, l) << 16);
}
__device__ __forceinline__ h16x4 wg_halves(unsigned d0, unsigned d1, unsigned d2, unsigned d3, bool hi) {
  // bytes of v_perm_b32(s0, s1, sel): 0-3 = s1, 4-7 = s0
  const unsigned sel = hi ? 0x05040100u : 0x07060302u;
  u32x2 r = {__builtin_amdgcn_perm(d1, d0, sel), __builtin_amdgcn_perm(d3, d2, sel)};
  return __builtin_bit_cast(h16x4, r);
}
template <int PT, int ACT, bool H2 = false>
__global__ __launch_bounds__(256, 2) void dense16_wgrad_kernel(WgArgs a) {
  extern __shared__ f32x4 smem4[];
  __shared__ float s_wsc[3];     // H2: scale of x, scale of dy, 2^(ex + edy - 28)
  if (H2 && threadIdx.x < 64) {
    const int lane_ = threadIdx.x;
    int ex[2];
#pragma unroll
    for (int w = 0; w < 2; ++w) {
      const float* rec = w ? a.dy_rec : a.x_rec;
      const int nrec = w ? a.dy_nrec : a.x_nrec;
      unsigned mb = 0u;
      for (int i = lane_; i < 16 * nrec; i += 64) {
        const unsigned v = reinterpret_cast<const unsigned*>(rec)[(long)(i >> 4) * (kAmaxSub * kAmaxSubStride) + (i & 15) * kAmaxSubStride];
        mb = v > mb ? v : mb;
      }
      for (int o = 32; o; o >>= 1) {
        const unsigned t = __shfl_xor(mb, o);
        mb = t > mb ? t : mb;
      }
      float amax = __uint_as_float(mb);
      if (w == 0 && ACT == 2 && amax == amax) amax = fmaxf(amax, 1.f);      // |elu(x)| <= max(|x|, 1)
      int e = 0;
      if (amax > 0.f) e = __builtin_amdgcn_frexp_expf(amax);
      ex[w] = e;
      if (lane_ == 0) s_wsc[w] = (amax <= 3.0e38f) ? __builtin_ldexpf(1.f, 14 - e) : __builtin_nanf("");   // (a NaN record stays loud)
    }
    if (lane_ == 0) s_wsc[2] = __builtin_ldexpf(1.f, ex[0] + ex[1] - 28);
  }
  if (H2) __syncthreads();
  const float sxs = H2 ? s_wsc[0] : 1.f, sds = H2 ? s_wsc[1] : 1.f;
  float* sA = reinterpret_cast<float*>(smem4);                       // [64*PT][40]
  float* sD = sA + 64 * PT * kAStride;                               // [(TR+2)*RS][24]
  constexpr int NA = 2 * PT;           // float4 staging items per thread: activations
  constexpr int ND = PT + 2;           // dy halo rows: (64*PT + 2W)*4/256 = PT + W/32 <= PT + 2
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p = lane & 15, g = lane >> 4;
  // (split, chunk) of this workgroup.  xmap (round 4, one-dimensional grid): workgroup b runs on XCD b % 8 (round-robin
  // placement, observed -- a speed matter only); the chunks of ONE pixel split are given to consecutive workgroups of the same
  // XCD, so that the dy tile every chunk of the split reads again -- half the kernel's traffic at seven chunks -- comes out
  // of that XCD's L2 instead of memory.
  int chunk = blockIdx.y, split = blockIdx.x;
  if (a.xmap) {
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    chunk = idx % a.nchunk;
    split = (idx / a.nchunk) * 8 + xcd;
    if (split >= a.nsplit) return;
  }
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  const int t_begin = split * a.tiles_per_split;
  const int t_end = min(t_begin + a.tiles_per_split, a.tiles);
  const int tiles_per_img = a.H / a.TR;

  // zero the dy tile once: halo columns and out-of-image rows stay zero
  const int dquads = (a.TR + 2) * a.RS * (kDyStride / 4);
  for (int i = tid; i < dquads; i += 256) reinterpret_cast<f32x4*>(sD)[i] = zero;

  const int slot = tid & 7;
  FwdLdsArgs fa;   // only the fields d16_quad reads
  fa.cmap = a.cmap; fa.C = a.C; fa.Ceff = a.Ceff; fa.doubled = a.doubled;
  i32x4 cm = {0, 0, 0, 0};
  {
    const int e = chunk * 32 + 4 * slot;
    if (a.cmap && e < a.Ceff) cm = *reinterpret_cast<const i32x4*>(a.cmap + e);
  }
  const QuadMap q = d16_quad(fa, chunk * 32 + 4 * slot, cm);
  const float sg = q.contig ? q.sg : 1.f;

  f32x4 RA[NA], RD[ND];
  const int dtotal = (a.TR + 2) * a.W * 4;
  auto stage_load = [&](int tile) {
    const int n = tile / tiles_per_img;
    const int r0 = (tile - n * tiles_per_img) * a.TR;
    const long img_base = (long)n * a.H * a.W;
#pragma unroll
    for (int it = 0; it < NA; ++it) {
      const int i = it * 256 + tid;
      const int px = i >> 3;                       // < 64*PT
      const float* xp = a.x + (img_base + (long)r0 * a.W + px) * a.ldx;
      if (q.contig) {
        RA[it] = q.valid ? *reinterpret_cast<const f32x4*>(xp + q.c) : zero;
      } else {
        f32x4 v = zero;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int cmj = a.cmap[q.e + j];
          const float t = q.valid ? xp[cmj & 0x7fffffff] : 0.f;
          v[j] = cmj < 0 ? -t : t;
        }
        RA[it] = v;
      }
    }
#pragma unroll
    for (int it = 0; it < ND; ++it) {
      const int i = it * 256 + tid;
      const int px = i >> 2;
      const int row = px >> a.logW, col = px & (a.W - 1);
      const int ir = r0 - 1 + row;
      const bool ok = i < dtotal && (unsigned)ir < (unsigned)a.H;
      RD[it] = ok ? *reinterpret_cast<const f32x4*>(a.dy + (img_base + (long)ir * a.W + col) * a.ldy + 4 * (i & 3))
                  : zero;
    }
  };
  auto stage_store = [&]() {
#pragma unroll
    for (int it = 0; it < NA; ++it) {
      const int i = it * 256 + tid;
      f32x4 v = RA[it];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = d16_act<ACT>(v[j] * sg);
      if (H2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = __uint_as_float(wg_pack_pieces(v[j] * sxs));
      }
      *reinterpret_cast<f32x4*>(sA + (i >> 3) * kAStride + 4 * slot) = v;
    }
#pragma unroll
    for (int it = 0; it < ND; ++it) {
      const int i = it * 256 + tid;
      if (i < dtotal) {
        const int px = i >> 2;
        const int row = px >> a.logW, col = px & (a.W - 1);
        f32x4 v = RD[it];
        if (H2) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = __uint_as_float(wg_pack_pieces(v[j] * sds));
        }
        *reinterpret_cast<f32x4*>(sD + (row * a.RS + col + 1) * kDyStride + 4 * (i & 3)) = v;
      }
    }
  };

  f32x4 acc[9][2];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    acc[t][0] = zero;
    acc[t][1] = zero;
  }

  if (t_begin < t_end) stage_load(t_begin);
  __syncthreads();
  if (t_begin < t_end) stage_store();
  __syncthreads();
  const int perm = ((g & 1) << 1) | (g >> 1);
  for (int tile = t_begin; tile < t_end; ++tile) {
    const bool more = tile + 1 < t_end;
    if (more) stage_load(tile + 1);
    if constexpr (H2) {
      const unsigned* uA = reinterpret_cast<const unsigned*>(sA);
      const unsigned* uD = reinterpret_cast<const unsigned*>(sD);
#pragma unroll 1
      for (int s16 = 0; s16 < PT; ++s16) {
        const int q0 = (wave * PT + s16) * 16 + 4 * g;          // first of this lane's four pixels (one image row)
        const int row = q0 >> a.logW, col = q0 & (a.W - 1);
        const unsigned* ap = uA + q0 * kAStride + p;
        const unsigned a00 = ap[0], a01 = ap[kAStride], a02 = ap[2 * kAStride], a03 = ap[3 * kAStride];
        const unsigned a10 = ap[16], a11 = ap[kAStride + 16], a12 = ap[2 * kAStride + 16], a13 = ap[3 * kAStride + 16];
        const h16x4 A0h = wg_halves(a00, a01, a02, a03, true), A0l = wg_halves(a00, a01, a02, a03, false);
        const h16x4 A1h = wg_halves(a10, a11, a12, a13, true), A1l = wg_halves(a10, a11, a12, a13, false);
#pragma unroll
        for (int r3 = 0; r3 < 3; ++r3) {                       // filter row: dy_ = r3 - 1 reads halo row (row + 1 - dy_)
          const unsigned* dp = uD + ((row + 2 - r3) * a.RS + col) * kDyStride + p;     // halo column col = pixel col - 1
          unsigned w[6];
#pragma unroll
          for (int c = 0; c < 6; ++c) w[c] = dp[c * kDyStride];
#pragma unroll
          for (int c3 = 0; c3 < 3; ++c3) {                     // dx_ = c3 - 1: pixels col + j - dx_  ->  window index j + 2 - c3
            const int tap = r3 * 3 + c3;
            const h16x4 Bh = wg_halves(w[2 - c3], w[3 - c3], w[4 - c3], w[5 - c3], true);
            const h16x4 Bl = wg_halves(w[2 - c3], w[3 - c3], w[4 - c3], w[5 - c3], false);
            acc[tap][0] = __builtin_amdgcn_mfma_f32_16x16x16f16(A0l, Bh, acc[tap][0], 0, 0, 0);
            acc[tap][0] = __builtin_amdgcn_mfma_f32_16x16x16f16(A0h, Bl, acc[tap][0], 0, 0, 0);
            acc[tap][0] = __builtin_amdgcn_mfma_f32_16x16x16f16(A0h, Bh, acc[tap][0], 0, 0, 0);
            acc[tap][1] = __builtin_amdgcn_mfma_f32_16x16x16f16(A1l, Bh, acc[tap][1], 0, 0, 0);
            acc[tap][1] = __builtin_amdgcn_mfma_f32_16x16x16f16(A1h, Bl, acc[tap][1], 0, 0, 0);
            acc[tap][1] = __builtin_amdgcn_mfma_f32_16x16x16f16(A1h, Bh, acc[tap][1], 0, 0, 0);
          }
        }
      }
    } else
#pragma unroll 2
    for (int s = 0; s < 4 * PT; ++s) {
      const int qq = (wave * 4 * PT + s) * 4 + perm;          // tile pixel of this k-slot
      const int row = qq >> a.logW, col = qq & (a.W - 1);
      const float a0 = sA[qq * kAStride + p];
      const float a1 = sA[qq * kAStride + 16 + p];
      const float* dp = sD + ((row + 1) * a.RS + col + 1) * kDyStride + p;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int dy_ = tap / 3 - 1, dx_ = tap % 3 - 1;
        const float b = dp[-(dy_ * a.RS + dx_) * kDyStride];
        acc[tap][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b, acc[tap][0], 0, 0, 0);
        acc[tap][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b, acc[tap][1], 0, 0, 0);
      }
    }
    __syncthreads();
    if (more) {
      stage_store();
      __syncthreads();
    }
  }
  // sum the four waves' accumulators through LDS: [tap][mt][lane] float4
  f32x4* red = smem4;
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const int idx = (t * 2 + mt) * 64 + lane;
          red[idx] = (w == 0) ? acc[t][mt] : red[idx] + acc[t][mt];
        }
    }
    __syncthreads();
  }
  // D[m = 4g + r][n = p]  ->  slab[tap][chunk*32 + 16mt + m][n]; thread = (e_local, n): coalesced
  float* slab = a.slabs + (long)split * a.slab_elems;
  const float* redf = reinterpret_cast<const float*>(red);
  const int n_out = tid & 15, el = tid >> 4;
#pragma unroll
  for (int i = 0; i < 18; ++i) {
    const int tap = i >> 1, mt = i & 1;
    const int e = chunk * 32 + 16 * mt + el;
    const int src_lane = (el >> 2) * 16 + n_out;
    if (e < a.Ceff) slab[((long)tap * a.Ceff + e) * 16 + n_out] = redf[(i * 64 + src_lane) * 4 + (el & 3)] * (H2 ? s_wsc[2] : 1.f);
  }
}

// =======================================================================================
// Input gradient.  MFMA roles: M = 16 pixels, N = 16 SOURCE channels, K = 9 taps x 16 output
// channels.  The dy rows of the block's pixel tile (one-pixel halo) are staged in LDS once;
// there is no K loop over global memory and no barrier after the staging.  Per source-channel
// tile the lanes fetch the positive and the negative weight rows of "their" channel (64-byte
// rows of the HWIO tensor: lanes 0-15 x 4 groups = 1 KB contiguous when the rows are
// consecutive), contract them against the nine dy fragments (one ds_read_b128 per tap), and
// combine both halves with the activation derivative in registers before the single
// read-modify-write of the gradient buffer.
// =======================================================================================
struct DgArgs {
  const float* dy;
  const float* w;
  const float* x;
  const int32_t* inv;
  float* dx;
  int N, H, W, logW, ldx, lddx, C, Ceff, ldy, accumulate;
  int TR, RS, csplit;
};

__device__ __forceinline__ float d16_deriv(int act, float v) {
  if (act == 1) return v > 0.f ? 1.f : 0.f;
  if (act == 2) return v > 0.f ? 1.f : expf(v);
  return 1.f;
}

template <int PT, int ACT, bool PAIRED, bool W8>
__global__ __launch_bounds__(256, 2) void dense16_dgrad_kernel(DgArgs a) {
  extern __shared__ f32x4 smem4[];
  float* sD = reinterpret_cast<float*>(smem4);   // [(TR+2)*RS][24]
  constexpr int ND = PT + 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p = lane & 15, g = lane >> 4;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  const int tiles_per_img = a.H / a.TR;
  const int n = blockIdx.x / tiles_per_img;
  const int r0 = (blockIdx.x - n * tiles_per_img) * a.TR;
  const long img_base = (long)n * a.H * a.W;

  const int dquads = (a.TR + 2) * a.RS * (kDyStride / 4);
  for (int i = tid; i < dquads; i += 256) smem4[i] = zero;
  __syncthreads();
  const int dtotal = (a.TR + 2) * a.W * 4;
#pragma unroll
  for (int it = 0; it < ND; ++it) {
    const int i = it * 256 + tid;
    const int px = i >> 2;
    const int row = px >> a.logW, col = px & (a.W - 1);
    const int ir = r0 - 1 + row;
    if (i < dtotal && (unsigned)ir < (unsigned)a.H)
      *reinterpret_cast<f32x4*>(sD + (row * a.RS + col + 1) * kDyStride + 4 * (i & 3)) =
          *reinterpret_cast<const f32x4*>(a.dy + (img_base + (long)ir * a.W + col) * a.ldy + 4 * (i & 3));
  }
  __syncthreads();

  // fragment base of an M-tile: dy[q - tap][4g ..]
  auto frag_base = [&](int t) {
    const int q0 = (wave * PT + t) * 16;
    int rr, cc;
    if (W8) {
      rr = (q0 >> 3) + (p >> 3);
      cc = p & 7;
    } else {
      rr = q0 >> a.logW;
      cc = (q0 & (a.W - 1)) + p;
    }
    return ((rr + 1) * a.RS + cc + 1) * kDyStride + 4 * g;
  };

  const int nct = (a.C + 15) >> 4;
  const int per = (nct + a.csplit - 1) / a.csplit;
  const int ct_begin = blockIdx.y * per;
  const int ct_end = min(ct_begin + per, nct);
  const long tapstride = (long)a.Ceff * 16;

  auto load_w = [&](int ct, f32x4 (&Wp)[9], f32x4 (&Wn)[9]) {
    const int c = ct * 16 + p;
    const bool ok = c < a.C;
    const int cc = ok ? c : 0;
    const int ep = a.inv ? a.inv[cc] : cc;
    const float* wp = a.w + (long)ep * 16 + 4 * g;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) Wp[tap] = ok ? *reinterpret_cast<const f32x4*>(wp + tap * tapstride) : zero;
    if (PAIRED) {
      const int en = a.inv ? a.inv[a.C + cc] : a.C + cc;
      const float* wn = a.w + (long)en * 16 + 4 * g;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) Wn[tap] = ok ? *reinterpret_cast<const f32x4*>(wn + tap * tapstride) : zero;
    }
  };

  f32x4 Wp0[9], Wn0[9], Wp1[9], Wn1[9];
  if (ct_begin < ct_end) load_w(ct_begin, Wp0, Wn0);
  // MFMA roles are swapped (weights as the row operand): D[m = channel 4g + r][n = pixel p], so a
  // lane ends up with FOUR CONSECUTIVE channels of one pixel -- 16-byte x / dx accesses.
  auto body = [&](int ct, const f32x4 (&Wp)[9], const f32x4 (&Wn)[9]) {
    const int c4 = ct * 16 + 4 * g;
    const bool cok = c4 < a.C;
#pragma unroll 1
    for (int t = 0; t < PT; ++t) {
      f32x4 gp = zero, gn = zero;
      const int abt = frag_base(t);
      const long m = img_base + (long)r0 * a.W + (wave * PT + t) * 16 + p;
      // issue the epilogue's loads first: they land while the 36 / 72 MFMAs run
      f32x4 xv = zero, old = zero;
      if ((ACT != 0 || PAIRED) && cok) xv = *reinterpret_cast<const f32x4*>(a.x + m * a.ldx + c4);
      float* dst = a.dx + m * a.lddx + c4;
      if (a.accumulate && cok) old = *reinterpret_cast<const f32x4*>(dst);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int dy_ = tap / 3 - 1, dx_ = tap % 3 - 1;
        const f32x4 A = *reinterpret_cast<const f32x4*>(sD + abt - (dy_ * a.RS + dx_) * kDyStride);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          gp = __builtin_amdgcn_mfma_f32_16x16x4f32(Wp[tap][i], A[i], gp, 0, 0, 0);
          if (PAIRED) gn = __builtin_amdgcn_mfma_f32_16x16x4f32(Wn[tap][i], A[i], gn, 0, 0, 0);
        }
      }
      if (cok) {
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float t_ = gp[r];
          if (ACT != 0 || PAIRED) {
            t_ = d16_deriv(ACT, xv[r]) * gp[r];
            if (PAIRED) t_ -= d16_deriv(ACT, -xv[r]) * gn[r];
          }
          v[r] = old[r] + t_;
        }
        *reinterpret_cast<f32x4*>(dst) = v;
      }
    }
  };
  for (int ct = ct_begin; ct < ct_end; ct += 2) {
    if (ct + 1 < ct_end) load_w(ct + 1, Wp1, Wn1);
    body(ct, Wp0, Wn0);
    if (ct + 2 < ct_end) load_w(ct + 2, Wp0, Wn0);
    if (ct + 1 < ct_end) body(ct + 1, Wp1, Wn1);
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

bool dense16_enabled() { return true; }

int dense16_fwd(const Dense16Geo& g, const float* x, const float* wT, const float* bias, float* y,
                int ldy, int coff, int accumulate, hipStream_t s, float* amax_out) {
  FwdArgs a;
  a.accumulate = accumulate;
  a.amax = amax_out;
  a.x = x; a.cmap = g.cmap; a.wT = wT; a.bias = bias; a.y = y;
  a.M = g.N * g.H * g.W;
  a.H = g.H; a.W = g.W; a.logW = ilog2i(g.W);
  a.ldx = g.ldx; a.C = g.C; a.Ceff = g.Ceff; a.doubled = g.doubled;
  a.K = 9 * g.Ceff; a.ldy = ldy; a.coff = coff;
  if (g.W >= 8 && g.W <= 64 && g.H * g.W >= 64) {
    // haloed LDS tile: block = TR full rows = 64*PT pixels of one image
    int PT = g.H * g.W >= 256 ? 4 : g.H * g.W / 64;
    while (PT > 1 && (long)g.N * g.H * g.W / (64 * PT) < 512) PT >>= 1;
    FwdLdsArgs l;
    l.accumulate = accumulate;
    l.amax = amax_out;
    l.x = x; l.cmap = g.cmap; l.wT = wT; l.bias = bias; l.y = y;
    l.N = g.N; l.H = g.H; l.W = g.W; l.logW = ilog2i(g.W); l.ldx = g.ldx; l.C = g.C; l.Ceff = g.Ceff;
    l.doubled = g.doubled; l.K = 9 * g.Ceff; l.ldy = ldy; l.coff = coff;
    l.TR = 64 * PT / g.W;
    l.RS = g.W == 8 ? 16 : g.W + 2;
    if (l.TR >= 1 && g.H % l.TR == 0) {
      const size_t lds = (size_t)(l.TR + 2) * l.RS * kPixQuads * 16;
      const dim3 grid(g.N * (g.H / l.TR)), blk(256);
      const bool w8 = g.W == 8;
      // split-precision forward only where it pays: the 64*4-pixel tiles of the 32x32 (and larger) stages
      // (measured 1.18x there; the 16x16 / 8x8 stages are bound by their chunk barriers, not by the matrix pipe)
      const bool x3 = PT == 4;
      const size_t lds3 = (size_t)3 * (l.TR + 2) * l.RS * 32;
#define D16_LAUNCH(PT_, ACT_)                                                                               \
  do {                                                                                                      \
    if (x3) {                                                                                               \
      if (w8) hipLaunchKernelGGL((dense16_fwd_x3_kernel<PT_, ACT_, true>), grid, blk, lds3, s, l);          \
      else hipLaunchKernelGGL((dense16_fwd_x3_kernel<PT_, ACT_, false>), grid, blk, lds3, s, l);            \
    } else if (w8) hipLaunchKernelGGL((dense16_fwd_lds_kernel<PT_, ACT_, true>), grid, blk, lds, s, l);     \
    else hipLaunchKernelGGL((dense16_fwd_lds_kernel<PT_, ACT_, false>), grid, blk, lds, s, l);              \
  } while (0)
#define D16_LAUNCH_ACT(PT_)                       \
  do {                                            \
    if (g.act == 1) D16_LAUNCH(PT_, 1);           \
    else if (g.act == 2) D16_LAUNCH(PT_, 2);      \
    else D16_LAUNCH(PT_, 0);                      \
  } while (0)
      if (PT == 4) D16_LAUNCH_ACT(4);
      else if (PT == 2) D16_LAUNCH_ACT(2);
      else D16_LAUNCH_ACT(1);
#undef D16_LAUNCH_ACT
#undef D16_LAUNCH
      return OTGAN_OK;
    }
  }
  const int tiles = (a.M + 15) / 16;
  const bool sgn = g.doubled || g.cmap != nullptr;
  // enough waves to fill 256 CUs x 4 SIMDs a few times over; fewer, fatter waves when there are plenty
  if (tiles >= 8192) launch_fwd_pt<4>(a, g.act, sgn, (tiles + 15) / 16, s);
  else if (tiles >= 4096) launch_fwd_pt<2>(a, g.act, sgn, (tiles + 7) / 8, s);
  else launch_fwd_pt<1>(a, g.act, sgn, (tiles + 3) / 4, s);
  return OTGAN_OK;
}

// pixel tiles of 16 per wave (workgroup = 64 PT pixels = full rows) of the fp16 x 2 growth kernels (half-height tiles --
// three workgroups per compute unit at 32 x 32 -- measured 4 % slower in round 4)
static int h2_pt(int N, int H, int W) {
  int PT = H * W >= 256 ? 4 : H * W / 64;
  while (PT > 1 && (long)N * H * W / (64 * PT) < 512) PT >>= 1;
  return PT;
}
size_t dense16_h2_filter_bytes(int nsl) { return nsl > 0 ? (size_t)kH2HdrBytes + (size_t)nsl * kH2SliceU16 * 2 : 0; }

bool dense16_h2_shape_ok(int N, int H, int W) {
  if (!(W == 8 || W == 16 || W == 32) || H * W < 64) return false;
  const int PT = h2_pt(N, H, W);
  const int TR = 64 * PT / W;
  return TR >= 1 && H % TR == 0 && (W != 8 || PT == 1);   // (8-wide images: only the one-tile instantiation exists)
}

int dense16_h2_prepare(const float* const* wT, const int* nsl, void* const* out, int count, hipStream_t s) {
  H2PrepArgs a;
  memset(&a, 0, sizeof(a));
  for (int i = 0; i < count; ++i) { a.wT[i] = wT[i]; a.nsl[i] = nsl[i]; a.out[i] = (unsigned char*)out[i]; }
  hipLaunchKernelGGL(dense16_h2_prep_kernel, dim3(count), dim3(1024), 0, s, a);
  return OTGAN_OK;
}

int dense16_fwd_h2(int N, int H, int W, int nsl, const float* x, int ldx, const void* wq, const float* rec, int nrec,
                   float* y, int ldy, int coff, hipStream_t s, float* amax_out) {
  const int PT = h2_pt(N, H, W);
  FwdH2Args a;
  a.x = x; a.wq = (const unsigned char*)wq; a.y = y; a.rec = rec; a.nrec = nrec; a.nsl = nsl;
  a.N = N; a.H = H; a.W = W; a.logW = ilog2i(W); a.ldx = ldx; a.ldy = ldy; a.coff = coff;
  a.TR = 64 * PT / W;
  a.RS = W == 8 ? 16 : W + 2;
  a.amax = amax_out;
  const size_t lds = (size_t)4 * (a.TR + 2) * a.RS * 32 + (size_t)kH2SliceU16 * 2;
  const dim3 grid(N * (H / a.TR)), blk(256);
  const bool w8 = W == 8;
#define D16_H2(PT_, W_) hipLaunchKernelGGL((dense16_fwd_h2_kernel<PT_, W_>), grid, blk, lds, s, a)
  (void)w8;
  if (W == 32) { if (PT == 4) D16_H2(4, 32); else if (PT == 2) D16_H2(2, 32); else D16_H2(1, 32); }
  else if (W == 16) { if (PT == 4) D16_H2(4, 16); else if (PT == 2) D16_H2(2, 16); else D16_H2(1, 16); }
  else D16_H2(1, 8);
#undef D16_H2
  return OTGAN_OK;
}

bool dense16_chain_bwd_h2(int N, int H, int W, int nslices, float* g, int ldg, const float* x, int ldx,
                          const void* const* filters, const float* rec0, float* slice_records, hipStream_t s) {
  // 8 x 8 only: at 16 x 16 one workgroup per image (256 of them, PT = 4) measured 124.6 us per chain against 7 x 15.3 us for the
  // per-slice kernels on half images (512 workgroups, PT = 2) -- profiles/r06_pmc_kernels_densenet.txt; 8 x 8: 45.6 against 64 us
  if (!(H == W && W == 8) || nslices < 2 || nslices > 17) return false;
  ChainBwdH2Args a;
  memset(&a, 0, sizeof(a));
  a.g = g; a.x = x; a.rec0 = rec0; a.slice_rec = slice_records; a.nslices = nslices;
  a.N = N; a.H = H; a.W = W; a.ldg = ldg; a.ldx = ldx;
  for (int c = 0; c + 1 < nslices; ++c) a.wq[c] = (const unsigned char*)filters[c];
  const int TR = H, RS = W == 8 ? 16 : W + 2;
  const size_t lds = (size_t)2 * (TR + 2) * RS * 32 + (size_t)kH2SliceU16 * 2;
  if (W == 8) hipLaunchKernelGGL((dense16_chain_bwd_h2_kernel<1, 8>), dim3(N), dim3(256), lds, s, a);
  else hipLaunchKernelGGL((dense16_chain_bwd_h2_kernel<4, 16>), dim3(N), dim3(256), lds, s, a);
  return true;
}
// the whole chain of a group in ONE launch where a workgroup covers an image (dense16_chain_fwd_h2_kernel); false: not this shape
bool dense16_chain_fwd_h2(int N, int H, int W, int nslices, float* buf, int ld, const void* const* filters, float* records,
                          hipStream_t s) {
  if (!(H == W && (W == 8 || W == 16)) || nslices < 2 || nslices > 17) return false;
  ChainH2Args a;
  memset(&a, 0, sizeof(a));
  a.buf = buf; a.rec = records; a.nslices = nslices; a.N = N; a.H = H; a.W = W; a.ld = ld;
  for (int j = 1; j < nslices; ++j) a.wq[j - 1] = (const unsigned char*)filters[j - 1];
  const int PT = W == 8 ? 1 : 4, TR = H, RS = W == 8 ? 16 : W + 2;
  const size_t lds = (size_t)4 * (TR + 2) * RS * 32 + (size_t)kH2SliceU16 * 2;
  if (W == 8) hipLaunchKernelGGL((dense16_chain_fwd_h2_kernel<1, 8>), dim3(N), dim3(256), lds, s, a);       // 40 KB of LDS
  else hipLaunchKernelGGL((dense16_chain_fwd_h2_kernel<4, 16>), dim3(N), dim3(256), lds, s, a);             // 61 KB
  (void)PT;
  return true;
}

size_t dense16_h2_bwd_filter_bytes(int nsl) { return dense16_h2_filter_bytes(nsl); }

int dense16_h2_bwd_prepare(const Dense16BwdPair* pairs, int npairs, const void* const* allfwd, int nall, hipStream_t s) {
  for (int base = 0; base < npairs; base += 64) {
    H2BwdPrepArgs a;
    memset(&a, 0, sizeof(a));
    const int cnt = npairs - base < 64 ? npairs - base : 64;
    for (int i = 0; i < cnt; ++i) {
      const Dense16BwdPair& pr = pairs[base + i];
      a.w[i] = pr.w; a.fwd[i] = (const unsigned char*)pr.fwd; a.nch[i] = pr.nch; a.cidx[i] = pr.cidx;
      a.hdr[i] = (unsigned char*)pr.out_base;
      a.out[i] = (unsigned char*)pr.out_base + kH2HdrBytes + (size_t)pr.pair_index * kH2SliceU16 * 2;
      a.first[i] = pr.pair_index == 0;
    }
    a.nall = nall;
    for (int i = 0; i < nall; ++i) a.allfwd[i] = (const unsigned char*)allfwd[i];
    hipLaunchKernelGGL(dense16_h2_bwd_prep_kernel, dim3(cnt), dim3(256), 0, s, a);
  }
  return OTGAN_OK;
}

int dense16_bwd_h2(int N, int H, int W, int nsl, const float* g, int ldg, const void* wq, const float* x, int ldx, float* dx,
                   const float* rec0, int nrec0, const float* rec1, int nrec1, hipStream_t s, float* amax_out) {
  const int PT = h2_pt(N, H, W);
  BwdH2Args a;
  a.g = g; a.wq = (const unsigned char*)wq; a.x = x; a.dx = dx;
  a.rec0 = rec0; a.rec1 = rec1 ? rec1 : rec0; a.nrec0 = nrec0; a.nrec1 = rec1 ? nrec1 : 0; a.nsl = nsl;
  a.N = N; a.H = H; a.W = W; a.logW = ilog2i(W); a.ldg = ldg; a.ldx = ldx;
  a.TR = 64 * PT / W;
  a.RS = W == 8 ? 16 : W + 2;
  a.amax = amax_out;
  const size_t lds = (size_t)2 * (a.TR + 2) * a.RS * 32 + (size_t)kH2SliceU16 * 2;
  const dim3 grid(N * (H / a.TR)), blk(256);
  const bool w8 = W == 8;
#define D16_B2(PT_, W_) hipLaunchKernelGGL((dense16_bwd_h2_kernel<PT_, W_>), grid, blk, lds, s, a)
  (void)w8;
  if (W == 32) { if (PT == 4) D16_B2(4, 32); else if (PT == 2) D16_B2(2, 32); else D16_B2(1, 32); }
  else if (W == 16) { if (PT == 4) D16_B2(4, 16); else if (PT == 2) D16_B2(2, 16); else D16_B2(1, 16); }
  else D16_B2(1, 8);
#undef D16_B2
  return OTGAN_OK;
}

Dense16Tiling dense16_tiling(int N, int H, int W, int Ceff) {
  Dense16Tiling t;
  memset(&t, 0, sizeof(t));
  if (!(W >= 8 && W <= 64 && H * W >= 64)) return t;
  int PT = H * W >= 256 ? 4 : H * W / 64;
  while (PT > 1 && (long)N * H * W / (64 * PT) < 512) PT >>= 1;
  t.PT = PT;
  t.TR = 64 * PT / W;
  if (t.TR < 1 || H % t.TR) return t;
  t.RS = W == 8 ? 16 : W + 2;
  t.tiles = N * (H / t.TR);
  t.nchunk = (Ceff + 31) / 32;
  // workgroups of the weight-gradient grid (pixel splits x channel chunks).  1536 until round 4 (three rounds of the 512
  // resident ones, tuned for the fp32-pipe kernel); with the fp16-pipe kernel the per-workgroup epilogue (four-wave
  // reduction through LDS, slab write) and the slab reduction weigh more: 768 measured best (wgrad + slab_reduce of a DenseNet
  // step 2.52 -> 2.27 ms; 512: 2.41, 1024: 2.46)
  constexpr int wg_target = 768;
  int want = (wg_target + t.nchunk - 1) / t.nchunk;
  if (want > t.tiles) want = t.tiles;
  if (want < 1) want = 1;
  t.tiles_per_split = (t.tiles + want - 1) / want;
  t.nsplit = (t.tiles + t.tiles_per_split - 1) / t.tiles_per_split;
  t.ok = 1;
  return t;
}

int dense16_wgrad(const Dense16Geo& g, const float* x, const float* dy, int ldy, int coff, float* slabs,
                  hipStream_t s, const float* x_rec, int x_nrec, const float* dy_rec, int dy_nrec) {
  const Dense16Tiling t = dense16_tiling(g.N, g.H, g.W, g.Ceff);
  if (!t.ok) {
    otgan_set_error("dense16 wgrad: unsupported geometry");
    return OTGAN_ERR_UNSUPPORTED;
  }
  WgArgs a;
  a.x = x; a.cmap = g.cmap; a.dy = dy + coff; a.slabs = slabs;
  a.N = g.N; a.H = g.H; a.W = g.W; a.logW = ilog2i(g.W); a.ldx = g.ldx; a.C = g.C; a.Ceff = g.Ceff;
  a.doubled = g.doubled; a.ldy = ldy;
  a.TR = t.TR; a.RS = t.RS; a.tiles = t.tiles; a.tiles_per_split = t.tiles_per_split;
  a.slab_elems = (long)9 * g.Ceff * 16;
  const bool h2 = x_rec && dy_rec && x_nrec > 0 && dy_nrec > 0 && g.W >= 4;
  a.x_rec = x_rec; a.dy_rec = dy_rec; a.x_nrec = x_nrec; a.dy_nrec = dy_nrec;
  size_t lds = ((size_t)64 * t.PT * kAStride + (size_t)(t.TR + 2) * t.RS * kDyStride) * 4;
  if (lds < 18 * 64 * 16) lds = 18 * 64 * 16;   // wave-reduction scratch
  a.xmap = t.nchunk > 1 ? 1 : 0;
  a.nsplit = t.nsplit; a.nchunk = t.nchunk;
  const dim3 grid = a.xmap ? dim3(8u * ((t.nsplit + 7) / 8) * t.nchunk, 1) : dim3(t.nsplit, t.nchunk);
  const dim3 blk(256);
#define D16_WG2(PT_, H2_)                                                                             \
  do {                                                                                                \
    if (g.act == 1) hipLaunchKernelGGL((dense16_wgrad_kernel<PT_, 1, H2_>), grid, blk, lds, s, a);    \
    else if (g.act == 2) hipLaunchKernelGGL((dense16_wgrad_kernel<PT_, 2, H2_>), grid, blk, lds, s, a); \
    else hipLaunchKernelGGL((dense16_wgrad_kernel<PT_, 0, H2_>), grid, blk, lds, s, a);               \
  } while (0)
#define D16_WG(PT_)          \
  do {                       \
    if (h2) D16_WG2(PT_, true); \
    else D16_WG2(PT_, false);   \
  } while (0)
  if (t.PT == 4) {
    // 72.6 KB of LDS: above the 64 KB default of dynamic shared memory
    static const bool once = [] {
      hipFuncSetAttribute(reinterpret_cast<const void*>(&dense16_wgrad_kernel<4, 0, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
      hipFuncSetAttribute(reinterpret_cast<const void*>(&dense16_wgrad_kernel<4, 1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
      hipFuncSetAttribute(reinterpret_cast<const void*>(&dense16_wgrad_kernel<4, 2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
      hipFuncSetAttribute(reinterpret_cast<const void*>(&dense16_wgrad_kernel<4, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
      hipFuncSetAttribute(reinterpret_cast<const void*>(&dense16_wgrad_kernel<4, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
      hipFuncSetAttribute(reinterpret_cast<const void*>(&dense16_wgrad_kernel<4, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
      return true;
    }();
    (void)once;
    D16_WG(4);
  } else if (t.PT == 2) D16_WG(2);
  else D16_WG(1);
#undef D16_WG
#undef D16_WG2
  return OTGAN_OK;
}

int dense16_dgrad(const Dense16Geo& g, const float* dy, int ldy, int coff, const float* w, const float* x,
                  const int32_t* inv, float* dx, int lddx, int accumulate, hipStream_t s) {
  const Dense16Tiling t = dense16_tiling(g.N, g.H, g.W, g.Ceff);
  if (!t.ok) {
    otgan_set_error("dense16 dgrad: unsupported geometry");
    return OTGAN_ERR_UNSUPPORTED;
  }
  DgArgs a;
  a.dy = dy + coff; a.w = w; a.x = x; a.inv = inv; a.dx = dx;
  a.N = g.N; a.H = g.H; a.W = g.W; a.logW = ilog2i(g.W); a.ldx = g.ldx; a.lddx = lddx; a.C = g.C;
  a.Ceff = g.Ceff; a.ldy = ldy; a.accumulate = accumulate;
  a.TR = t.TR; a.RS = t.RS;
  const int nct = (g.C + 15) / 16;
  int cs = (1024 + t.tiles - 1) / t.tiles;      // spread channel tiles over blockIdx.y on small grids
  if (cs > nct) cs = nct;
  if (cs < 1) cs = 1;
  a.csplit = cs;
  const size_t lds = (size_t)(t.TR + 2) * t.RS * kDyStride * 4;
  const dim3 grid(t.tiles, cs), blk(256);
  const bool w8 = g.W == 8;
#define D16_DG3(PT_, ACT_, PAIR_)                                                                       \
  do {                                                                                                  \
    if (w8) hipLaunchKernelGGL((dense16_dgrad_kernel<PT_, ACT_, PAIR_, true>), grid, blk, lds, s, a);   \
    else hipLaunchKernelGGL((dense16_dgrad_kernel<PT_, ACT_, PAIR_, false>), grid, blk, lds, s, a);     \
  } while (0)
#define D16_DG(PT_)                                                   \
  do {                                                                \
    if (g.doubled) {                                                  \
      if (g.act == 2) D16_DG3(PT_, 2, true);                          \
      else D16_DG3(PT_, 1, true);                                     \
    } else if (g.act == 1) D16_DG3(PT_, 1, false);                    \
    else if (g.act == 2) D16_DG3(PT_, 2, false);                      \
    else D16_DG3(PT_, 0, false);                                      \
  } while (0)
  if (t.PT == 4) D16_DG(4);
  else if (t.PT == 2) D16_DG(2);
  else D16_DG(1);
#undef D16_DG
#undef D16_DG3
  return OTGAN_OK;
}
