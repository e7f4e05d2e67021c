// conv.hip -- NHWC implicit-GEMM convolution on the fp32 MFMA engine (gfx950):
// forward, data gradient and weight gradient, with the reference's pre-activation
// (CReLU / CELU / ELU / ReLU over a list of inputs), 2x nearest-neighbour upsampling and TF
// 'SAME' padding fused into the operand gathers.
// Replaces reference utils/nn.py:190-206 (pre-activation), :234-241 (conv), :327-338.
//
// GEMM views (K index always = (tap, channel) or pixels):
//   fwd   : Y[pixel, co]        = sum_{tap, d}  A(x)[pixel+tap, d] * W[tap, d, co]
//   dgrad : dXe[in-pixel, d]    = sum_{tap, co} dY[in-pixel-tap, co] * W[tap, d, co]
//           (stride 2: one launch per input-parity class, only the taps that hit it)
//           then dX[c] = act'(x_c) (dXe[d+(c)] , dXe[d-(c)]) in the epilogue
//   wgrad : dW[tap, d, co]      = sum_{pixel}  A(x)[pixel+tap, d] * dY[pixel, co]
//           (split over pixels, slabs reduced afterwards)
#include "gemm_tile.h"
#include "../../include/otgan.h"

namespace {

using CfgMain = GemmCfg<2, 2, 2, 2, 16>;    // 128 x 128 block tile
using CfgNarrow = GemmCfg<4, 1, 2, 1, 16>;  // 256 x 32 block tile (Cout / Cin <= 32)

constexpr int kMaxTaps = 25;

struct Taps {
  int n;
  short dh[kMaxTaps];
  short dw[kMaxTaps];
  int boff[kMaxTaps];  // offset of this tap's weight block in the B operand
};

// Source of the gathered ("A") operand: an NHWC tensor read through tap offsets, optional 2x
// nearest upsampling, channel map (list interleave + sign) and activation.
struct GatherA {
  const float* x;
  int ldx;
  int H, W;      // stored dims
  int logUp;     // virtual dims = H << logUp
  int logGH, logGW;  // the row grid [*, GH, GW] (powers of two)
  int Mtot;      // rows = N * GH * GW
  int sa;        // virtual coord = grid coord * sa + tap offset
  int Ck;        // effective channels per tap
  const int* cmap;
  int Creal;
  int doubled;   // default map: d < Creal -> +x[d], else -x[d - Creal]
  int act;       // 0 none, 1 relu, 2 elu -- applied to sign * x
};

__device__ __forceinline__ void map_channel(const GatherA& g, int d, int& c, float& sgn) {
  if (g.cmap) {
    const int v = g.cmap[d];
    c = v & 0x7fffffff;
    sgn = v < 0 ? -1.f : 1.f;
  } else if (g.doubled && d >= g.Creal) {
    c = d - g.Creal;
    sgn = -1.f;
  } else {
    c = d;
    sgn = 1.f;
  }
}

__device__ __forceinline__ float act_apply(int act, float v) {
  if (act == 1) return fmaxf(v, 0.f);
  if (act == 2) return v > 0.f ? v : expm1f(v);
  return v;
}
// derivative of act at v
__device__ __forceinline__ float act_deriv(int act, float v) {
  if (act == 1) return v > 0.f ? 1.f : 0.f;
  if (act == 2) return v > 0.f ? 1.f : expf(v);
  return 1.f;
}

// ---------------------------------------------------------------------------------------
// A loaders (rows = pixels, K = (tap, channel), K contiguous in memory)
// ---------------------------------------------------------------------------------------
template <class Cfg, int BR, bool VEC>
struct ConvALoader {
  static constexpr int BK = Cfg::BK;
  static constexpr int LD = BR + KPad<BK>::value;
  static constexpr int FLOATS = BK * LD;
  static constexpr int CPR = BK / 4;
  static constexpr int RPP = Cfg::THREADS / CPR;
  static constexpr int PASSES = (BR + RPP - 1) / RPP;
  const GatherA& g;
  const Taps& taps;
  int pixbase[PASSES], ia[PASSES], ib[PASSES];
  float4 reg[PASSES];
  int vH, vW, Ktot;

  __device__ __forceinline__ ConvALoader(const GatherA& g_, const Taps& t_) : g(g_), taps(t_) {}

  __device__ __forceinline__ void init(int m0) {
    const int r0 = threadIdx.x / CPR;
    vH = g.H << g.logUp;
    vW = g.W << g.logUp;
    Ktot = taps.n * g.Ck;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int r = r0 + p * RPP;
      const int m = m0 + r;
      if (r < BR && m < g.Mtot) {
        const int b = m & ((1 << g.logGW) - 1);
        const int a = (m >> g.logGW) & ((1 << g.logGH) - 1);
        const int n = m >> (g.logGW + g.logGH);
        ia[p] = a * g.sa;
        ib[p] = b * g.sa;
        pixbase[p] = n * g.H * g.W;
      } else {
        ia[p] = -100000;
        ib[p] = -100000;
        pixbase[p] = 0;
      }
    }
  }

  __device__ __forceinline__ float fetch1(int p, int k) const {
    if (k >= Ktot) return 0.f;
    const int t = k / g.Ck;
    const int d = k - t * g.Ck;
    const int ih = ia[p] + taps.dh[t], iw = ib[p] + taps.dw[t];
    if ((unsigned)ih >= (unsigned)vH || (unsigned)iw >= (unsigned)vW) return 0.f;
    int c;
    float sgn;
    map_channel(g, d, c, sgn);
    const long pix = (long)pixbase[p] + (long)(ih >> g.logUp) * g.W + (iw >> g.logUp);
    return act_apply(g.act, sgn * g.x[pix * g.ldx + c]);
  }

  __device__ __forceinline__ void load(int kt) {
    const int c4 = threadIdx.x % CPR;
    const int k = kt * BK + 4 * c4;
    if (VEC) {
      // Ck % BK == 0: the whole K tile lies inside one tap
      const int t = (kt * BK) / g.Ck;
      const int d = k - t * g.Ck;
      int sc;
      float sgn;
      map_channel(g, d, sc, sgn);
      const int dh = taps.dh[t], dw = taps.dw[t];
#pragma unroll
      for (int p = 0; p < PASSES; ++p) {
        const int ih = ia[p] + dh, iw = ib[p] + dw;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((unsigned)ih < (unsigned)vH && (unsigned)iw < (unsigned)vW) {
          const long pix = (long)pixbase[p] + (long)(ih >> g.logUp) * g.W + (iw >> g.logUp);
          v = *reinterpret_cast<const float4*>(g.x + pix * g.ldx + sc);
          v.x = act_apply(g.act, sgn * v.x);
          v.y = act_apply(g.act, sgn * v.y);
          v.z = act_apply(g.act, sgn * v.z);
          v.w = act_apply(g.act, sgn * v.w);
        }
        reg[p] = v;
      }
    } else {
#pragma unroll
      for (int p = 0; p < PASSES; ++p) {
        float4 v;
        v.x = fetch1(p, k);
        v.y = fetch1(p, k + 1);
        v.z = fetch1(p, k + 2);
        v.w = fetch1(p, k + 3);
        reg[p] = v;
      }
    }
  }

  __device__ __forceinline__ void store(float* t) const {
    const int c4 = threadIdx.x % CPR, r0 = threadIdx.x / CPR;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int r = r0 + p * RPP;
      if (r < BR) {
        float* d = t + (4 * c4) * LD + r;
        d[0] = reg[p].x;
        d[LD] = reg[p].y;
        d[2 * LD] = reg[p].z;
        d[3 * LD] = reg[p].w;
      }
    }
  }
};

// Weight ("B") operand: element (n, k=(tap, c)) at w[boff[tap] + row(n)*ldbn + c].
struct WeightB {
  const float* w;
  long ldbn;
  int Nvalid;       // valid logical rows
  int Ck;           // channels per tap in K
  int paired;       // 1: block rows map to (+c, -c) channel pairs (dgrad through CReLU/CELU)
  int Creal;        // paired: number of real channels
  const int* inv;   // paired: effective index of +c (inv[c]) and -c (inv[Creal + c]); nullable
};

template <class Cfg, int BR, bool VEC>
struct ConvBLoader {
  static constexpr int BK = Cfg::BK;
  static constexpr int LD = BR + KPad<BK>::value;
  static constexpr int FLOATS = BK * LD;
  static constexpr int CPR = BK / 4;
  static constexpr int RPP = Cfg::THREADS / CPR;
  static constexpr int PASSES = (BR + RPP - 1) / RPP;
  const WeightB& b;
  const Taps& taps;
  long rowoff[PASSES];  // row(n) * ldbn, or -1 when the row is invalid
  float4 reg[PASSES];
  int Ktot;

  __device__ __forceinline__ ConvBLoader(const WeightB& b_, const Taps& t_) : b(b_), taps(t_) {}

  // nblk: index of the N tile
  __device__ __forceinline__ void init(int nblk) {
    const int r0 = threadIdx.x / CPR;
    Ktot = taps.n * b.Ck;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int r = r0 + p * RPP;
      long off = -1;
      if (r < BR) {
        if (b.paired) {
          // block tile = 64 real channels; wave column wn owns 32 of them; its two 32-wide
          // MFMA column tiles hold the (+c) and the (-c) effective channels.
          const int wn = r >> 6, nt = (r >> 5) & 1, j = r & 31;
          const int c = nblk * 64 + wn * 32 + j;
          if (c < b.Creal) {
            const int d = b.inv ? b.inv[nt * b.Creal + c] : (nt * b.Creal + c);
            off = (long)d * b.ldbn;
          }
        } else {
          const int n = nblk * BR + r;
          if (n < b.Nvalid) off = (long)n * b.ldbn;
        }
      }
      rowoff[p] = off;
    }
  }

  __device__ __forceinline__ void load(int kt) {
    const int c4 = threadIdx.x % CPR;
    const int k = kt * BK + 4 * c4;
    if (VEC) {
      const int t = (kt * BK) / b.Ck;
      const int d = k - t * b.Ck;
      const float* base = b.w + taps.boff[t] + d;
#pragma unroll
      for (int p = 0; p < PASSES; ++p) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rowoff[p] >= 0) v = *reinterpret_cast<const float4*>(base + rowoff[p]);
        reg[p] = v;
      }
    } else {
#pragma unroll
      for (int p = 0; p < PASSES; ++p) {
        float e[4] = {0.f, 0.f, 0.f, 0.f};
        if (rowoff[p] >= 0) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int kk = k + q;
            if (kk < Ktot) {
              const int t = kk / b.Ck;
              const int d = kk - t * b.Ck;
              e[q] = b.w[taps.boff[t] + rowoff[p] + d];
            }
          }
        }
        reg[p] = make_float4(e[0], e[1], e[2], e[3]);
      }
    }
  }

  __device__ __forceinline__ void store(float* t) const {
    const int c4 = threadIdx.x % CPR, r0 = threadIdx.x / CPR;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int r = r0 + p * RPP;
      if (r < BR) {
        float* d = t + (4 * c4) * LD + r;
        d[0] = reg[p].x;
        d[LD] = reg[p].y;
        d[2 * LD] = reg[p].z;
        d[3 * LD] = reg[p].w;
      }
    }
  }
};

// ---------------------------------------------------------------------------------------
// fwd / dgrad kernel
// ---------------------------------------------------------------------------------------
enum { EPI_FWD = 0, EPI_DG_PLAIN = 1, EPI_DG_ACT = 2, EPI_DG_PAIR = 3 };

struct EpiArgs {
  float* out;
  int ldo, coff;
  int so, oa, ob;        // out pixel = (n, a*so + oa, b*so + ob) on the [OHf, OWf] grid
  int OHf, OWf;
  const float* bias;     // fwd
  int ncols;             // valid output columns (fwd: Cout; dgrad: real channels)
  int accumulate;
  const float* xsrc;     // dgrad: layer input (activation derivative), stored resolution
  int ldxs, xH, xW, logUpX;
  int act;               // 1 relu-type, 2 elu-type
};

template <class Cfg, bool VEC, int EPI>
__global__ __launch_bounds__(Cfg::THREADS) void conv_igemm_kernel(GatherA g, Taps taps, WeightB wb,
                                                                 EpiArgs e) {
  using LA = ConvALoader<Cfg, Cfg::BM, VEC>;
  using LB = ConvBLoader<Cfg, Cfg::BN, VEC>;
  __shared__ __attribute__((aligned(16))) float smem[2 * LA::FLOATS + 2 * LB::FLOATS];
  const int m0 = blockIdx.x * Cfg::BM;
  const int nblk = blockIdx.y;
  LA la(g, taps);
  LB lb(wb, taps);
  la.init(m0);
  lb.init(nblk);
  f32x16 acc[Cfg::MT][Cfg::NT];
  zero_acc<Cfg>(acc);
  const int nkt = (taps.n * g.Ck + Cfg::BK - 1) / Cfg::BK;
  gemm_mainloop<Cfg>(la, lb, nkt, smem, acc);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
  const int li = lane & 31, lh = lane >> 5;
  const int gmask = (1 << g.logGW) - 1, hmask = (1 << g.logGH) - 1;
#pragma unroll
  for (int mt = 0; mt < Cfg::MT; ++mt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wm * Cfg::MT * 32 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int m = m0 + row;
      if (m >= g.Mtot) continue;
      const int b = m & gmask, a = (m >> g.logGW) & hmask, n = m >> (g.logGW + g.logGH);
      const int oh = a * e.so + e.oa, ow = b * e.so + e.ob;
      const long opix = ((long)n * e.OHf + oh) * e.OWf + ow;
      if (EPI == EPI_DG_PAIR) {
        static_assert(EPI != EPI_DG_PAIR || Cfg::NT == 2, "paired epilogue needs NT == 2");
        const int c = nblk * 64 + wn * 32 + li;
        if (c < e.ncols) {
          const long xpix = ((long)n * e.xH + (oh >> e.logUpX)) * e.xW + (ow >> e.logUpX);
          const float xv = e.xsrc[xpix * e.ldxs + c];
          // d/dx [act(x) ; act(-x)] . [g+ ; g-] = act'(x) g+ - act'(-x) g-
          const float gp = acc[mt][0][r], gn = acc[mt][Cfg::NT - 1][r];
          const float v = act_deriv(e.act, xv) * gp - act_deriv(e.act, -xv) * gn;
          float* dst = e.out + opix * e.ldo + e.coff + c;
          *dst = e.accumulate ? (*dst + v) : v;
        }
      } else {
#pragma unroll
        for (int nt = 0; nt < Cfg::NT; ++nt) {
          const int col = nblk * Cfg::BN + wn * Cfg::NT * 32 + nt * 32 + li;
          if (col >= e.ncols) continue;
          float v = acc[mt][nt][r];
          if (EPI == EPI_FWD) {
            v += e.bias ? e.bias[col] : 0.f;
          } else if (EPI == EPI_DG_ACT) {
            const long xpix = ((long)n * e.xH + (oh >> e.logUpX)) * e.xW + (ow >> e.logUpX);
            v *= act_deriv(e.act, e.xsrc[xpix * e.ldxs + col]);
          }
          float* dst = e.out + opix * e.ldo + e.coff + col;
          *dst = (EPI != EPI_FWD && e.accumulate) ? (*dst + v) : v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// wgrad kernels
// ---------------------------------------------------------------------------------------
struct WgradArgs {
  const float* dy;   // already offset to the layer's channel range
  int ldy, Cout;
  float* slab;       // [nsplit][Ktot * Cout]
  int kt_per_split;  // pixel tiles (of BK) per split
  int tiles_n;
  long slab_stride;
};

// rows = effective channel d within tap blockIdx.z; float4 along d (needs Ck % 4 == 0,
// contiguous cmap quads) and along co (Cout % 4 == 0, ldy % 4 == 0).
template <class Cfg>
struct WgALoaderV {
  static constexpr int BK = Cfg::BK, BR = Cfg::BM;
  static constexpr int LD = BR + 4;
  static constexpr int FLOATS = BK * LD;
  static constexpr int CPK = BR / 4;
  static constexpr int KPP = Cfg::THREADS / CPK;
  static constexpr int PASSES = (BK + KPP - 1) / KPP;
  const GatherA& g;
  int sc, dh, dw, vH, vW, m_begin;
  float sgn;
  bool rowok;
  float4 reg[PASSES];
  __device__ __forceinline__ WgALoaderV(const GatherA& g_) : g(g_) {}
  __device__ __forceinline__ void init(int d0, int tap_dh, int tap_dw, int mb) {
    const int c = threadIdx.x % CPK;
    const int d = d0 + 4 * c;
    rowok = d < g.Ck;
    sc = 0;
    sgn = 1.f;
    if (rowok) map_channel(g, d, sc, sgn);
    dh = tap_dh;
    dw = tap_dw;
    vH = g.H << g.logUp;
    vW = g.W << g.logUp;
    m_begin = mb;
  }
  __device__ __forceinline__ void load(int kt) {
    const int k0 = threadIdx.x / CPK;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int kk = k0 + p * KPP;
      const int m = m_begin + kt * BK + kk;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (kk < BK && rowok && m < g.Mtot) {
        const int b = m & ((1 << g.logGW) - 1);
        const int a = (m >> g.logGW) & ((1 << g.logGH) - 1);
        const int n = m >> (g.logGW + g.logGH);
        const int ih = a * g.sa + dh, iw = b * g.sa + dw;
        if ((unsigned)ih < (unsigned)vH && (unsigned)iw < (unsigned)vW) {
          const long pix = (long)n * g.H * g.W + (long)(ih >> g.logUp) * g.W + (iw >> g.logUp);
          v = *reinterpret_cast<const float4*>(g.x + pix * g.ldx + sc);
          v.x = act_apply(g.act, sgn * v.x);
          v.y = act_apply(g.act, sgn * v.y);
          v.z = act_apply(g.act, sgn * v.z);
          v.w = act_apply(g.act, sgn * v.w);
        }
      }
      reg[p] = v;
    }
  }
  __device__ __forceinline__ void store(float* t) const {
    const int c = threadIdx.x % CPK, k0 = threadIdx.x / CPK;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int kk = k0 + p * KPP;
      if (kk < BK) *reinterpret_cast<float4*>(t + kk * LD + 4 * c) = reg[p];
    }
  }
};

template <class Cfg>
struct WgBLoaderV {
  static constexpr int BK = Cfg::BK, BR = Cfg::BN;
  static constexpr int LD = BR + 4;
  static constexpr int FLOATS = BK * LD;
  static constexpr int CPK = BR / 4;
  static constexpr int KPP = (Cfg::THREADS / CPK);
  static constexpr int PASSES = (BK + KPP - 1) / KPP;
  const float* dy;
  int ldy, Cout, Mtot, co, m_begin;
  float4 reg[PASSES];
  __device__ __forceinline__ void init(const WgradArgs& a, int co0, int Mtot_, int mb) {
    dy = a.dy; ldy = a.ldy; Cout = a.Cout; Mtot = Mtot_;
    co = co0 + 4 * (threadIdx.x % CPK);
    m_begin = mb;
  }
  __device__ __forceinline__ void load(int kt) {
    const int k0 = threadIdx.x / CPK;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int kk = k0 + p * KPP;
      const int m = m_begin + kt * BK + kk;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (kk < BK && m < Mtot && co < Cout)
        v = *reinterpret_cast<const float4*>(dy + (long)m * ldy + co);
      reg[p] = v;
    }
  }
  __device__ __forceinline__ void store(float* t) const {
    const int c = threadIdx.x % CPK, k0 = threadIdx.x / CPK;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int kk = k0 + p * KPP;
      if (kk < BK) *reinterpret_cast<float4*>(t + kk * LD + 4 * c) = reg[p];
    }
  }
};

template <class Cfg>
__global__ __launch_bounds__(Cfg::THREADS) void conv_wgrad_kernel(GatherA g, Taps taps,
                                                                 WgradArgs a) {
  using LA = WgALoaderV<Cfg>;
  using LB = WgBLoaderV<Cfg>;
  __shared__ __attribute__((aligned(16))) float smem[2 * LA::FLOATS + 2 * LB::FLOATS];
  const int tm = blockIdx.x / a.tiles_n, tn = blockIdx.x % a.tiles_n;
  const int split = blockIdx.y, t = blockIdx.z;
  const int d0 = tm * Cfg::BM, co0 = tn * Cfg::BN;
  const int nkt_total = (g.Mtot + Cfg::BK - 1) / Cfg::BK;
  const int kt0 = split * a.kt_per_split;
  int nkt = nkt_total - kt0;
  if (nkt > a.kt_per_split) nkt = a.kt_per_split;
  LA la(g);
  LB lb;
  la.init(d0, taps.dh[t], taps.dw[t], kt0 * Cfg::BK);
  lb.init(a, co0, g.Mtot, kt0 * Cfg::BK);
  f32x16 acc[Cfg::MT][Cfg::NT];
  zero_acc<Cfg>(acc);
  gemm_mainloop<Cfg>(la, lb, nkt, smem, acc);
  float* out = a.slab + (long)split * a.slab_stride + (long)t * g.Ck * a.Cout;
  const int Ck = g.Ck, Cout = a.Cout;
  foreach_acc<Cfg>(acc, [&](int r, int c, int, int, int, float v) {
    const int d = d0 + r, co = co0 + c;
    if (d < Ck && co < Cout) out[(long)d * Cout + co] = v;
  });
}

// Generic scalar variant: rows = flattened (tap, d) index over ALL taps; any Ck / Cout / ld.
template <class Cfg>
struct WgALoaderS {
  static constexpr int BK = Cfg::BK, BR = Cfg::BM;
  static constexpr int LD = BR + 4;
  static constexpr int FLOATS = BK * LD;
  static constexpr int CPK = BR / 4;
  static constexpr int KPP = Cfg::THREADS / CPK;
  static constexpr int PASSES = (BK + KPP - 1) / KPP;
  const GatherA& g;
  int sc[4], dh[4], dw[4];
  float sgn[4];
  bool ok[4];
  int vH, vW, m_begin;
  float4 reg[PASSES];
  __device__ __forceinline__ WgALoaderS(const GatherA& g_) : g(g_) {}
  __device__ __forceinline__ void init(int r0, const Taps& taps, int mb) {
    const int c = threadIdx.x % CPK;
    const int Rtot = taps.n * g.Ck;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = r0 + 4 * c + q;
      ok[q] = r < Rtot;
      sc[q] = 0; sgn[q] = 1.f; dh[q] = 0; dw[q] = 0;
      if (ok[q]) {
        const int t = r / g.Ck, d = r - t * g.Ck;
        map_channel(g, d, sc[q], sgn[q]);
        dh[q] = taps.dh[t];
        dw[q] = taps.dw[t];
      }
    }
    vH = g.H << g.logUp;
    vW = g.W << g.logUp;
    m_begin = mb;
  }
  __device__ __forceinline__ void load(int kt) {
    const int k0 = threadIdx.x / CPK;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int kk = k0 + p * KPP;
      const int m = m_begin + kt * BK + kk;
      float e[4] = {0.f, 0.f, 0.f, 0.f};
      if (kk < BK && m < g.Mtot) {
        const int b = m & ((1 << g.logGW) - 1);
        const int a = (m >> g.logGW) & ((1 << g.logGH) - 1);
        const int n = m >> (g.logGW + g.logGH);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int ih = a * g.sa + dh[q], iw = b * g.sa + dw[q];
          if (ok[q] && (unsigned)ih < (unsigned)vH && (unsigned)iw < (unsigned)vW) {
            const long pix = (long)n * g.H * g.W + (long)(ih >> g.logUp) * g.W + (iw >> g.logUp);
            e[q] = act_apply(g.act, sgn[q] * g.x[pix * g.ldx + sc[q]]);
          }
        }
      }
      reg[p] = make_float4(e[0], e[1], e[2], e[3]);
    }
  }
  __device__ __forceinline__ void store(float* t) const {
    const int c = threadIdx.x % CPK, k0 = threadIdx.x / CPK;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int kk = k0 + p * KPP;
      if (kk < BK) *reinterpret_cast<float4*>(t + kk * LD + 4 * c) = reg[p];
    }
  }
};

template <class Cfg>
struct WgBLoaderS {
  static constexpr int BK = Cfg::BK, BR = Cfg::BN;
  static constexpr int LD = BR + 4;
  static constexpr int FLOATS = BK * LD;
  static constexpr int CPK = BR / 4;
  static constexpr int KPP = (Cfg::THREADS / CPK);
  static constexpr int PASSES = (BK + KPP - 1) / KPP;
  const float* dy;
  int ldy, Cout, Mtot, co, m_begin;
  float4 reg[PASSES];
  __device__ __forceinline__ void init(const WgradArgs& a, int co0, int Mtot_, int mb) {
    dy = a.dy; ldy = a.ldy; Cout = a.Cout; Mtot = Mtot_;
    co = co0 + 4 * (threadIdx.x % CPK);
    m_begin = mb;
  }
  __device__ __forceinline__ void load(int kt) {
    const int k0 = threadIdx.x / CPK;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int kk = k0 + p * KPP;
      const int m = m_begin + kt * BK + kk;
      float e[4] = {0.f, 0.f, 0.f, 0.f};
      if (kk < BK && m < Mtot) {
        const float* src = dy + (long)m * ldy;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (co + q < Cout) e[q] = src[co + q];
      }
      reg[p] = make_float4(e[0], e[1], e[2], e[3]);
    }
  }
  __device__ __forceinline__ void store(float* t) const {
    const int c = threadIdx.x % CPK, k0 = threadIdx.x / CPK;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int kk = k0 + p * KPP;
      if (kk < BK) *reinterpret_cast<float4*>(t + kk * LD + 4 * c) = reg[p];
    }
  }
};

template <class Cfg>
__global__ __launch_bounds__(Cfg::THREADS) void conv_wgrad_scalar_kernel(GatherA g, Taps taps,
                                                                        WgradArgs a) {
  using LA = WgALoaderS<Cfg>;
  using LB = WgBLoaderS<Cfg>;
  __shared__ __attribute__((aligned(16))) float smem[2 * LA::FLOATS + 2 * LB::FLOATS];
  const int tm = blockIdx.x / a.tiles_n, tn = blockIdx.x % a.tiles_n;
  const int split = blockIdx.y;
  const int r0 = tm * Cfg::BM, co0 = tn * Cfg::BN;
  const int nkt_total = (g.Mtot + Cfg::BK - 1) / Cfg::BK;
  const int kt0 = split * a.kt_per_split;
  int nkt = nkt_total - kt0;
  if (nkt > a.kt_per_split) nkt = a.kt_per_split;
  LA la(g);
  LB lb;
  la.init(r0, taps, kt0 * Cfg::BK);
  lb.init(a, co0, g.Mtot, kt0 * Cfg::BK);
  f32x16 acc[Cfg::MT][Cfg::NT];
  zero_acc<Cfg>(acc);
  gemm_mainloop<Cfg>(la, lb, nkt, smem, acc);
  float* out = a.slab + (long)split * a.slab_stride;
  const int Rtot = taps.n * g.Ck, Cout = a.Cout;
  foreach_acc<Cfg>(acc, [&](int r, int c, int, int, int, float v) {
    const int rr = r0 + r, co = co0 + c;
    if (rr < Rtot && co < Cout) out[(long)rr * Cout + co] = v;
  });
}

__global__ void slab_reduce_kernel(const float* __restrict__ slab, int nsplit, long n,
                                   float* __restrict__ out) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < nsplit; ++k) s += slab[(long)k * n + i];
    out[i] = s;
  }
}

// dx[n,h,w,c] (+)= sum over the 2x2 replicas of dxv[n,2h+i,2w+j,c]   (backward of NN upsample)
__global__ void pool2_sum_kernel(const float* __restrict__ dxv, int N, int H, int W, int C,
                                 float* __restrict__ dx, int lddx, int accumulate) {
  const long total = (long)N * H * W * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long pix = i / C;
    const int w = (int)(pix % W);
    const int h = (int)((pix / W) % H);
    const int n = (int)(pix / ((long)W * H));
    const long base = (((long)n * 2 * H + 2 * h) * 2 * W + 2 * w) * C + c;
    const float s = dxv[base] + dxv[base + C] + dxv[base + (long)2 * W * C] +
                    dxv[base + (long)2 * W * C + C];
    float* dst = dx + pix * lddx + c;
    *dst = accumulate ? (*dst + s) : s;
  }
}

// ---------------------------------------------------------------------------------------
// host-side geometry
// ---------------------------------------------------------------------------------------
inline int ilog2_exact(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return ((1 << l) == v) ? l : -1;
}
inline bool doubled_act(int a) { return a == OTGAN_ACT_CRELU || a == OTGAN_ACT_CELU; }
inline int act_kind(int a) {  // 0 none, 1 relu-type, 2 elu-type
  if (a == OTGAN_ACT_CRELU || a == OTGAN_ACT_RELU) return 1;
  if (a == OTGAN_ACT_CELU || a == OTGAN_ACT_ELU) return 2;
  return 0;
}

struct Geo {
  int Hin, Win, OH, OW, pad_t, pad_l, Ceff, logUp;
};
int make_geo(const otgan_conv_desc* d, Geo* g) {
  OTGAN_CHECK_ARG(d, "null desc");
  OTGAN_CHECK_ARG(d->N > 0 && d->H > 0 && d->W > 0 && d->C > 0 && d->Cout > 0, "bad conv sizes");
  OTGAN_CHECK_ARG(d->stride == 1 || d->stride == 2, "stride must be 1 or 2");
  OTGAN_CHECK_ARG(d->upsample == 0 || d->upsample == 1, "upsample must be 0 or 1");
  OTGAN_CHECK_ARG(d->KH >= 1 && d->KW >= 1 && d->KH * d->KW <= kMaxTaps, "filter too large");
  OTGAN_CHECK_ARG(d->ldx >= d->C && d->ldy >= d->y_coff + d->Cout, "bad leading dimensions");
  OTGAN_CHECK_ARG(d->preact >= 0 && d->preact <= 4, "unknown pre-activation");
  g->logUp = d->upsample;
  g->Hin = d->H << d->upsample;
  g->Win = d->W << d->upsample;
  g->OH = (g->Hin + d->stride - 1) / d->stride;
  g->OW = (g->Win + d->stride - 1) / d->stride;
  const int ph = (g->OH - 1) * d->stride + d->KH - g->Hin, pw = (g->OW - 1) * d->stride + d->KW - g->Win;
  g->pad_t = (ph > 0 ? ph : 0) / 2;  // TF 'SAME': the extra pixel goes after
  g->pad_l = (pw > 0 ? pw : 0) / 2;
  g->Ceff = d->C * (doubled_act(d->preact) ? 2 : 1);
  OTGAN_CHECK_ARG(ilog2_exact(g->OH) >= 0 && ilog2_exact(g->OW) >= 0 && ilog2_exact(g->Hin) >= 0 &&
                      ilog2_exact(g->Win) >= 0,
                  "spatial sizes must be powers of two (got %dx%d)", g->Hin, g->Win);
  return OTGAN_OK;
}

struct WgPlan {
  bool vec;
  int tiles_m, tiles_n, nsplit, kt_per_split, ntap_z;
  long slab_elems;
};
WgPlan plan_wgrad(const otgan_conv_desc* d, const Geo& g) {
  WgPlan p;
  const int taps = d->KH * d->KW;
  const long M = (long)d->N * g.OH * g.OW;
  p.vec = (g.Ceff % 4 == 0) && (d->Cout % 4 == 0) && (d->ldy % 4 == 0) && (d->y_coff % 4 == 0) &&
          (d->ldx % 4 == 0) && (g.Ceff >= 32);
  const bool narrow = d->Cout <= 32;
  const int BM = narrow ? CfgNarrow::BM : CfgMain::BM, BN = narrow ? CfgNarrow::BN : CfgMain::BN;
  if (p.vec) {
    p.tiles_m = ceil_div(g.Ceff, BM);
    p.ntap_z = taps;
  } else {
    p.tiles_m = ceil_div(taps * g.Ceff, BM);
    p.ntap_z = 1;
  }
  p.tiles_n = ceil_div(d->Cout, BN);
  const int nkt = (int)ceil_div_l(M, 16);
  const int blocks = p.tiles_m * p.tiles_n * p.ntap_z;
  int want = ceil_div(1024, blocks);
  if (want < 1) want = 1;
  if (want > 64) want = 64;
  if (want > nkt) want = nkt;
  p.kt_per_split = ceil_div(nkt, want);
  p.nsplit = ceil_div(nkt, p.kt_per_split);
  p.slab_elems = (long)taps * g.Ceff * d->Cout;
  return p;
}

void fill_gather_x(const otgan_conv_desc* d, const Geo& g, const float* x, const int32_t* cmap,
                   int GH, int GW, int sa, GatherA* ga) {
  ga->x = x;
  ga->ldx = d->ldx;
  ga->H = d->H;
  ga->W = d->W;
  ga->logUp = g.logUp;
  ga->logGH = ilog2_exact(GH);
  ga->logGW = ilog2_exact(GW);
  ga->Mtot = d->N * GH * GW;
  ga->sa = sa;
  ga->Ck = g.Ceff;
  ga->cmap = cmap;
  ga->Creal = d->C;
  ga->doubled = doubled_act(d->preact) ? 1 : 0;
  ga->act = act_kind(d->preact);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <class Cfg, int EPI>
void launch_igemm(bool vec, dim3 grid, hipStream_t s, const GatherA& ga, const Taps& taps,
                  const WeightB& wb, const EpiArgs& e) {
  if (vec)
    hipLaunchKernelGGL((conv_igemm_kernel<Cfg, true, EPI>), grid, dim3(Cfg::THREADS), 0, s, ga, taps, wb, e);
  else
    hipLaunchKernelGGL((conv_igemm_kernel<Cfg, false, EPI>), grid, dim3(Cfg::THREADS), 0, s, ga, taps, wb, e);
}

}  // namespace

extern "C" {

size_t otgan_conv2d_workspace_bytes(const otgan_conv_desc* d, int which) {
  Geo g;
  if (make_geo(d, &g) != OTGAN_OK) return 0;
  if (which == 1) {
    // dgrad through a 2x upsample: gradient w.r.t. the virtual (upsampled) input
    return d->upsample ? align_up(sizeof(float) * (size_t)d->N * g.Hin * g.Win * d->C, 256) : 256;
  }
  if (which == 2) {
    const WgPlan p = plan_wgrad(d, g);
    return p.nsplit > 1 ? align_up(sizeof(float) * (size_t)p.slab_elems * p.nsplit, 256) : 256;
  }
  return 256;
}

int otgan_conv2d_fwd_f32(const otgan_conv_desc* d, const float* x, const int32_t* cmap,
                         const float* wT, const float* bias, float* y, void* stream) {
  Geo g;
  int rc = make_geo(d, &g);
  if (rc) return rc;
  OTGAN_CHECK_ARG(x && wT && y, "null pointer");
  hipStream_t s = (hipStream_t)stream;
  GatherA ga;
  fill_gather_x(d, g, x, cmap, g.OH, g.OW, d->stride, &ga);
  Taps taps;
  memset(&taps, 0, sizeof(taps));
  taps.n = d->KH * d->KW;
  for (int kh = 0; kh < d->KH; ++kh)
    for (int kw = 0; kw < d->KW; ++kw) {
      const int t = kh * d->KW + kw;
      taps.dh[t] = (short)(kh - g.pad_t);
      taps.dw[t] = (short)(kw - g.pad_l);
      taps.boff[t] = t * g.Ceff;
    }
  const int Ktot = taps.n * g.Ceff;
  WeightB wb;
  memset(&wb, 0, sizeof(wb));
  wb.w = wT;
  wb.ldbn = Ktot;
  wb.Nvalid = d->Cout;
  wb.Ck = g.Ceff;
  EpiArgs e;
  memset(&e, 0, sizeof(e));
  e.out = y; e.ldo = d->ldy; e.coff = d->y_coff;
  e.so = 1; e.oa = 0; e.ob = 0; e.OHf = g.OH; e.OWf = g.OW;
  e.bias = bias; e.ncols = d->Cout;
  const bool vec = (g.Ceff % 16 == 0) && (d->ldx % 4 == 0) && aligned16(x) && aligned16(wT);
  const double flops = 2.0 * ga.Mtot * (double)Ktot * d->Cout;
  ProfScope ps(OTGAN_PROF_CONV_FWD, flops, 0.0, s);
  if (d->Cout <= 32) {
    dim3 grid(ceil_div(ga.Mtot, CfgNarrow::BM), ceil_div(d->Cout, CfgNarrow::BN));
    launch_igemm<CfgNarrow, EPI_FWD>(vec, grid, s, ga, taps, wb, e);
  } else {
    dim3 grid(ceil_div(ga.Mtot, CfgMain::BM), ceil_div(d->Cout, CfgMain::BN));
    launch_igemm<CfgMain, EPI_FWD>(vec, grid, s, ga, taps, wb, e);
  }
  OTGAN_CHECK_LAUNCH("conv2d fwd");
  return OTGAN_OK;
}

int otgan_conv2d_dgrad_f32(const otgan_conv_desc* d, const float* dy, const float* w,
                           const float* x, const int32_t* inv, float* dx, int lddx,
                           int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
  Geo g;
  int rc = make_geo(d, &g);
  if (rc) return rc;
  OTGAN_CHECK_ARG(dy && w && dx && lddx >= d->C, "null pointer / bad lddx");
  const int kind = act_kind(d->preact);
  OTGAN_CHECK_ARG(kind == 0 || x, "dgrad through a pre-activation needs the layer input x");
  hipStream_t s = (hipStream_t)stream;
  const size_t need = otgan_conv2d_workspace_bytes(d, 1);
  float* target = dx;
  int tld = lddx, tacc = accumulate;
  if (d->upsample) {
    if (!workspace || workspace_bytes < need) {
      otgan_set_error("conv2d dgrad workspace too small: need %zu, got %zu", need, workspace_bytes);
      return OTGAN_ERR_WORKSPACE;
    }
    target = (float*)workspace;  // gradient on the virtual (upsampled) grid, dense [.,C]
    tld = d->C;
    tacc = 0;
  }
  // A operand: dy over the OUTPUT grid, one K block per (tap, co)
  GatherA ga;
  memset(&ga, 0, sizeof(ga));
  ga.x = dy + d->y_coff;
  ga.ldx = d->ldy;
  ga.H = g.OH; ga.W = g.OW; ga.logUp = 0;
  ga.sa = 1;
  ga.Ck = d->Cout;
  ga.cmap = nullptr; ga.Creal = d->Cout; ga.doubled = 0; ga.act = 0;
  WeightB wb;
  memset(&wb, 0, sizeof(wb));
  wb.w = w;
  wb.ldbn = d->Cout;
  wb.Ck = d->Cout;
  const bool paired = doubled_act(d->preact);
  wb.paired = paired ? 1 : 0;
  wb.Creal = d->C;
  wb.inv = inv;
  wb.Nvalid = g.Ceff;
  EpiArgs e;
  memset(&e, 0, sizeof(e));
  e.out = target; e.ldo = tld; e.coff = 0;
  e.OHf = g.Hin; e.OWf = g.Win;
  e.ncols = d->C;
  e.accumulate = tacc;
  e.xsrc = x; e.ldxs = d->ldx; e.xH = d->H; e.xW = d->W; e.logUpX = g.logUp;
  e.act = kind;
  const bool vec = (d->Cout % 16 == 0) && (d->ldy % 4 == 0) && (d->y_coff % 4 == 0) &&
                   aligned16(dy) && aligned16(w);
  const int st = d->stride;
  const int nclass = st * st;
  for (int cls = 0; cls < nclass; ++cls) {
    const int ph = cls / st, pw = cls % st;
    const int GH = g.Hin / st, GW = g.Win / st;
    ga.logGH = ilog2_exact(GH);
    ga.logGW = ilog2_exact(GW);
    OTGAN_CHECK_ARG(ga.logGH >= 0 && ga.logGW >= 0, "dgrad grid must be a power of two");
    ga.Mtot = d->N * GH * GW;
    Taps taps;
    memset(&taps, 0, sizeof(taps));
    int nt = 0;
    for (int kh = 0; kh < d->KH; ++kh) {
      if (((ph + g.pad_t - kh) % st + st) % st != 0) continue;
      for (int kw = 0; kw < d->KW; ++kw) {
        if (((pw + g.pad_l - kw) % st + st) % st != 0) continue;
        // oh = (ih + pad - kh) / stride, ih = a*st + ph
        const int nh = ph + g.pad_t - kh, nw = pw + g.pad_l - kw;
        taps.dh[nt] = (short)(nh >= 0 ? nh / st : -((-nh) / st));
        taps.dw[nt] = (short)(nw >= 0 ? nw / st : -((-nw) / st));
        taps.boff[nt] = (kh * d->KW + kw) * g.Ceff * d->Cout;
        ++nt;
      }
    }
    taps.n = nt;
    e.so = st; e.oa = ph; e.ob = pw;
    if (nt == 0) continue;  // (cannot happen for KH, KW >= stride)
    const double flops = 2.0 * ga.Mtot * (double)nt * d->Cout * g.Ceff;
    ProfScope ps(OTGAN_PROF_CONV_DGRAD, flops, 0.0, s);
    if (paired) {
      dim3 grid(ceil_div(ga.Mtot, CfgMain::BM), ceil_div(d->C, 64));
      launch_igemm<CfgMain, EPI_DG_PAIR>(vec, grid, s, ga, taps, wb, e);
    } else if (d->C <= 32) {
      dim3 grid(ceil_div(ga.Mtot, CfgNarrow::BM), ceil_div(d->C, CfgNarrow::BN));
      if (kind) launch_igemm<CfgNarrow, EPI_DG_ACT>(vec, grid, s, ga, taps, wb, e);
      else launch_igemm<CfgNarrow, EPI_DG_PLAIN>(vec, grid, s, ga, taps, wb, e);
    } else {
      dim3 grid(ceil_div(ga.Mtot, CfgMain::BM), ceil_div(d->C, CfgMain::BN));
      if (kind) launch_igemm<CfgMain, EPI_DG_ACT>(vec, grid, s, ga, taps, wb, e);
      else launch_igemm<CfgMain, EPI_DG_PLAIN>(vec, grid, s, ga, taps, wb, e);
    }
    OTGAN_CHECK_LAUNCH("conv2d dgrad");
  }
  if (d->upsample) {
    const long total = (long)d->N * d->H * d->W * d->C;
    long blocks = ceil_div_l(total, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pool2_sum_kernel, dim3((int)blocks), dim3(256), 0, s, target, d->N, d->H,
                       d->W, d->C, dx, lddx, accumulate);
    OTGAN_CHECK_LAUNCH("pool2_sum");
  }
  return OTGAN_OK;
}

int otgan_conv2d_wgrad_f32(const otgan_conv_desc* d, const float* x, const int32_t* cmap,
                           const float* dy, float* dw, void* workspace, size_t workspace_bytes,
                           void* stream) {
  Geo g;
  int rc = make_geo(d, &g);
  if (rc) return rc;
  OTGAN_CHECK_ARG(x && dy && dw, "null pointer");
  hipStream_t s = (hipStream_t)stream;
  WgPlan p = plan_wgrad(d, g);
  p.vec = p.vec && aligned16(x) && aligned16(dy);
  const size_t need = otgan_conv2d_workspace_bytes(d, 2);
  if (p.nsplit > 1 && (!workspace || workspace_bytes < need)) {
    otgan_set_error("conv2d wgrad workspace too small: need %zu, got %zu", need, workspace_bytes);
    return OTGAN_ERR_WORKSPACE;
  }
  GatherA ga;
  fill_gather_x(d, g, x, cmap, g.OH, g.OW, d->stride, &ga);
  Taps taps;
  memset(&taps, 0, sizeof(taps));
  taps.n = d->KH * d->KW;
  for (int kh = 0; kh < d->KH; ++kh)
    for (int kw = 0; kw < d->KW; ++kw) {
      const int t = kh * d->KW + kw;
      taps.dh[t] = (short)(kh - g.pad_t);
      taps.dw[t] = (short)(kw - g.pad_l);
    }
  WgradArgs a;
  memset(&a, 0, sizeof(a));
  a.dy = dy + d->y_coff;
  a.ldy = d->ldy;
  a.Cout = d->Cout;
  a.slab = p.nsplit > 1 ? (float*)workspace : dw;
  a.kt_per_split = p.kt_per_split;
  a.tiles_n = p.tiles_n;
  a.slab_stride = p.slab_elems;
  const bool narrow = d->Cout <= 32;
  dim3 grid(p.tiles_m * p.tiles_n, p.nsplit, p.ntap_z);
  {
    ProfScope ps(OTGAN_PROF_CONV_WGRAD, 2.0 * ga.Mtot * (double)p.slab_elems, 0.0, s);
    if (p.vec) {
      if (narrow) hipLaunchKernelGGL((conv_wgrad_kernel<CfgNarrow>), grid, dim3(CfgNarrow::THREADS), 0, s, ga, taps, a);
      else hipLaunchKernelGGL((conv_wgrad_kernel<CfgMain>), grid, dim3(CfgMain::THREADS), 0, s, ga, taps, a);
    } else {
      if (narrow) hipLaunchKernelGGL((conv_wgrad_scalar_kernel<CfgNarrow>), grid, dim3(CfgNarrow::THREADS), 0, s, ga, taps, a);
      else hipLaunchKernelGGL((conv_wgrad_scalar_kernel<CfgMain>), grid, dim3(CfgMain::THREADS), 0, s, ga, taps, a);
    }
  }
  OTGAN_CHECK_LAUNCH("conv2d wgrad");
  if (p.nsplit > 1) {
    long blocks = ceil_div_l(p.slab_elems, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((int)blocks), dim3(256), 0, s, (const float*)workspace,
                       p.nsplit, p.slab_elems, dw);
    OTGAN_CHECK_LAUNCH("slab_reduce");
  }
  return OTGAN_OK;
}

}  // extern "C"
