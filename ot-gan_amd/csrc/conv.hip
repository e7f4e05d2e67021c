// conv.hip -- NHWC implicit-GEMM convolution on the fp32 MFMA engine (gfx950):
// forward, data gradient and weight gradient, with the reference's pre-activation
// (CReLU / CELU / ELU / ReLU over a list of inputs), 2x nearest-neighbour upsampling and TF
// 'SAME' padding fused into the operand gathers.
// Replaces reference utils/nn.py:190-206 (pre-activation), :234-241 (conv), :327-338.
//
// GEMM views (K index = (tap, channel) or pixels):
//   fwd   : Y[pixel, co]        = sum_{tap, d}  A(x)[pixel+tap, d] * W[tap, d, co]
//   dgrad : dXe[in-pixel, d]    = sum_{tap, co} dY[in-pixel-tap, co] * W[tap, d, co]
//           then dX[c] = act'(x_c) (dXe[d+(c)] , dXe[d-(c)]) in the epilogue
//   wgrad : dW[tap, d, co]      = sum_{pixel}  A(x)[pixel+tap, d] * dY[pixel, co]
//           (split over pixels, slabs reduced afterwards)
//
// "Classes": a launch may cover up to four tap tables selected by blockIdx.z.
//   * stride-2 dgrad: one class per input-pixel parity (only the taps that reach it);
//   * upsample folding: conv(k x k) o nearest-upsample(2x) == four parity-class convs with
//     ceil((k+1)/2)^2 pre-summed taps on the SMALL grid (k=5: 9 instead of 25 taps; k=3: 4
//     instead of 9) -- identical mathematics, 25/9 (9/4) fewer MACs in fwd, dgrad and wgrad.
//
// Operand staging: load() only ISSUES the global loads of the next K tile (raw values into
// registers); sign / activation are applied in store(), after the current tile's MFMAs, so
// the memory latency hides under BK/2 * MT*NT * 64 cycles of matrix work.
#include "gemm_tile.h"
#include "dense16.h"
#include <vector>
#include "winograd.h"

// winograd.hip exists twice in the library (winograd_api.inc): two scaled fp16 pieces per operand element (wino_p2,
// default) or three bf16 pieces (wino_p3: 24 significand bits, six MFMAs per product) -- OTGAN_WINO_PIECES=3, read per
// call so that one process can measure both (prepared filters belong to the mode they were made in)
static inline int wino_pieces() {
  const char* e = getenv("OTGAN_WINO_PIECES");
  return (e && e[0] == '3') ? 3 : 2;
}
#define WINO(fn) (wino_pieces() == 3 ? wino_p3::fn : wino_p2::fn)
#include "../../include/otgan.h"

namespace {

// Tile configurations (measured with tools/ablate/gemm_ablate.hip on MI355X, fp32 MFMA):
// 128x128x32 is the best general tile (111-127 TFLOP/s, 2 workgroups/CU at 66 KB LDS);
// 64x128x32 wins when the 128x128 grid would have < 2 workgroups per CU (113 vs 102);
// BK = 16 variants serve channel counts that are not a multiple of 32 and the scalar paths.
using CfgMain = GemmCfg<2, 2, 2, 2, 32>;    // 128 x 128 x 32
using CfgSmall = GemmCfg<2, 2, 1, 2, 32>;   // 64 x 128 x 32
using CfgMain16 = GemmCfg<2, 2, 2, 2, 16>;  // 128 x 128 x 16
using CfgSmall16 = GemmCfg<2, 2, 1, 2, 16>; // 64 x 128 x 16
using CfgW160k16 = GemmCfg<4, 1, 1, 5, 16>;  // 128 x 160 x 16 (split-precision main loop)
using CfgNarrow = GemmCfg<4, 1, 2, 1, 16>;  // 256 x 32 x 16 (Cout / Cin <= 32)
using CfgN16 = GemmCfg<4, 1, 4, 1, 16, 16>;  // 256 x 16 x 16 on v_mfma_f32_16x16x4_f32 (Cout <= 16: DenseNet)
// One column tile for outputs just above a multiple of 128 (the DenseNet transition layers: Cout = 144, 200, 208):
// two 128-wide tiles would compute 256 columns for 144 (44 % of the matrix work wasted).  Four waves stacked along
// M, each 32 rows x all columns.
using CfgW160 = GemmCfg<4, 1, 1, 5, 32>;     // 128 x 160 x 32  (74 KB LDS: two workgroups per CU)
using CfgW224 = GemmCfg<4, 1, 1, 7, 16>;     // 128 x 224 x 16  (46 KB LDS)

// Kernels use dynamic LDS (the 128x128x32 tile needs 66 KB > the 64 KB static limit).
template <auto Kern>
inline void ensure_lds(size_t bytes) {
  static const bool done = [bytes] {
    hipFuncSetAttribute(reinterpret_cast<const void*>(Kern),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    return true;
  }();
  (void)done;
}

constexpr size_t kIgemmCmapBytes = 8192;
constexpr int kMaxTaps = 36;
constexpr int kMaxClass = 4;

struct Taps {
  int n;
  int dhw[kMaxTaps];   // (dh << 16) | (dw & 0xffff)
  int boff[kMaxTaps];  // offset of this tap's weight block in the B operand (elements)
};
struct ClassTab {
  int ncls;
  int oa[kMaxClass], ob[kMaxClass];  // output-pixel offsets of the class
  int zbase[kMaxClass + 1];          // prefix sums of taps (wgrad: blockIdx.z -> class, tap)
  long woff[kMaxClass];              // offset of the class's weight block (elements)
  Taps taps[kMaxClass];
};

__device__ __forceinline__ int sx16(int v) { return (int)(short)(v & 0xffff); }

// Source of the gathered ("A") operand.
struct GatherA {
  const float* x;
  int ldx;
  int H, W;          // stored dims
  int logUp;         // virtual dims = H << logUp (legacy un-folded upsample path)
  int logGH, logGW;  // row grid [*, GH, GW] (powers of two)
  int Mtot;          // rows = N * GH * GW
  int sa;            // virtual coord = grid coord * sa + tap offset
  int Ck;            // effective channels per tap
  const int* cmap;
  int Creal;
  int doubled;       // default map: d < Creal -> +x[d], else -x[d - Creal]
};

__device__ __forceinline__ void decode_map(const GatherA& g, int d, int cmv, int& c, float& sgn) {
  if (g.cmap) {
    c = cmv & 0x7fffffff;
    sgn = cmv < 0 ? -1.f : 1.f;
  } else if (g.doubled && d >= g.Creal) {
    c = d - g.Creal;
    sgn = -1.f;
  } else {
    c = d;
    sgn = 1.f;
  }
}

template <int ACT>
__device__ __forceinline__ float act_apply(float v) {
  if (ACT == 1) return fmaxf(v, 0.f);
  if (ACT == 2) return v > 0.f ? v : expm1f(v);
  return v;
}
__device__ __forceinline__ float act_deriv(int act, float v) {
  if (act == 1) return v > 0.f ? 1.f : 0.f;
  if (act == 2) return v > 0.f ? 1.f : expf(v);
  return 1.f;
}

// ---------------------------------------------------------------------------------------
// A loader (rows = pixels, K = (tap, channel), K contiguous in memory)
// ---------------------------------------------------------------------------------------
template <class Cfg, int BR, bool VEC, int ACT>
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
  unsigned neg[PASSES];  // scalar path: per-element sign bits
  float sgn;             // vector path: sign of this thread's quad in the pending tile
  int cm_next;           // prefetched channel-map entry of the next tile
  int nt, nd;            // (tap, channel offset) of the next tile to load
  int vH, vW, Ktot;

  __device__ __forceinline__ ConvALoader(const GatherA& g_, const Taps& t_) : g(g_), taps(t_) {}

  __device__ __forceinline__ void init(int m0) {
    const int r0 = threadIdx.x / CPR;
    vH = g.H << g.logUp;
    vW = g.W << g.logUp;
    Ktot = taps.n * g.Ck;
    nt = 0;
    nd = 0;
    sgn = 1.f;
    cm_next = (VEC && g.cmap && 4 * (threadIdx.x % CPR) < g.Ck) ? g.cmap[4 * (threadIdx.x % CPR)] : 0;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int r = r0 + p * RPP;
      const int m = m0 + r;
      neg[p] = 0;
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

  // vector path: start at channel offset d0 (a multiple of BK) instead of 0 -- the K split of the forward pass
  __device__ __forceinline__ void seek(int d0) {
    nd = d0;
    nt = 0;
    cm_next = (VEC && g.cmap && d0 + 4 * (threadIdx.x % CPR) < g.Ck) ? g.cmap[d0 + 4 * (threadIdx.x % CPR)] : 0;
  }
  // issue the loads of K tile kt (tiles are requested in order 0, 1, 2, ...)
  __device__ __forceinline__ void load(int kt) {
    const int c4 = threadIdx.x % CPR;
    if (VEC) {
      // every K tile lies inside one tap: a tap contributes ceil(Ck / BK) tiles, the last one zero-filled past
      // Ck (Ck % 4 == 0; the DenseNet transitions have Ck = 200, 228)
      const int t = __builtin_amdgcn_readfirstlane(nt);
      const int d = __builtin_amdgcn_readfirstlane(nd) + 4 * c4;
      int sc;
      decode_map(g, d, cm_next, sc, sgn);
      const int dhw = taps.dhw[t];
      const int dh = dhw >> 16, dw = sx16(dhw);
      const bool kin = d < g.Ck;
#pragma unroll
      for (int p = 0; p < PASSES; ++p) {
        const int ih = ia[p] + dh, iw = ib[p] + dw;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (kin && (unsigned)ih < (unsigned)vH && (unsigned)iw < (unsigned)vW) {
          const long pix = (long)pixbase[p] + (long)(ih >> g.logUp) * g.W + (iw >> g.logUp);
          v = *reinterpret_cast<const float4*>(g.x + pix * g.ldx + sc);
        }
        reg[p] = v;
      }
      // K order of the vector path: taps run FASTEST, channel slices slowest.  A block then reads
      // one BK-wide channel slice of its (haloed) pixel patch for all taps back to back, so the
      // patch slice (pixels x 128 B) stays L1/L2 resident instead of being re-fetched per tap
      // (tap-major order re-streams pixels x Ck x 4 B per tap and misses L2).
      nt += 1;
      if (nt >= taps.n) {
        nt = 0;
        nd += BK;
      }
      if (g.cmap && nd + 4 * c4 < g.Ck) cm_next = g.cmap[nd + 4 * c4];  // consumed one tile later
    } else {
      const int k = kt * BK + 4 * c4;
#pragma unroll
      for (int p = 0; p < PASSES; ++p) {
        float e[4] = {0.f, 0.f, 0.f, 0.f};
        unsigned nb = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int kk = k + q;
          if (kk < Ktot) {
            const int t = kk / g.Ck;
            const int d = kk - t * g.Ck;
            const int dhw = taps.dhw[t];
            const int ih = ia[p] + (dhw >> 16), iw = ib[p] + sx16(dhw);
            if ((unsigned)ih < (unsigned)vH && (unsigned)iw < (unsigned)vW) {
              int c;
              float s;
              decode_map(g, d, g.cmap ? g.cmap[d] : 0, c, s);
              const long pix = (long)pixbase[p] + (long)(ih >> g.logUp) * g.W + (iw >> g.logUp);
              e[q] = g.x[pix * g.ldx + c];
              if (s < 0.f) nb |= 1u << q;
            }
          }
        }
        reg[p] = make_float4(e[0], e[1], e[2], e[3]);
        neg[p] = nb;
      }
    }
  }

  __device__ __forceinline__ void store(float* t) const {
    const int c4 = threadIdx.x % CPR, r0 = threadIdx.x / CPR;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int r = r0 + p * RPP;
      if (r < BR) {
        float4 v = reg[p];
        if (VEC) {
          v.x *= sgn; v.y *= sgn; v.z *= sgn; v.w *= sgn;
        } else {
          if (neg[p] & 1u) v.x = -v.x;
          if (neg[p] & 2u) v.y = -v.y;
          if (neg[p] & 4u) v.z = -v.z;
          if (neg[p] & 8u) v.w = -v.w;
        }
        float* d = t + (4 * c4) * LD + r;
        d[0] = act_apply<ACT>(v.x);
        d[LD] = act_apply<ACT>(v.y);
        d[2 * LD] = act_apply<ACT>(v.z);
        d[3 * LD] = act_apply<ACT>(v.w);
      }
    }
  }
  // the same tile as three k-contiguous bf16 planes (gemm_mainloop_x3s; vector path, BK = 16)
  __device__ __forceinline__ void store3(unsigned char* t) const {
    const int c4 = threadIdx.x % CPR, r0 = threadIdx.x / CPR;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int r = r0 + p * RPP;
      if (r < BR) {
        float4 v = reg[p];
        v.x = act_apply<ACT>(v.x * sgn); v.y = act_apply<ACT>(v.y * sgn);
        v.z = act_apply<ACT>(v.z * sgn); v.w = act_apply<ACT>(v.w * sgn);
        x3s_store4(t, BR * kX3sRowBytes, r, 4 * c4, v);
      }
    }
  }
  // ... as two scaled fp16 planes (gemm_mainloop_x2h)
  float xs = 1.f;
  __device__ __forceinline__ void store2(unsigned char* t) const {
    const int c4 = threadIdx.x % CPR, r0 = threadIdx.x / CPR;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int r = r0 + p * RPP;
      if (r < BR) {
        float4 v = reg[p];
        v.x = act_apply<ACT>(v.x * sgn) * xs; v.y = act_apply<ACT>(v.y * sgn) * xs;
        v.z = act_apply<ACT>(v.z * sgn) * xs; v.w = act_apply<ACT>(v.w * sgn) * xs;
        x2h_store4(t, BR * kX3sRowBytes, r, 4 * c4, v);
      }
    }
  }
  // ---- stage interface of gemm_mainloop_x2h_d2 (vector path): the data of a k tile in flight lives in a caller-owned stage
  struct Stage {
    float4 reg[PASSES];
    unsigned ok;         // bit p: pass p loaded a real element (inside the image, inside the K range)
    float sgn;
  };
  const int* cmap_lds = nullptr;   // the channel map in LDS (set by the kernel; the global one would tie every trip to a load)
  __device__ __forceinline__ void load_s(Stage& st, bool live) {
    static_assert(VEC, "stage interface: vector gathers only");
    const int c4 = threadIdx.x % CPR;
    const int t = __builtin_amdgcn_readfirstlane(nt);
    const int d = __builtin_amdgcn_readfirstlane(nd) + 4 * c4;
    const bool kin = live && d < g.Ck;
    int sc;
    float sg;
    decode_map(g, d, (g.cmap && kin) ? cmap_lds[d] : 0, sc, sg);
    if (!kin) sc = 0;
    const int dhw = taps.dhw[t];
    const int dh = dhw >> 16, dw = sx16(dhw);
    unsigned ok = 0u;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int ih = ia[p] + dh, iw = ib[p] + dw;
      const bool in = kin && (unsigned)ih < (unsigned)vH && (unsigned)iw < (unsigned)vW;
      const int ihc = in ? ih : 0, iwc = in ? iw : 0;          // (branch-free: a masked element reads pixel 0 of its image)
      const long pix = (long)pixbase[p] + (long)(ihc >> g.logUp) * g.W + (iwc >> g.logUp);
      st.reg[p] = *reinterpret_cast<const float4*>(g.x + pix * g.ldx + sc);
      ok |= in ? (1u << p) : 0u;
    }
    st.ok = ok;
    st.sgn = sg;
    nt += 1;
    if (nt >= taps.n) {
      nt = 0;
      nd += BK;
    }
  }
  __device__ __forceinline__ void store2_s(unsigned char* t, const Stage& st) const {
    const int c4 = threadIdx.x % CPR, r0 = threadIdx.x / CPR;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int r = r0 + p * RPP;
      if (r < BR) {
        const bool ok = (st.ok >> p) & 1u;
        float4 v = st.reg[p];
        v.x = ok ? act_apply<ACT>(v.x * st.sgn) * xs : 0.f; v.y = ok ? act_apply<ACT>(v.y * st.sgn) * xs : 0.f;
        v.z = ok ? act_apply<ACT>(v.z * st.sgn) * xs : 0.f; v.w = ok ? act_apply<ACT>(v.w * st.sgn) * xs : 0.f;
        x2h_store4(t, BR * kX3sRowBytes, r, 4 * c4, v);
      }
    }
  }
  __device__ __forceinline__ static void pin_s(Stage& st) {
#pragma unroll
    for (int p = 0; p < PASSES; ++p)
      asm volatile("" : "+v"(*reinterpret_cast<gt_f32x4*>(&st.reg[p])));   // (one 128-bit operand: the load's register tuple stays whole)
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
  const float* wbase;
  long rowoff[PASSES];  // row(n) * ldbn, or -1 when the row is invalid
  float4 reg[PASSES];
  int nt, nd, Ktot;

  __device__ __forceinline__ ConvBLoader(const WeightB& b_, const Taps& t_, const float* wb)
      : b(b_), taps(t_), wbase(wb) {}

  __device__ __forceinline__ void init(int nblk) {
    const int r0 = threadIdx.x / CPR;
    Ktot = taps.n * b.Ck;
    nt = 0;
    nd = 0;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int r = r0 + p * RPP;
      long off = -1;
      if (r < BR) {
        if (b.paired) {
          // block tile = 64 real channels; wave column wn owns 32 of them; its two 32-wide
          // MFMA column tiles hold the (+c) and the (-c) effective channels.
          const int wn = r >> 6, ntile = (r >> 5) & 1, j = r & 31;
          const int c = nblk * 64 + wn * 32 + j;
          if (c < b.Creal) {
            const int d = b.inv ? b.inv[ntile * b.Creal + c] : (ntile * b.Creal + c);
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

  __device__ __forceinline__ void seek(int d0) {
    nd = d0;
    nt = 0;
  }
  __device__ __forceinline__ void load(int kt) {
    const int c4 = threadIdx.x % CPR;
    if (VEC) {
      const int t = __builtin_amdgcn_readfirstlane(nt);
      const int d = __builtin_amdgcn_readfirstlane(nd) + 4 * c4;
      const float* base = wbase + taps.boff[t] + d;
      const bool kin = d < b.Ck;
#pragma unroll
      for (int p = 0; p < PASSES; ++p) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (kin && rowoff[p] >= 0) v = *reinterpret_cast<const float4*>(base + rowoff[p]);
        reg[p] = v;
      }
      nt += 1;  // taps fastest, channel slices slowest (same order as the A loader)
      if (nt >= taps.n) {
        nt = 0;
        nd += BK;
      }
    } else {
      const int k = kt * BK + 4 * c4;
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
              e[q] = wbase[taps.boff[t] + rowoff[p] + d];
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
  __device__ __forceinline__ void store3(unsigned char* t) const {
    const int c4 = threadIdx.x % CPR, r0 = threadIdx.x / CPR;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int r = r0 + p * RPP;
      if (r < BR) x3s_store4(t, BR * kX3sRowBytes, r, 4 * c4, reg[p]);
    }
  }
  float xs = 1.f;
  __device__ __forceinline__ void store2(unsigned char* t) const {
    const int c4 = threadIdx.x % CPR, r0 = threadIdx.x / CPR;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int r = r0 + p * RPP;
      if (r < BR) {
        float4 v = reg[p];
        v.x *= xs; v.y *= xs; v.z *= xs; v.w *= xs;
        x2h_store4(t, BR * kX3sRowBytes, r, 4 * c4, v);
      }
    }
  }
  // ---- stage interface of gemm_mainloop_x2h_d2 (vector path)
  struct Stage {
    float4 reg[PASSES];
    unsigned ok;
  };
  __device__ __forceinline__ void load_s(Stage& st, bool live) {
    static_assert(VEC, "stage interface: vector gathers only");
    const int c4 = threadIdx.x % CPR;
    const int t = __builtin_amdgcn_readfirstlane(nt);
    int d = __builtin_amdgcn_readfirstlane(nd) + 4 * c4;
    const bool kin = live && d < b.Ck;
    if (!kin) d = 0;                           // (a masked quad reads the row's first one)
    const float* base = wbase + taps.boff[t] + d;
    unsigned ok = 0u;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const bool in = kin && rowoff[p] >= 0;
      st.reg[p] = *reinterpret_cast<const float4*>(base + (rowoff[p] >= 0 ? rowoff[p] : 0));
      ok |= in ? (1u << p) : 0u;
    }
    st.ok = ok;
    nt += 1;
    if (nt >= taps.n) {
      nt = 0;
      nd += BK;
    }
  }
  __device__ __forceinline__ void store2_s(unsigned char* t, const Stage& st) const {
    const int c4 = threadIdx.x % CPR, r0 = threadIdx.x / CPR;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int r = r0 + p * RPP;
      if (r < BR) {
        const float m = ((st.ok >> p) & 1u) ? xs : 0.f;
        float4 v = st.reg[p];
        v.x *= m; v.y *= m; v.z *= m; v.w *= m;
        x2h_store4(t, BR * kX3sRowBytes, r, 4 * c4, v);
      }
    }
  }
  __device__ __forceinline__ static void pin_s(Stage& st) {
#pragma unroll
    for (int p = 0; p < PASSES; ++p)
      asm volatile("" : "+v"(*reinterpret_cast<gt_f32x4*>(&st.reg[p])));   // (one 128-bit operand: the load's register tuple stays whole)
  }
};

// ---------------------------------------------------------------------------------------
// fwd / dgrad kernel
// ---------------------------------------------------------------------------------------
enum { EPI_FWD = 0, EPI_DG_PLAIN = 1, EPI_DG_ACT = 2, EPI_DG_PAIR = 3 };

struct EpiArgs {
  float* out;
  int ldo, coff;
  int so;                // out pixel = (n, a*so + oa, b*so + ob) on the [OHf, OWf] grid
  int OHf, OWf;
  const float* bias;     // fwd
  int ncols;             // valid output columns (fwd: Cout; dgrad: real channels)
  int accumulate;
  const float* xsrc;     // dgrad: layer input (activation derivative), stored resolution
  int ldxs, xH, xW, logUpX;
  int act;               // 1 relu-type, 2 elu-type
  // XS == 2 (two scaled fp16 pieces): amax records of the A tensor (x / dy) and of the weights; floor_one: the A operand
  // passes an ELU-type activation (|elu(x)| <= max(|x|, 1))
  const float* amax_a;
  const float* amax_b;
  int floor_one;
  float* amax_out;       // fwd: amax record of the values written (otgan_conv_desc::y_amax_out), or null
  // fwd, vector path, one class, so == 1 (round 4): K split over channel slices -- blockIdx.z = split, the workgroup takes
  // slices [split * per, ...) of every tap and writes its sums to partial[split][m][col]; igemm_splitk_finish_kernel adds
  // them up, with bias and amax record.  0 / 1: no split.
  int ksplit;
  float* partial;
};

// XS: 0 fp32 MFMA, 1 three bf16 pieces (six MFMAs per product), 2 two scaled fp16 pieces (three)
template <class Cfg, bool VEC, int EPI, int ACT, int XS = 0>
__global__ __launch_bounds__(Cfg::THREADS) void conv_igemm_kernel(GatherA g, ClassTab ct,
                                                                 WeightB wb, EpiArgs e) {
  using LA = ConvALoader<Cfg, Cfg::BM, VEC, ACT>;
  using LB = ConvBLoader<Cfg, Cfg::BN, VEC>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int m0 = blockIdx.x * Cfg::BM;
  const int nblk = blockIdx.y;
  const bool ksplit = VEC && EPI == EPI_FWD && e.ksplit > 1;
  const int cls = ksplit ? 0 : blockIdx.z;
  const Taps& taps = ct.taps[cls];
  LA la(g, taps);
  LB lb(wb, taps, wb.w + ct.woff[cls]);
  la.init(m0);
  lb.init(nblk);
  typename Cfg::acc_t acc[Cfg::MT][Cfg::NT];
  zero_acc<Cfg>(acc);
  int nkt = VEC ? taps.n * ((g.Ck + Cfg::BK - 1) / Cfg::BK) : (taps.n * g.Ck + Cfg::BK - 1) / Cfg::BK;
  if (ksplit) {
    const int nsl = (g.Ck + Cfg::BK - 1) / Cfg::BK, per = (nsl + e.ksplit - 1) / e.ksplit;
    const int s0 = blockIdx.z * per;
    int s1 = s0 + per;
    if (s1 > nsl) s1 = nsl;
    nkt = s1 > s0 ? taps.n * (s1 - s0) : 0;
    la.seek(s0 * Cfg::BK);
    lb.seek(s0 * Cfg::BK);
  }
  float descale = 1.f;
  if constexpr (XS == 2) {
    float aA = amax_record_value(e.amax_a);
    const float aB = amax_record_value(e.amax_b);
    if (e.floor_one && aA == aA) aA = fmaxf(aA, 1.f);
    int eA, eB;
    la.xs = x2h_scale(aA, &eA);
    lb.xs = x2h_scale(aB, &eB);
    descale = __builtin_ldexpf(1.f, eA + eB - 28);
#ifndef OTGAN_IGEMM_D1
    if constexpr (VEC) {
      // loads two k steps ahead (gemm_tile.h); the channel map of a list input moves to LDS behind the operand planes
      // (kIgemmCmapBytes at most: the launcher checks), so that no trip waits on a global load it has just issued
      int* s_cmap = reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(smem) + X2hLds<Cfg>::BYTES);
      if (g.cmap) {
        for (int i = threadIdx.x; i < g.Ck; i += Cfg::THREADS) s_cmap[i] = g.cmap[i];
        __syncthreads();
      }
      la.cmap_lds = s_cmap;
      gemm_mainloop_x2h_d2<Cfg>(la, lb, nkt, reinterpret_cast<unsigned char*>(smem), acc);
    } else
#endif
    gemm_mainloop_x2h<Cfg>(la, lb, nkt, reinterpret_cast<unsigned char*>(smem), acc);
  } else if constexpr (XS == 1) {
    gemm_mainloop_x3s<Cfg>(la, lb, nkt, reinterpret_cast<unsigned char*>(smem), acc);   // bf16 pipe, three pieces
  } else {
    gemm_mainloop<Cfg>(la, lb, nkt, smem, acc);
  }

  const int oa = ct.oa[cls], ob = ct.ob[cls];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
  const int lcol = Cfg::acc_col(lane);
  const int gmask = (1 << g.logGW) - 1, hmask = (1 << g.logGH) - 1;
  unsigned omax = 0u;
#pragma unroll
  for (int mt = 0; mt < Cfg::MT; ++mt) {
#pragma unroll
    for (int r = 0; r < Cfg::ACC; ++r) {
      const int row = (wm * Cfg::MT + mt) * Cfg::TS + Cfg::acc_row(lane, r);
      const int m = m0 + row;
      if (m >= g.Mtot) continue;
      const int b = m & gmask, a = (m >> g.logGW) & hmask, n = m >> (g.logGW + g.logGH);
      const int oh = a * e.so + oa, ow = b * e.so + ob;
      const long opix = ((long)n * e.OHf + oh) * e.OWf + ow;
      if (EPI == EPI_DG_PAIR) {
        static_assert(EPI != EPI_DG_PAIR || (Cfg::NT == 2 && Cfg::TS == 32), "paired epilogue needs 2 x 32 columns");
        const int c = nblk * 64 + wn * 32 + lcol;
        if (c < e.ncols) {
          const long xpix = ((long)n * e.xH + (oh >> e.logUpX)) * e.xW + (ow >> e.logUpX);
          const float xv = e.xsrc[xpix * e.ldxs + c];
          // d/dx [act(x) ; act(-x)] . [g+ ; g-] = act'(x) g+ - act'(-x) g-
          const float gp = acc[mt][0][r] * descale, gn = acc[mt][Cfg::NT - 1][r] * descale;
          const float v = act_deriv(e.act, xv) * gp - act_deriv(e.act, -xv) * gn;
          float* dst = e.out + opix * e.ldo + e.coff + c;
          const float o = e.accumulate ? (*dst + v) : v;
          *dst = o;
          { const unsigned b_ = amax_bits(o); omax = b_ > omax ? b_ : omax; }
        }
      } else {
#pragma unroll
        for (int nt = 0; nt < Cfg::NT; ++nt) {
          const int col = nblk * Cfg::BN + (wn * Cfg::NT + nt) * Cfg::TS + lcol;
          if (col >= e.ncols) continue;
          float v = acc[mt][nt][r] * descale;
          if (EPI == EPI_FWD && ksplit) {
            e.partial[((long)blockIdx.z * g.Mtot + m) * e.ncols + col] = v;
            continue;
          }
          if (EPI == EPI_FWD) {
            v += e.bias ? e.bias[col] : 0.f;
            { const unsigned b_ = amax_bits(v); omax = b_ > omax ? b_ : omax; }
          } else if (EPI == EPI_DG_ACT) {
            const long xpix = ((long)n * e.xH + (oh >> e.logUpX)) * e.xW + (ow >> e.logUpX);
            v *= act_deriv(e.act, e.xsrc[xpix * e.ldxs + col]);
          }
          float* dst = e.out + opix * e.ldo + e.coff + col;
          const float o = (EPI != EPI_FWD && e.accumulate) ? (*dst + v) : v;
          *dst = o;
          if (EPI != EPI_FWD) { const unsigned b_ = amax_bits(o); omax = b_ > omax ? b_ : omax; }
        }
      }
    }
  }
  // (every thread of the workgroup reaches this point.)  Round 4: the input-gradient epilogues leave the record too --
  // the largest magnitude of what ends up in memory, sums included -- so that the gradient a transition hands to the dense
  // block in front of it needs no reduction pass over the whole buffer (0.1 ms each at 32 x 32 x 480 channels).
  if (e.amax_out && !ksplit) amax_commit(e.amax_out, omax);
}

// out[m][coff + col] = bias[col] + sum over the K splits of partial[split][m][col]  (fixed order: deterministic), float4
// along the columns; the amax record of the values written
__global__ __launch_bounds__(256) void igemm_splitk_finish_kernel(const float* __restrict__ partial, int nsplit, long M, int ncols,
                                                                  const float* __restrict__ bias, float* __restrict__ out, int ldo,
                                                                  int coff, float* amax) {
  const int c4n = ncols >> 2;
  const long total = M * c4n;
  unsigned mb = 0u;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long m = i / c4n;
    const int c = (int)(i - m * c4n) * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(partial + m * ncols + c);
    for (int sp = 1; sp < nsplit; ++sp) v += *reinterpret_cast<const f32x4*>(partial + ((long)sp * M + m) * ncols + c);
    if (bias) v += *reinterpret_cast<const f32x4*>(bias + c);
    *reinterpret_cast<f32x4*>(out + m * ldo + coff + c) = v;
    mb = amax_bits4(v, mb);
  }
  if (amax) amax_commit(amax, mb);
}

// ---------------------------------------------------------------------------------------
// wgrad kernels
// ---------------------------------------------------------------------------------------
struct WgradArgs {
  const float* dy;   // already offset to the layer's channel range
  int ldy, Cout;
  int so, OHf, OWf;  // dy pixel of row-grid pixel (n,a,b): (n, a*so + oa, b*so + ob)
  float* slab;       // [nsplit][slab_stride]
  int kt_per_split;  // pixel tiles (of BK) per split
  int tiles_m, tiles_n;
  long slab_stride;
};

// rows = effective channel d within one (class, tap) selected by blockIdx.z; float4 along d
// (Ck % 4 == 0, contiguous cmap quads) and along co (Cout % 4 == 0, ldy % 4 == 0).
template <class Cfg, int ACT>
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
  __device__ __forceinline__ void init(int d0, int dhw, int mb) {
    const int c = threadIdx.x % CPK;
    const int d = d0 + 4 * c;
    rowok = d < g.Ck;
    sc = 0;
    sgn = 1.f;
    if (rowok) decode_map(g, d, g.cmap ? g.cmap[d] : 0, sc, sgn);
    dh = dhw >> 16;
    dw = sx16(dhw);
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
      if (kk < BK) {
        float4 v = reg[p];
        v.x = act_apply<ACT>(sgn * v.x);
        v.y = act_apply<ACT>(sgn * v.y);
        v.z = act_apply<ACT>(sgn * v.z);
        v.w = act_apply<ACT>(sgn * v.w);
        *reinterpret_cast<float4*>(t + kk * LD + 4 * c) = v;
      }
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
  int logGH, logGW, so, oa, ob, OHf, OWf;
  float4 reg[PASSES];
  __device__ __forceinline__ void init(const WgradArgs& a, const GatherA& g, int co0, int oa_,
                                       int ob_, int mb) {
    dy = a.dy; ldy = a.ldy; Cout = a.Cout; Mtot = g.Mtot;
    logGH = g.logGH; logGW = g.logGW;
    so = a.so; oa = oa_; ob = ob_; OHf = a.OHf; OWf = a.OWf;
    co = co0 + 4 * (threadIdx.x % CPK);
    m_begin = mb;
  }
  __device__ __forceinline__ long pixel(int m) const {
    const int b = m & ((1 << logGW) - 1);
    const int a = (m >> logGW) & ((1 << logGH) - 1);
    const int n = m >> (logGW + logGH);
    return ((long)n * OHf + a * so + oa) * OWf + b * so + ob;
  }
  __device__ __forceinline__ void load(int kt) {
    const int k0 = threadIdx.x / CPK;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int kk = k0 + p * KPP;
      const int m = m_begin + kt * BK + kk;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (kk < BK && m < Mtot && co < Cout)
        v = *reinterpret_cast<const float4*>(dy + pixel(m) * ldy + co);
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

template <class Cfg, int ACT>
__global__ __launch_bounds__(Cfg::THREADS) void conv_wgrad_kernel(GatherA g, ClassTab ct,
                                                                 WgradArgs a) {
  using LA = WgALoaderV<Cfg, ACT>;
  using LB = WgBLoaderV<Cfg>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // blockIdx.x = tile + ntiles * (class,tap): all taps and tiles of ONE pixel range are
  // dispatched together, so the concurrently resident blocks share the same x / dy pixels in L2
  const int ntiles = a.tiles_m * a.tiles_n;
  const int tile = blockIdx.x % ntiles, z = blockIdx.x / ntiles;
  const int tm = tile / a.tiles_n, tn = tile % a.tiles_n;
  const int split = blockIdx.y;
  int cls = 0;
#pragma unroll
  for (int c = 1; c < kMaxClass; ++c)
    if (c < ct.ncls && z >= ct.zbase[c]) cls = c;
  const int t = z - ct.zbase[cls];
  const int d0 = tm * Cfg::BM, co0 = tn * Cfg::BN;
  const int nkt_total = (g.Mtot + Cfg::BK - 1) / Cfg::BK;
  const int kt0 = split * a.kt_per_split;
  int nkt = nkt_total - kt0;
  if (nkt > a.kt_per_split) nkt = a.kt_per_split;
  LA la(g);
  LB lb;
  la.init(d0, ct.taps[cls].dhw[t], kt0 * Cfg::BK);
  lb.init(a, g, co0, ct.oa[cls], ct.ob[cls], kt0 * Cfg::BK);
  typename Cfg::acc_t acc[Cfg::MT][Cfg::NT];
  zero_acc<Cfg>(acc);
  gemm_mainloop<Cfg>(la, lb, nkt, smem, acc);
  float* out = a.slab + (long)split * a.slab_stride + ct.woff[cls] + (long)t * g.Ck * a.Cout;
  const int Ck = g.Ck, Cout = a.Cout;
  foreach_acc<Cfg>(acc, [&](int r, int c, int, int, int, float v) {
    const int d = d0 + r, co = co0 + c;
    if (d < Ck && co < Cout) out[(long)d * Cout + co] = v;
  });
}

// Generic scalar variant: rows = flattened (tap, d) index over ALL taps of a single class;
// any Ck / Cout / ld.
template <class Cfg, int ACT>
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
        decode_map(g, d, g.cmap ? g.cmap[d] : 0, sc[q], sgn[q]);
        dh[q] = taps.dhw[t] >> 16;
        dw[q] = sx16(taps.dhw[t]);
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
            e[q] = g.x[pix * g.ldx + sc[q]];
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
      if (kk < BK) {
        float4 v = reg[p];
        v.x = act_apply<ACT>(sgn[0] * v.x);
        v.y = act_apply<ACT>(sgn[1] * v.y);
        v.z = act_apply<ACT>(sgn[2] * v.z);
        v.w = act_apply<ACT>(sgn[3] * v.w);
        *reinterpret_cast<float4*>(t + kk * LD + 4 * c) = v;
      }
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

template <class Cfg, int ACT>
__global__ __launch_bounds__(Cfg::THREADS) void conv_wgrad_scalar_kernel(GatherA g, ClassTab ct,
                                                                        WgradArgs a) {
  using LA = WgALoaderS<Cfg, ACT>;
  using LB = WgBLoaderS<Cfg>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tm = blockIdx.x / a.tiles_n, tn = blockIdx.x % a.tiles_n;
  const int split = blockIdx.y;
  const int r0 = tm * Cfg::BM, co0 = tn * Cfg::BN;
  const int nkt_total = (g.Mtot + Cfg::BK - 1) / Cfg::BK;
  const int kt0 = split * a.kt_per_split;
  int nkt = nkt_total - kt0;
  if (nkt > a.kt_per_split) nkt = a.kt_per_split;
  const Taps& taps = ct.taps[0];
  LA la(g);
  LB lb;
  la.init(r0, taps, kt0 * Cfg::BK);
  lb.init(a, co0, g.Mtot, kt0 * Cfg::BK);
  typename Cfg::acc_t acc[Cfg::MT][Cfg::NT];
  zero_acc<Cfg>(acc);
  gemm_mainloop<Cfg>(la, lb, nkt, smem, acc);
  float* out = a.slab + (long)split * a.slab_stride;
  const int Rtot = taps.n * g.Ck, Cout = a.Cout;
  foreach_acc<Cfg>(acc, [&](int r, int c, int, int, int, float v) {
    const int rr = r0 + r, co = co0 + c;
    if (rr < Rtot && co < Cout) out[(long)rr * Cout + co] = v;
  });
}


// ---------------------------------------------------------------------------------------
// Few-channel layers (RGB in / RGB out): one side of the GEMM has <= 4 columns, where an MFMA
// tile would idle >= 28 of 32 columns.  These run on the vector ALU instead:
//   fewout : out[pix][j<4] = sum_{tap, d} act(src[pix (+) tap][d]) * Wf[tap][d][j]
//            (forward with Cout <= 4, and dgrad with Cin <= 4 where src = dy)
//            thread = pixel, weights are wave-uniform (scalar loads), 16 FMAs per 16-byte load
//   outer  : dW[tap][wide][j<4] = sum_pix wide[pix][.] * narrow[pix][j]
//            (wgrad with Cout <= 4 or Cin <= 4)   thread = wide channel, pixels streamed
// ---------------------------------------------------------------------------------------
struct FewOutArgs {
  const float* w;      // weight of (tap t, output j, channel d) at w[boff[t] + j*sJ + d]
  long sJ;
  float* out;
  int ldo, coff, J;    // J valid outputs (<= 4)
  int so, OHf, OWf;
  const float* bias;
  int accumulate;
};

// Lanes run along the channels (coalesced 16-byte loads: one wave-load = two full pixel rows of
// 128 channels); a wave owns 16 consecutive pixels (8 per 32-lane half), keeps the tap's
// weights in registers while it sweeps them, and finishes with a 32-lane shuffle reduction.
template <int ACT>
__global__ __launch_bounds__(256) void conv_fewout_kernel(GatherA g, Taps taps, FewOutArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ps = lane >> 5, q = lane & 31;
  const int m0 = (blockIdx.x * 4 + wave) * 16;
  if (m0 >= g.Mtot) return;
  const int vH = g.H << g.logUp, vW = g.W << g.logUp;
  int pa[8], pb[8];
  long pbase[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = m0 + 2 * i + ps;
    if (m < g.Mtot) {
      pb[i] = (m & ((1 << g.logGW) - 1)) * g.sa;
      pa[i] = ((m >> g.logGW) & ((1 << g.logGH) - 1)) * g.sa;
      pbase[i] = (long)(m >> (g.logGW + g.logGH)) * g.H * g.W;
    } else {
      pa[i] = pb[i] = -100000;
      pbase[i] = 0;
    }
  }
  float acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  for (int t = 0; t < taps.n; ++t) {
    const int dhw = taps.dhw[t];
    const int dh = dhw >> 16, dw = sx16(dhw);
    const float* wt = a.w + taps.boff[t];
    for (int d0 = 0; d0 < g.Ck; d0 += 128) {
      const int d = d0 + 4 * q;
      const bool dok = d < g.Ck;
      int sc = 0;
      float sgn = 1.f;
      if (dok) decode_map(g, d, g.cmap ? g.cmap[d] : 0, sc, sgn);
      float4 w[4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        w[j] = (dok && j < a.J) ? *reinterpret_cast<const float4*>(wt + j * a.sJ + d)
                                : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int ih = pa[i] + dh, iw = pb[i] + dw;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (dok && (unsigned)ih < (unsigned)vH && (unsigned)iw < (unsigned)vW) {
          const long pix = pbase[i] + (long)(ih >> g.logUp) * g.W + (iw >> g.logUp);
          v = *reinterpret_cast<const float4*>(g.x + pix * g.ldx + sc);
        }
        v.x = act_apply<ACT>(sgn * v.x);
        v.y = act_apply<ACT>(sgn * v.y);
        v.z = act_apply<ACT>(sgn * v.z);
        v.w = act_apply<ACT>(sgn * v.w);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] += v.x * w[j].x + v.y * w[j].y + v.z * w[j].z + v.w * w[j].w;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v = acc[i][j];
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);  // stays inside the 32-lane half
      acc[i][j] = v;
    }
  if (q == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int m = m0 + 2 * i + ps;
      if (m >= g.Mtot) continue;
      const int b = m & ((1 << g.logGW) - 1);
      const int aa = (m >> g.logGW) & ((1 << g.logGH) - 1);
      const int n = m >> (g.logGW + g.logGH);
      float* dst = a.out + (((long)n * a.OHf + aa * a.so) * a.OWf + b * a.so) * a.ldo + a.coff;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (j < a.J) {
          const float v = acc[i][j] + (a.bias ? a.bias[j] : 0.f);
          dst[j] = a.accumulate ? dst[j] + v : v;
        }
      }
    }
  }
}


// Tile variant of the few-output kernel for stride-1 layers (the RGB-out convolution of both generators,
// the input gradient of the RGB-in convolution): the kernel above re-reads the source once per tap
// (25x for a 5x5 filter, L2-bound at ~10 TB/s).  Here a block owns TR rows x W columns of one image
// (256 pixels, 128 threads x two vertically adjacent pixels), stages the tile plus halo for 16 channels at a
// time in LDS (39 KiB: four workgroups per CU; 32 channels = two per CU measured 0.18 vs 0.14 ms, 8 channels 0.18) -- sign / activation applied once, at fill time -- and every tap reads its shifted pixels as
// ds_read_b128 (pixel stride 5 slots of 16 bytes: conflict-free).  The chunk's weights sit in LDS
// too and are read as broadcasts, one read serving both pixels (scalar loads left the loop waiting on the
// scalar cache for most of its time: SQ_WAIT_ANY 79 %).
struct FewTileArgs {
  int TR, LH, LW, dh0, dw0;   // tile rows; LDS tile = LH x LW pixels starting at (r0 + dh0, dw0)
  unsigned lw_magic;          // ceil(2^20 / LW)
};

constexpr int kFewTileThreads = 128;
constexpr int kFewQ = 4;            // float4 (= 4 channels) per pixel and channel chunk; pixel stride kFewQ + 1 slots (odd)
constexpr int kFewS = kFewQ + 1;
template <int ACT, int NJ>
__global__ __launch_bounds__(kFewTileThreads) void conv_fewout_tile_kernel(GatherA g, Taps taps, FewOutArgs a, FewTileArgs ft) {
  extern __shared__ __attribute__((aligned(16))) float4 s_fx[];   // [LH * LW][kFewS] source tile, then [taps][NJ][kFewQ] weights
  const int tid = threadIdx.x;
  const int W = 1 << g.logGW, H = 1 << g.logGH;
  const int tiles_per_img = H / ft.TR;
  const int n = blockIdx.x / tiles_per_img, r0 = (blockIdx.x - n * tiles_per_img) * ft.TR;
  // a thread owns the two vertically adjacent pixels (2 pr2, pc), (2 pr2 + 1, pc): one weight read serves both
  const int pr2 = tid >> g.logGW, pc = tid & (W - 1);
  const long img = (long)n * H * W;
  const int npix = ft.LH * ft.LW;
  float4* s_w = s_fx + npix * kFewS;
  // two partial sums per output (even / odd channel pairs): the inner product is written as packed FMAs
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 acc2[2][NJ];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc2[p][j] = f32x2{0.f, 0.f};
  for (int d0 = 0; d0 < g.Ck; d0 += 4 * kFewQ) {
    __syncthreads();   // the previous chunk's tile is fully consumed
    // all loads of a batch in flight before the first LDS store (a load-store loop exposes one memory round trip
    // per element); p / LW through a multiply-shift (exact for p < 4096, LW <= 68)
    constexpr int kBatch = 9;
    for (int i0 = tid; i0 < npix * kFewQ; i0 += kFewTileThreads * kBatch) {
      float4 v[kBatch];
      float sg[kBatch];
#pragma unroll
      for (int b = 0; b < kBatch; ++b) {
        const int i = i0 + b * kFewTileThreads;
        const int q = i % kFewQ, p = i / kFewQ;
        const int lr = (int)(((unsigned)p * ft.lw_magic) >> 20), lc = p - lr * ft.LW;
        const int ih = r0 + lr + ft.dh0, iw = lc + ft.dw0;
        const int d = d0 + 4 * q;
        v[b] = make_float4(0.f, 0.f, 0.f, 0.f);
        sg[b] = 1.f;
        if (i < npix * kFewQ && d < g.Ck && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) {
          int sc;
          decode_map(g, d, g.cmap ? g.cmap[d] : 0, sc, sg[b]);
          v[b] = *reinterpret_cast<const float4*>(g.x + (img + (long)ih * W + iw) * g.ldx + sc);
        }
      }
#pragma unroll
      for (int b = 0; b < kBatch; ++b) {
        const int i = i0 + b * kFewTileThreads;
        if (i < npix * kFewQ) {
          float4 o;
          o.x = act_apply<ACT>(sg[b] * v[b].x);
          o.y = act_apply<ACT>(sg[b] * v[b].y);
          o.z = act_apply<ACT>(sg[b] * v[b].z);
          o.w = act_apply<ACT>(sg[b] * v[b].w);
          s_fx[(i / kFewQ) * kFewS + (i % kFewQ)] = o;
        }
      }
    }
    // the chunk's weights: s_w[(t * NJ + j) * kFewQ + q] = w[tap t][output j][d0 + 4q .. +3] (zero past the last channel)
    for (int i = tid; i < taps.n * NJ * kFewQ; i += kFewTileThreads) {
      const int q = i % kFewQ, tj = i / kFewQ;
      const int t = tj / NJ, j = tj - t * NJ;
      float4 wv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (d0 + 4 * q < g.Ck && j < a.J) wv = *reinterpret_cast<const float4*>(a.w + taps.boff[t] + j * a.sJ + d0 + 4 * q);
      s_w[i] = wv;
    }
    __syncthreads();
    // Software pipeline over half taps (kFewQ / 2 float4 each): the LDS reads of the next half are issued
    // before the FMAs of the current one, and a compiler barrier keeps them there -- left alone, the scheduler sinks
    // every read next to its use (read, wait, four FMAs), and with one wave per SIMD resident (72 KiB of LDS per
    // block) nothing else hides the LDS round trip.
    struct Half {
      float4 d0[kFewQ / 2], d1[kFewQ / 2], w[NJ][kFewQ / 2];
    };
    auto load_half = [&](Half& h, int t, int hq) {
      const int dhw = taps.dhw[t];
      const int dh = dhw >> 16, dw = sx16(dhw);
      const float4* src0 = s_fx + ((2 * pr2 + dh - ft.dh0) * ft.LW + (pc + dw - ft.dw0)) * kFewS + (kFewQ / 2) * hq;
      const float4* src1 = src0 + ft.LW * kFewS;
      const float4* wt = s_w + t * NJ * kFewQ + (kFewQ / 2) * hq;
#pragma unroll
      for (int q = 0; q < kFewQ / 2; ++q) {
        h.d0[q] = src0[q];
        h.d1[q] = src1[q];
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int q = 0; q < kFewQ / 2; ++q) h.w[j][q] = wt[j * kFewQ + q];   // the same address in every lane: LDS broadcast
      asm volatile("" ::: "memory");
    };
    auto fma_half = [&](const Half& h) {
#pragma unroll
      for (int q = 0; q < kFewQ / 2; ++q) {
        const f32x2 a01 = {h.d0[q].x, h.d0[q].y}, a23 = {h.d0[q].z, h.d0[q].w};
        const f32x2 b01 = {h.d1[q].x, h.d1[q].y}, b23 = {h.d1[q].z, h.d1[q].w};
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const f32x2 w01 = {h.w[j][q].x, h.w[j][q].y}, w23 = {h.w[j][q].z, h.w[j][q].w};
          acc2[0][j] = __builtin_elementwise_fma(a01, w01, acc2[0][j]);
          acc2[0][j] = __builtin_elementwise_fma(a23, w23, acc2[0][j]);
          acc2[1][j] = __builtin_elementwise_fma(b01, w01, acc2[1][j]);
          acc2[1][j] = __builtin_elementwise_fma(b23, w23, acc2[1][j]);
        }
      }
    };
    Half h0, h1;
    load_half(h0, 0, 0);
    for (int t = 0; t < taps.n; ++t) {
      load_half(h1, t, 1);
      fma_half(h0);
      load_half(h0, t + 1 < taps.n ? t + 1 : t, 0);   // past the last tap: a redundant reload, never used
      fma_half(h1);
    }
  }
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    float* dst = a.out + (img + (long)(r0 + 2 * pr2 + p) * W + pc) * a.ldo + a.coff;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (j < a.J) {
        const float v = acc2[p][j][0] + acc2[p][j][1] + (a.bias ? a.bias[j] : 0.f);
        dst[j] = a.accumulate ? dst[j] + v : v;
      }
    }
  }
}

// launches the tile kernel when the layer qualifies (stride 1 on a power-of-two grid of 16..64 columns, whole
// tiles, 3 or 4 outputs); returns false otherwise
template <int ACT>
static bool launch_fewout_tile(const GatherA& ga, const Taps& t, const FewOutArgs& fa, hipStream_t s) {
  if (ga.sa != 1 || ga.logUp != 0 || (fa.J != 3 && fa.J != 4) || fa.so != 1) return false;
  const int W = 1 << ga.logGW, H = 1 << ga.logGH;
  if (W != ga.W || H != ga.H || W < 16 || W > 64) return false;
  FewTileArgs ft;
  ft.TR = 256 / W;
  if (ft.TR > H || H % ft.TR) return false;
  int dh0 = 1 << 20, dh1 = -(1 << 20), dw0 = 1 << 20, dw1 = -(1 << 20);
  for (int i = 0; i < t.n; ++i) {
    const int dh = t.dhw[i] >> 16, dw = (int)(short)(t.dhw[i] & 0xffff);
    dh0 = dh < dh0 ? dh : dh0; dh1 = dh > dh1 ? dh : dh1;
    dw0 = dw < dw0 ? dw : dw0; dw1 = dw > dw1 ? dw : dw1;
  }
  ft.dh0 = dh0; ft.dw0 = dw0;
  ft.LH = ft.TR + dh1 - dh0; ft.LW = W + dw1 - dw0;
  ft.lw_magic = ((1u << 20) + ft.LW - 1) / ft.LW;
  if (ft.TR % 2) return false;
  const int nj = fa.J == 3 ? 3 : 4;
  const size_t lds = sizeof(float4) * (kFewS * (size_t)ft.LH * ft.LW + (size_t)t.n * nj * kFewQ);
  if (lds > 80 * 1024) return false;
  const dim3 grid(ga.Mtot / 256);
  if (fa.J == 3) {
    static bool once3 = (hipFuncSetAttribute((const void*)conv_fewout_tile_kernel<ACT, 3>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024), true);
    (void)once3;
    hipLaunchKernelGGL((conv_fewout_tile_kernel<ACT, 3>), grid, dim3(kFewTileThreads), lds, s, ga, t, fa, ft);
  } else {
    static bool once4 = (hipFuncSetAttribute((const void*)conv_fewout_tile_kernel<ACT, 4>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024), true);
    (void)once4;
    hipLaunchKernelGGL((conv_fewout_tile_kernel<ACT, 4>), grid, dim3(kFewTileThreads), lds, s, ga, t, fa, ft);
  }
  return true;
}

// RGB-out input gradient (the last convolution of both generators, 3 output channels: models/dcgan.py:48,
// models/densenet.py:86): dx[pix][c] = act'(x) . sum_{tap, j<3} dy[pix - tap][j] * W[tap][d(c)][j] for MANY channels c
// from a K = 9*3 or 25*3 deep sum -- a streaming kernel (read x for the activation mask, write dx), not a GEMM: the
// implicit-GEMM path ran its 16-deep K tiles almost empty and wrote through the paired epilogue at 1.1 TB/s.
// Lanes run along channel quads (every load / store of a wave is 4 pixels x 256 contiguous bytes); a workgroup owns
// (image, 16-column strip, 64 channels), keeps those channels' weights in LDS for the whole strip ([tap][half][j][64],
// one ds_read_b128 per (tap, half, j)) and walks down the strip 4 rows at a time with the dy tile (+ halo, padded
// to float4 per pixel) in LDS; a lane accumulates 4 neighbouring pixels so that every weight read feeds 12 FMAs per
// channel and a dy row window is read once per filter row.
// Few-output layers on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32: exact fp32 products, the same arithmetic as the
// VALU kernels, at twice the scalar FMA rate and without the per-FMA operand traffic).  Three outputs waste a 16-column
// tile, so the filter COLUMNS go into the tile as well: with n = kw * J + j (15 of 16 columns for a 5 x 5 RGB layer)
//     acc[(h, w')][n] = sum_{kh, d} act(src[(h + kh - pad, w')][d]) * W[(kh, kw)][j][d]      K = KH * channels
// is a GEMM whose A rows are whole image rows read where they lie -- the vertical taps are K steps, not a gather -- and
//     out[(h, w)][j] = sum_kw acc[(h, w + dw(kw))][kw * J + j]
// is a horizontal shift-and-add inside one image row (zero padding at both ends, no halo between workgroups), done
// through a per-wave LDS strip.  A wave owns RB = 4 consecutive output rows of one image and STREAMS the RB + KH - 1
// input rows they depend on past its RB row accumulators: an input row is loaded once (a lane: one float4 of its pixel's
// channels per 16-channel step, whose four elements feed four MFMAs against one ds_read_b128 of weights -- the K index
// of the instruction is a permutation of the channels, the same one on both operands) and multiplied into the up to KH
// output rows it belongs to; the row loop is unrolled, so which accumulator a (row, kh) pair adds into is static.
// (Row by row with the vertical taps as the outer K loop every input row was fetched KH times from L2 -- a 16 KiB row
// does not survive in L1 -- and the kernel ran at the VALU kernel's speed: 138 us against 147 us.)
// 5.4 GFLOP for the DCGAN RGB-out forward at 256 images: 34 us at the fp32 matrix peak, 27 us to read x once.
template <int ACT, int MT, int KH>
__global__ __launch_bounds__(256) void conv_fewout_mfma_kernel(GatherA g, Taps taps, FewOutArgs a, int KW, int flip) {
  extern __shared__ __attribute__((aligned(16))) float4 s_mw[];   // [KH][8][64] weights of a 128-channel chunk, then the strips
  constexpr int W = MT * 16, PADW = 2, PAD = (KH - 1) / 2, RB = 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const int H = 1 << g.logGH;
  const int blocks_per_img = H / RB;
  const int task = blockIdx.x * 4 + wave;
  const int n = task / blocks_per_img, ha = (task - n * blocks_per_img) * RB;
  const int NJ = a.J, ncol = KW * NJ;
  float* P = reinterpret_cast<float*>(s_mw + KH * 8 * 64) + wave * (W + 2 * PADW) * 16;
  P[lane < 32 ? lane : (W + PADW) * 16 + (lane - 32)] = 0.f;   // the strip's two zero pixels at either end
  f32x4 acc[RB][MT];
#pragma unroll
  for (int r = 0; r < RB; ++r)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[r][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int d0 = 0; d0 < g.Ck; d0 += 128) {
    __syncthreads();   // the previous chunk's weights are consumed
    for (int i = threadIdx.x; i < KH * 8 * 64; i += 256) {
      const int l = i & 63, cs = (i >> 6) & 7, kh = i >> 9;
      const int nn = l & 15, kw = nn / NJ, j = nn - kw * NJ;
      const int d = d0 + cs * 16 + 4 * (l >> 4);
      float4 wv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (nn < ncol && d < g.Ck)
        wv = *reinterpret_cast<const float4*>(a.w + taps.boff[(flip ? KH - 1 - kh : kh) * KW + kw] + j * a.sJ + d);
      s_mw[i] = wv;
    }
    __syncthreads();
    // the lane's eight channel quads of this chunk (source channel and sign: channel map / doubled pre-activation); the
    // last chunk may be a half (64 | channels)
    const int nhalf = g.Ck - d0 >= 128 ? 2 : 1;
    int sc[8];
    float sg[8];
#pragma unroll
    for (int cs = 0; cs < 8; ++cs) {
      const int d = d0 + cs * 16 + 4 * kq;
      sc[cs] = 0;
      sg[cs] = 1.f;
      if (d < g.Ck) decode_map(g, d, g.cmap ? g.cmap[d] : 0, sc[cs], sg[cs]);
    }
    // input row t of the wave is image row ha - PAD + t and adds, through filter row kh, into output row ha + t - kh
#pragma unroll
    for (int t = 0; t < RB + 2 * PAD; ++t) {
      const int ih = ha - PAD + t;
      if ((unsigned)ih >= (unsigned)H) continue;   // (a row of the zero padding adds nothing)
      const float* xrow = g.x + (((long)n * H + ih) * W + li) * g.ldx;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        if (half >= nhalf) break;
        float4 xv[4][MT];
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            xv[c4][mt] = *reinterpret_cast<const float4*>(xrow + (long)(mt * 16) * g.ldx + sc[half * 4 + c4]);
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
          const float s_ = sg[half * 4 + c4];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            xv[c4][mt].x = act_apply<ACT>(s_ * xv[c4][mt].x);
            xv[c4][mt].y = act_apply<ACT>(s_ * xv[c4][mt].y);
            xv[c4][mt].z = act_apply<ACT>(s_ * xv[c4][mt].z);
            xv[c4][mt].w = act_apply<ACT>(s_ * xv[c4][mt].w);
          }
        }
#pragma unroll
        for (int kh = 0; kh < KH; ++kh) {
          if (t - kh < 0 || t - kh >= RB) continue;   // static
          constexpr int kRB = RB;
          const int r = (t - kh + kRB) % kRB;
#pragma unroll
          for (int c4 = 0; c4 < 4; ++c4) {
            const float4 wv = s_mw[(kh * 8 + half * 4 + c4) * 64 + lane];
            // element-major: the MT chains of one element are independent of each other
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[r][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[c4][mt].x, wv.x, acc[r][mt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[r][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[c4][mt].y, wv.y, acc[r][mt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[r][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[c4][mt].z, wv.z, acc[r][mt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[r][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[c4][mt].w, wv.w, acc[r][mt], 0, 0, 0);
          }
        }
      }
    }
  }
  // shift-and-add along the row: the accumulator tile (lane: column n = li, rows 4 kq .. 4 kq + 3) goes to the wave's strip,
  // then lane o = (w, j) sums its KW shifted entries.  Lanes exchange data through the strip, so every lane must pass the
  // same points in the same order: a loop whose trip count depends on the lane lets the compiler run the lanes that leave
  // early AHEAD (it sank the next row's strip writes of lanes 48 - 63 above the other lanes' reads of this row); uniform
  // trip count + wave barriers.
#pragma unroll
  for (int r = 0; r < RB; ++r) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int q = 0; q < 4; ++q) P[(PADW + mt * 16 + 4 * kq + q) * 16 + li] = acc[r][mt][q];
    __builtin_amdgcn_wave_barrier();
    const long pix0 = ((long)n * H + ha + r) * W;
    for (int o0 = 0; o0 < W * NJ; o0 += 64) {
      const int o = o0 + lane;
      if (o < W * NJ) {
        const int w = o / NJ, j = o - w * NJ;
        float v = a.bias ? a.bias[j] : 0.f;
        for (int kw = 0; kw < KW; ++kw) v += P[(PADW + w + sx16(taps.dhw[kw])) * 16 + kw * NJ + j];
        float* dst = a.out + (pix0 + w) * a.ldo + a.coff + j;
        *dst = a.accumulate ? *dst + v : v;
      }
    }
  }
}
// launches the matrix-pipe kernel when the layer qualifies: a KH x KW rectangle of taps (row-major in `t`, KH = 3 or 5,
// dh = +-(kh - pad)) with KW * J <= 16 columns and |dw| <= 2, stride 1, rows of 16 / 32 / 64 pixels, 4 | H, 64 | channels
template <int ACT>
static bool launch_fewout_mfma(const GatherA& ga, const Taps& t, const FewOutArgs& fa, int KH, int KW, hipStream_t s) {
  if (ga.sa != 1 || ga.logUp != 0 || fa.so != 1 || fa.J < 1 || KW * fa.J > 16 || t.n != KH * KW || (KH != 3 && KH != 5)) return false;
  const int W = 1 << ga.logGW, H = 1 << ga.logGH;
  if (W != ga.W || H != ga.H || (W != 16 && W != 32 && W != 64) || H % 4 || ga.Ck % 64 || ga.ldx % 4 || fa.sJ % 4) return false;
  const int pad = (KH - 1) / 2;
  const int flip = (t.dhw[0] >> 16) > 0 ? 1 : 0;
  for (int i = 0; i < t.n; ++i) {
    const int dw = (int)(short)(t.dhw[i] & 0xffff), kh = i / KW;
    if (dw < -2 || dw > 2 || t.boff[i] % 4) return false;
    if ((t.dhw[i] >> 16) != (flip ? pad - kh : kh - pad) || dw != (int)(short)(t.dhw[i % KW] & 0xffff)) return false;
  }
  const long rows = (long)ga.Mtot / W;
  if (rows % 16) return false;   // four waves of four output rows per workgroup
  const dim3 grid((unsigned)(rows / 16));
  const size_t lds = sizeof(float4) * (size_t)KH * 8 * 64 + sizeof(float) * 4 * (W + 4) * 16;
#define FEW_MFMA(MT_, KH_)                                                                                              \
  do {                                                                                                                  \
    static bool once = (hipFuncSetAttribute((const void*)conv_fewout_mfma_kernel<ACT, MT_, KH_>,                        \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024), true);              \
    (void)once;                                                                                                         \
    hipLaunchKernelGGL((conv_fewout_mfma_kernel<ACT, MT_, KH_>), grid, dim3(256), lds, s, ga, t, fa, KW, flip);         \
  } while (0)
  if (KH == 5) {
    if (W == 16) FEW_MFMA(1, 5);
    else if (W == 32) FEW_MFMA(2, 5);
    else FEW_MFMA(4, 5);
  } else {
    if (W == 16) FEW_MFMA(1, 3);
    else if (W == 32) FEW_MFMA(2, 3);
    else FEW_MFMA(4, 3);
  }
#undef FEW_MFMA
  return true;
}

// acc += w * x as four scalar v_fma_f32, pinned in asm: what the compiler makes of the vector form is v_pk_fma_f32 with x
// broadcast through op_sel, and beside a wave of the 256 x 128 Winograd-domain GEMM on the same SIMD the low halves of
// lanes 48 - 63 of exactly these accumulate chains come back wrong (this kernel reproduced it in 46 of 480 launches:
// elements 0 and 2 of the lanes with slot & 3 == 3, every channel quad alike, also with s_waitcnt lgkmcnt(0) + s_nop 7
// between the LDS reads and the FMAs).  Root cause, round 4 (DESIGN section 3 "Four hazards" item 3,
// tools/debug/corun_probe.*): a packed fp32 instruction with OP_SEL set on SRC1 -- the low result half reads the HIGH
// register of the source pair -- is the one form that fails beside a wave with MFMAs and LDS-DMA loads in flight; the
// library is built without packed fp32 outside the Winograd transforms and tests/test_isa_cpu.py scans the binary for
// the form.  Packed fp32 buys no VALU throughput on this machine, so nothing is lost.
__device__ __forceinline__ void fma4_pinned(f32x4& acc, const f32x4 w, const float x) {
  float a0 = acc[0], a1 = acc[1], a2 = acc[2], a3 = acc[3];
  asm("v_fma_f32 %0, %4, %8, %0\n\tv_fma_f32 %1, %5, %8, %1\n\tv_fma_f32 %2, %6, %8, %2\n\tv_fma_f32 %3, %7, %8, %3"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
      : "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(x));
  acc = f32x4{a0, a1, a2, a3};
}
struct FewDgArgs {
  const float* dy;   // offset to the layer's channel range; [N, H, W, ldy], NJ channels used
  int ldy;
  const float* w;    // HWIO [K*K][Ceff][NJ]
  const int* inv;    // paired: effective index of +c (inv[c]) and -c (inv[C + c]); null: c, C + c
  const float* x;    // layer input (activation derivative), null without a pre-activation
  int ldx;
  float* dx;
  int lddx;
  int N, H, W, C, Ceff, pad_t, pad_l, act, accumulate;
  float* amax;     // otgan_conv_desc::dx_amax_out or null (round 4: the dense block behind an RGB-out layer reads it)
};
constexpr int kFdTW = 16, kFdTH = 4, kFdCH = 64;
template <int K, bool PAIRED, int NJ>
__global__ __launch_bounds__(256) void conv_fewout_dgrad_kernel(FewDgArgs a) {
  constexpr int HALVES = PAIRED ? 2 : 1;
  constexpr int LH = kFdTH + K - 1, LW = kFdTW + K - 1, WIN = 4 + K - 1;
  __shared__ __attribute__((aligned(16))) float wl[K * K][HALVES][NJ][kFdCH];
  __shared__ float4 dyt[LH][LW];
  const int n = blockIdx.x, w0 = blockIdx.y * kFdTW, c0 = blockIdx.z * kFdCH;
  unsigned mb = 0u;
  for (int i = threadIdx.x; i < K * K * HALVES * kFdCH; i += 256) {
    const int cc = i % kFdCH, half = (i / kFdCH) % HALVES, tap = i / (kFdCH * HALVES);
    const int c = c0 + cc;
    float v[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) v[j] = 0.f;
    if (c < a.C) {
      const int d = a.inv ? a.inv[half * a.C + c] : half * a.C + c;
      const float* src = a.w + ((long)tap * a.Ceff + d) * NJ;
#pragma unroll
      for (int j = 0; j < NJ; ++j) v[j] = src[j];
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) wl[tap][half][j][cc] = v[j];
  }
  const int q = threadIdx.x & 15, slot = threadIdx.x >> 4;
  const int r = slot >> 2, cg = (slot & 3) * 4;
  const int c = c0 + 4 * q;
  const float* dyn = a.dy + (long)n * a.H * a.W * a.ldy;
  for (int h0 = 0; h0 < a.H; h0 += kFdTH) {
    __syncthreads();   // the previous tile's reads (and, the first time, the weights' writes)
    for (int i = threadIdx.x; i < LH * LW; i += 256) {
      const int rr = i / LW, cc = i - rr * LW;
      const int hh = h0 + a.pad_t - (K - 1) + rr, ww = w0 + a.pad_l - (K - 1) + cc;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (hh >= 0 && hh < a.H && ww >= 0 && ww < a.W) {
        const float* sp = dyn + ((long)hh * a.W + ww) * a.ldy;
        v.x = sp[0];
        if (NJ > 1) v.y = sp[1];
        if (NJ > 2) v.z = sp[2];
        if (NJ > 3) v.w = sp[3];
      }
      dyt[rr][cc] = v;
    }
    __syncthreads();
    f32x4 acc[HALVES][4];
#pragma unroll
    for (int hf = 0; hf < HALVES; ++hf)
#pragma unroll
      for (int px = 0; px < 4; ++px) acc[hf][px] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1   // (fully unrolled the compiler hoists all K*K*NJ weight reads: 512 VGPRs and scratch)
    for (int kh = 0; kh < K; ++kh) {
      float4 win[WIN];
#pragma unroll
      for (int i = 0; i < WIN; ++i) win[i] = dyt[r + K - 1 - kh][cg + i];
#pragma unroll
      for (int kw = 0; kw < K; ++kw) {
#pragma unroll
        for (int hf = 0; hf < HALVES; ++hf) {
          f32x4 wv[NJ];
#pragma unroll
          for (int j = 0; j < NJ; ++j) wv[j] = *reinterpret_cast<const f32x4*>(&wl[kh * K + kw][hf][j][4 * q]);
#pragma unroll
          for (int px = 0; px < 4; ++px) {
            const float4 dv = win[px + K - 1 - kw];
            fma4_pinned(acc[hf][px], wv[0], dv.x);
            if constexpr (NJ > 1) fma4_pinned(acc[hf][px], wv[1], dv.y);
            if constexpr (NJ > 2) fma4_pinned(acc[hf][px], wv[2], dv.z);
            if constexpr (NJ > 3) fma4_pinned(acc[hf][px], wv[3], dv.w);
          }
        }
      }
    }
    if (c < a.C) {
#pragma unroll
      for (int px = 0; px < 4; ++px) {
        const long pix = ((long)n * a.H + h0 + r) * a.W + w0 + cg + px;
        f32x4 o = acc[0][px];
        if (a.act) {
          const f32x4 xv = *reinterpret_cast<const f32x4*>(a.x + pix * a.ldx + c);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (PAIRED) o[e] = act_deriv(a.act, xv[e]) * acc[0][px][e] - act_deriv(a.act, -xv[e]) * acc[HALVES - 1][px][e];
            else o[e] *= act_deriv(a.act, xv[e]);
          }
        }
        float* dst = a.dx + pix * a.lddx + c;
        if (a.accumulate) o += *reinterpret_cast<const f32x4*>(dst);
        mb = amax_bits4(o, mb);
        *reinterpret_cast<f32x4*>(dst) = o;
      }
    }
  }
  if (a.amax) amax_commit(a.amax, mb);
}

// RGB-in forward (the first convolution of both critics: 3 input channels, no pre-activation): lanes run along
// the OUTPUT channels.  A lane keeps the KS*KS*3 weights of its two channels (co, co + 64: one packed accumulator)
// in registers for the whole tile, the input pixel -- identical for all lanes -- comes out of an LDS tile (with
// halo) as one broadcast ds_read_b128 and feeds both channels' FMAs.  Every store is 64
// consecutive channels of one pixel (256 contiguous bytes).  The generic implicit GEMM spent its time in scalar gathers of
// the 3-channel pixels (38 TFLOP/s); this form is bound by the packed FMAs and the 134 MB it writes.
struct FewInArgs {
  const float* x; int ldx;
  const float* wT; int K;        // [Cout][K], K = KS*KS*3
  const float* bias;
  float* y; int ldy, coff;
  int N, H, W, logW, ph, pw, TR;
  float* amax;   // amax record of y (common.h: amax_commit), or null
};

template <int KS>
__global__ __launch_bounds__(256) void conv_rgbin_fwd_kernel(FewInArgs a) {
  extern __shared__ __attribute__((aligned(16))) float4 s_in[];   // [(TR + KS - 1)][(W + KS - 1)] pixels (r, g, b, 0)
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int co = blockIdx.y * 128 + lane;          // this lane's channels: co and co + 64 (one packed accumulator)
  const int tiles_per_img = a.H / a.TR;
  const int n = blockIdx.x / tiles_per_img, r0 = (blockIdx.x - n * tiles_per_img) * a.TR;
  const int LW = a.W + KS - 1, LH = a.TR + KS - 1;
  const long img = (long)n * a.H * a.W;
  for (int i = tid; i < LH * LW; i += 256) {
    const int lr = i / LW, lc = i - lr * LW;
    const int ih = r0 + lr - a.ph, iw = lc - a.pw;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W) {
      const float* px = a.x + (img + (long)ih * a.W + iw) * a.ldx;
      v = make_float4(px[0], px[1], px[2], 0.f);
    }
    s_in[i] = v;
  }
  f32x2 w[KS * KS * 3];
  const float* wlo = a.wT + (long)co * a.K;
  const float* whi = wlo + 64L * a.K;
#pragma unroll
  for (int i = 0; i < KS * KS * 3; ++i) w[i] = f32x2{wlo[i], whi[i]};
  const f32x2 b = a.bias ? f32x2{a.bias[co], a.bias[co + 64]} : f32x2{0.f, 0.f};
  __syncthreads();
  // this wave's pixels: a quarter of the tile's rows, two horizontally adjacent pixels per iteration (they share
  // the KS + 1 input positions of every filter row)
  const int rows = a.TR >> 2, rbase = wave * rows, pairs = a.W >> 1;
  unsigned mb = 0u;
  for (int pi = 0; pi < rows * pairs; ++pi) {
    const int r = rbase + pi / pairs, c0 = (pi % pairs) * 2;
    f32x2 acc0 = b, acc1 = b;
#pragma unroll
    for (int kh = 0; kh < KS; ++kh) {
      float4 xs[KS + 1];
      const float4* src = s_in + (r + kh) * LW + c0;
#pragma unroll
      for (int j = 0; j < KS + 1; ++j) xs[j] = src[j];   // the same address in every lane: LDS broadcast
#pragma unroll
      for (int kw = 0; kw < KS; ++kw) {
        const f32x2* wt = w + (kh * KS + kw) * 3;
        // Two scalar v_fma_f32 per product, pinned in asm.  The packed form (v_pk_fma_f32 with the pixel value broadcast
        // through op_sel) is what the compiler makes of `acc = fma(w2, {x, x}, acc)`, and it is (a) slower here -- 137 us
        // against 94 us per launch at 256 images: packed fp32 VALU buys nothing on this machine -- and (b) FRAGILE beside
        // matrix work: with a wave of the 256 x 128 Winograd-domain GEMM (wino_bgemm_x3n_kernel, round 3: the first GEMM
        // kernel that lets other workgroups share its compute unit) resident on the same SIMD, the low half of lanes
        // 48 - 63 of these packed FMAs came back wrong in 60 % of the launches (tools/debug/corun_repro.py: two streams;
        // tools/debug/dist_two_rank_trace.py: two processes on one GPU).  The same kernel with scalar FMAs: 0 of 1800, as
        // for every other kernel of the library as the neighbour (tests/test_corun_gpu.py).  Round 4 named the mechanism:
        // OP_SEL on SRC1 of a packed fp32 instruction (see fma4_pinned above and DESIGN section 3).
#define RGB_FMA(acc, w2, xv) do { float r0_, r1_; asm volatile("v_fma_f32 %0, %2, %4, %5\n\tv_fma_f32 %1, %3, %4, %6" : "=&v"(r0_), "=&v"(r1_) : "v"((w2)[0]), "v"((w2)[1]), "v"(xv), "v"((acc)[0]), "v"((acc)[1])); (acc)[0] = r0_; (acc)[1] = r1_; } while (0)
        RGB_FMA(acc0, wt[0], xs[kw].x); RGB_FMA(acc0, wt[1], xs[kw].y); RGB_FMA(acc0, wt[2], xs[kw].z);
        RGB_FMA(acc1, wt[0], xs[kw + 1].x); RGB_FMA(acc1, wt[1], xs[kw + 1].y); RGB_FMA(acc1, wt[2], xs[kw + 1].z);
#undef RGB_FMA
      }
    }
    float* dst = a.y + (img + (long)(r0 + r) * a.W + c0) * a.ldy + a.coff + co;
    dst[0] = acc0[0];
    dst[64] = acc0[1];
    dst[a.ldy] = acc1[0];
    dst[a.ldy + 64] = acc1[1];
    const unsigned b0 = amax_bits(acc0[0]), b1 = amax_bits(acc0[1]), b2 = amax_bits(acc1[0]), b3 = amax_bits(acc1[1]);
    const unsigned b01 = b0 > b1 ? b0 : b1, b23 = b2 > b3 ? b2 : b3, bq = b01 > b23 ? b01 : b23;
    mb = bq > mb ? bq : mb;
  }
  if (a.amax) amax_commit(a.amax, mb);
}

// The same layer on the fp32 matrix pipe: M = 16 pixels of an image row, N = 16 output channels, K = (tap, rgb) -- 75 of
// 76 for a 5 x 5 filter, 19 v_mfma_f32_16x16x4_f32 per tile.  A lane keeps its column's weights for all eight column tiles
// of a 64-channel group in REGISTERS for the whole workgroup (19 x 4 = 76), the A operand is one ds_read_b32 per K step
// from the haloed input tile (a lane's K index picks tap and colour: an offset table; the pad index reads the pixel's
// zero fourth float) and feeds four MFMAs.  A workgroup = 4 waves x 4 segments of 16 pixels per 256-pixel tile, two tiles.
// 5 GFLOP at 256 images: 32 us at the fp32 matrix peak, 27 us to write y.
template <int KS>
__global__ __launch_bounds__(256) void conv_rgbin_mfma_kernel(FewInArgs a, int tiles_per_block) {
  extern __shared__ __attribute__((aligned(16))) float4 s_rm[];   // [(TR + KS - 1)][(W + KS - 1)] pixels (r, g, b, 0)
  constexpr int KT = KS * KS * 3, STEPS = (KT + 3) / 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kq = lane >> 4;
  constexpr int NT = 4;            // 64 channels per workgroup: 76 weight registers, three waves per SIMD (eight tiles: 264 VGPRs, one)
  const int co0 = blockIdx.y * (16 * NT);
  const int LW = a.W + KS - 1, LH = a.TR + KS - 1;
  // B fragments: b[k = 4 s + kq][n = li] of column tile nt
  float bw[STEPS][NT];
  int aoff[STEPS];
#pragma unroll
  for (int st = 0; st < STEPS; ++st) {
    const int k = 4 * st + kq;
    const bool kin = k < KT;
    const int tap = kin ? k / 3 : 0, j = kin ? k - 3 * tap : 3;     // j = 3: the zero component
    const int kh = tap / KS, kw = tap - kh * KS;
    aoff[st] = (kh * LW + kw) * 4 + j;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bw[st][nt] = kin ? a.wT[(long)(co0 + nt * 16 + li) * a.K + k] : 0.f;
  }
  float bias[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bias[nt] = a.bias ? a.bias[co0 + nt * 16 + li] : 0.f;
  const int tiles_per_img = a.H / a.TR;
  const float* s_f = reinterpret_cast<const float*>(s_rm);
  const int segs_per_row = a.W >> 4, segs = a.TR * segs_per_row;    // 16 segments of 16 pixels per tile
  unsigned mb = 0u;
  for (int tb = 0; tb < tiles_per_block; ++tb) {
    const int tile = blockIdx.x * tiles_per_block + tb;
    const int n = tile / tiles_per_img, r0 = (tile - n * tiles_per_img) * a.TR;
    const long img = (long)n * a.H * a.W;
    __syncthreads();   // the previous tile is consumed
    for (int i = tid; i < LH * LW; i += 256) {
      const int lr = i / LW, lc = i - lr * LW;
      const int ih = r0 + lr - a.ph, iw = lc - a.pw;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W) {
        const float* px = a.x + (img + (long)ih * a.W + iw) * a.ldx;
        v = make_float4(px[0], px[1], px[2], 0.f);
      }
      s_rm[i] = v;
    }
    __syncthreads();
    for (int sg = wave; sg < segs; sg += 4) {
      const int r = sg / segs_per_row, c0 = (sg - r * segs_per_row) << 4;
      const float* abase = s_f + (r * LW + c0 + li) * 4;
      f32x4 acc[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
      float av[STEPS];
#pragma unroll
      for (int st = 0; st < STEPS; ++st) av[st] = abase[aoff[st]];
#pragma unroll
      for (int st = 0; st < STEPS; ++st)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[st], bw[st][nt], acc[nt], 0, 0, 0);
      // D[i = 4 kq + q][n = li]: pixel c0 + 4 kq + q, channel co0 + nt * 16 + li
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float* dst = a.y + (img + (long)(r0 + r) * a.W + c0 + 4 * kq + q) * a.ldy + a.coff + co0 + li;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const float o = acc[nt][q] + bias[nt];
          dst[nt * 16] = o;
          const unsigned ob = amax_bits(o);
          mb = ob > mb ? ob : mb;
        }
      }
    }
  }
  if (a.amax) amax_commit(a.amax, mb);
}

// set by a pass whose output kernel fills the requested otgan_conv_desc::y_amax_out / dx_amax_out record itself; the
// entry points run a separate reduction of the output otherwise
// amax records of the two operand tensors of a generic implicit GEMM, for its two-scaled-fp16-piece main loop (three
// MFMAs per product instead of six on three bf16 pieces): the caller's records when it has them (otgan_conv_desc::
// x_amax / dy_amax / w_amax), else reduced here into two scratch records at the tail of the workspace (one pass over the
// tensor each: 60 us for a 300 MB block buffer against 250 us saved in the GEMM).  Leaves e.amax_a / e.amax_b null --
// the three-piece loop runs -- when the tensors do not qualify or the workspace has no room (and always in the three-piece
// build, OTGAN_WINO_PIECES=3).
static void igemm_records(EpiArgs& e, const float* a_rec, const float* a, long a_rows, int a_C, long a_ld,
                          const float* b_rec, const float* b, long b_elems, int elu_type, void* workspace,
                          size_t workspace_bytes, size_t ws_used, hipStream_t s) {
  e.amax_a = e.amax_b = nullptr;
  e.floor_one = elu_type ? 1 : 0;
  if (wino_pieces() != 2) return;
  const size_t rec_bytes = sizeof(float) * OTGAN_AMAX_RECORD_FLOATS;
  float* scratch = nullptr;
  const size_t at = (ws_used + 255) / 256 * 256;
  if (workspace && at + 2 * rec_bytes <= workspace_bytes) scratch = reinterpret_cast<float*>(static_cast<char*>(workspace) + at);
  if (!a_rec) {
    if (!scratch || a_C % 4 || a_ld % 4 || (reinterpret_cast<uintptr_t>(a) & 15)) return;
    wino_p2::wino_absmax(a, a_rows, a_C, a_ld, scratch, s, false);
    a_rec = scratch;
  }
  if (!b_rec) {
    if (!scratch || b_elems % 4 || (reinterpret_cast<uintptr_t>(b) & 15)) return;
    wino_p2::wino_absmax(b, 1, (int)b_elems, 0, scratch + OTGAN_AMAX_RECORD_FLOATS, s, false);
    b_rec = scratch + OTGAN_AMAX_RECORD_FLOATS;
  }
  e.amax_a = a_rec;
  e.amax_b = b_rec;
}

static thread_local bool g_amax_written = false;

static bool launch_rgbin_fwd(const otgan_conv_desc* d, int pad_t, int pad_l, const float* x, const float* wT,
                             const float* bias, float* y, hipStream_t s) {
  if (d->C != 3 || d->stride != 1 || d->upsample != 0 || d->KH != d->KW || (d->KH != 5 && d->KH != 3) ||
      d->Cout % 128 != 0 || d->W < 16 || d->W > 64)
    return false;
  FewInArgs a;
  a.x = x; a.ldx = d->ldx; a.wT = wT; a.K = d->KH * d->KW * 3; a.bias = bias;
  a.y = y; a.ldy = d->ldy; a.coff = d->y_coff;
  a.N = d->N; a.H = d->H; a.W = d->W; a.logW = 0; a.ph = pad_t; a.pw = pad_l;
  a.amax = d->y_amax_out;
  a.TR = 256 / d->W;
  if (a.TR < 4 || a.TR > d->H || d->H % a.TR) return false;
  const size_t lds = sizeof(float4) * (size_t)(a.TR + d->KH - 1) * (d->W + d->KW - 1);
  const int tiles = d->N * (d->H / a.TR);
  if (d->W % 16 == 0) {   // fp32 matrix pipe; two tiles per workgroup while that leaves two workgroups per CU
    const int tpb = (tiles % 2 == 0 && (long)(tiles / 2) * (d->Cout / 64) >= 1024) ? 2 : 1;
    const dim3 gm(tiles / tpb, d->Cout / 64);
    if (d->KH == 5) hipLaunchKernelGGL(conv_rgbin_mfma_kernel<5>, gm, dim3(256), lds, s, a, tpb);
    else hipLaunchKernelGGL(conv_rgbin_mfma_kernel<3>, gm, dim3(256), lds, s, a, tpb);
    g_amax_written = a.amax != nullptr;
    return true;
  }
  const dim3 grid(tiles, d->Cout / 128);
  if (d->KH == 5) hipLaunchKernelGGL(conv_rgbin_fwd_kernel<5>, grid, dim3(256), lds, s, a);
  else hipLaunchKernelGGL(conv_rgbin_fwd_kernel<3>, grid, dim3(256), lds, s, a);
  g_amax_written = a.amax != nullptr;
  return true;
}

struct OuterArgs {
  // wide operand: value(pix, c) ; narrow operand: value(pix, j)
  const float* wide; int ldw; int wideC;      // channels of the wide side
  const float* narrow; int ldn; int J;        // J <= 4
  int wide_shift;      // 1: the tap shift applies to the wide operand (x), 0: to the narrow one
  int H, W, logGH, logGW, Mtot, sa;           // pixel grid of the un-shifted operand; shifted dims H,W
  const int* cmap; int Creal, doubled;        // channel map of the x operand
  int chunk;           // pixels per block
  float* slab;         // [nchunks][ntaps*wideC*J or ...]
  long slab_stride;
  long sT, sC, sJ;     // out index = t*sT + c*sC + j*sJ
};

// thread = 4 consecutive wide channels (one 16-byte load per pixel); the block's 8 pixel
// streams are combined through LDS at the end.
template <int ACT>
__global__ __launch_bounds__(256) void conv_outer_kernel(OuterArgs a, Taps taps) {
  const int cq = threadIdx.x & 31, ps = threadIdx.x >> 5;
  const int c = blockIdx.x * 128 + 4 * cq;
  const int t = blockIdx.z;
  const int p0 = blockIdx.y * a.chunk;
  int p1 = p0 + a.chunk;
  if (p1 > a.Mtot) p1 = a.Mtot;
  const int dh = taps.dhw[t] >> 16, dw = sx16(taps.dhw[t]);
  const bool cok = c < a.wideC;
  int sc = c;
  float sgn = 1.f;
  if (a.wide_shift && cok) {
    GatherA gm;
    gm.cmap = a.cmap; gm.Creal = a.Creal; gm.doubled = a.doubled;
    decode_map(gm, c, a.cmap ? a.cmap[c] : 0, sc, sgn);
  }
  float acc[4][4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[k][j] = 0.f;
  if (cok) {
    for (int m = p0 + ps; m < p1; m += 8) {
      const int b = m & ((1 << a.logGW) - 1);
      const int aa = (m >> a.logGW) & ((1 << a.logGH) - 1);
      const int n = m >> (a.logGW + a.logGH);
      const int ih = aa * a.sa + dh, iw = b * a.sa + dw;
      if ((unsigned)ih >= (unsigned)a.H || (unsigned)iw >= (unsigned)a.W) continue;
      const long spix = ((long)n * a.H + ih) * a.W + iw;   // shifted (x) pixel
      float4 wv;
      const float* np;
      if (a.wide_shift) {
        wv = *reinterpret_cast<const float4*>(a.wide + spix * a.ldw + sc);
        wv.x = act_apply<ACT>(sgn * wv.x);
        wv.y = act_apply<ACT>(sgn * wv.y);
        wv.z = act_apply<ACT>(sgn * wv.z);
        wv.w = act_apply<ACT>(sgn * wv.w);
        np = a.narrow + (long)m * a.ldn;
      } else {
        wv = *reinterpret_cast<const float4*>(a.wide + (long)m * a.ldw + c);
        np = a.narrow + spix * a.ldn;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (j < a.J) {
          const float nv = a.wide_shift ? np[j] : act_apply<ACT>(np[j]);
          acc[0][j] += wv.x * nv;
          acc[1][j] += wv.y * nv;
          acc[2][j] += wv.z * nv;
          acc[3][j] += wv.w * nv;
        }
      }
    }
  }
  __shared__ float red[8][128][4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int j = 0; j < 4; ++j) red[ps][4 * cq + k][j] = acc[k][j];
  __syncthreads();
  // 256 threads finish 128 channels x (up to) 4 outputs: thread -> (channel, j pair)
  const int ch = threadIdx.x & 127, jh = threadIdx.x >> 7;
  const int cc = blockIdx.x * 128 + ch;
  if (cc < a.wideC) {
    float* out = a.slab + (long)blockIdx.y * a.slab_stride + t * a.sT + (long)cc * a.sC;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int j = 2 * jh + jj;
      if (j < a.J) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) v += red[k][ch][j];
        out[j * a.sJ] = v;
      }
    }
  }
}


// Stride-1 variant of the few-channel weight gradient: ALL taps in one block.  The kernel above reads
// the wide operand once per tap (25x for a 5x5 filter: L2-bound); here a thread owns (4 wide channels,
// one filter row kh, one of two pixel streams), loads each wide pixel ONCE for its KW taps and takes the
// narrow operand (<= 4 channels per pixel, padded to a float4) from an LDS tile of the unit's rows plus
// halo.  out[tap][c][j] = sum_q wide(q, c) * narrow(q + sgn * off(tap), j): sgn = -1 when the wide
// operand is the layer input (few outputs), +1 when it is dy (few inputs).
struct Outer2Args {
  const float* wide; int ldw; int wideC;
  const float* narrow; int ldn; int J;
  int sgn, wide_is_x;
  int N, H, W, logW;
  int KH, KW, ph, pw;
  const int* cmap; int Creal, doubled;
  int TRo, units, units_per_block;
  float* slab; long slab_stride; long sT, sC, sJ;
};
constexpr int kOuterHalo = 2;

template <int ACT, int KW>
__global__ __launch_bounds__(320) void conv_outer2_kernel(Outer2Args a) {
  extern __shared__ __attribute__((aligned(16))) float4 s_nar[];   // [(TRo + 4)][(W + 4)] float4
  const int tid = threadIdx.x, nthreads = blockDim.x;
  const int cq = tid & 31, kh = (tid >> 5) % a.KH, ps = tid / (32 * a.KH);
  const int c = blockIdx.x * 128 + 4 * cq;
  const bool cok = c < a.wideC;
  int sc = c;
  float sgnw = 1.f;
  if (a.wide_is_x && cok) {
    GatherA gm;
    gm.cmap = a.cmap; gm.Creal = a.Creal; gm.doubled = a.doubled;
    decode_map(gm, c, a.cmap ? a.cmap[c] : 0, sc, sgnw);
  }
  float acc[KW][4][4];
#pragma unroll
  for (int w = 0; w < KW; ++w)
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[w][k][j] = 0.f;
  const int LW = a.W + 2 * kOuterHalo;
  const int units_per_img = a.H / a.TRo;
  const int u0 = blockIdx.y * a.units_per_block;
  int u1 = u0 + a.units_per_block;
  if (u1 > a.units) u1 = a.units;
  const int dh = a.sgn * (kh - a.ph);
  for (int u = u0; u < u1; ++u) {
    const int n = u / units_per_img, r0 = (u - n * units_per_img) * a.TRo;
    const long img = (long)n * a.H * a.W;
    __syncthreads();   // previous unit's tile fully consumed
    for (int i = tid; i < (a.TRo + 2 * kOuterHalo) * LW; i += nthreads) {
      const int lr = i / LW, lc = i - lr * LW;
      const int r = r0 - kOuterHalo + lr, q = lc - kOuterHalo;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((unsigned)r < (unsigned)a.H && (unsigned)q < (unsigned)a.W) {
        const float* np = a.narrow + (img + (long)r * a.W + q) * a.ldn;
        v.x = np[0];
        if (a.J > 1) v.y = np[1];
        if (a.J > 2) v.z = np[2];
        if (a.J > 3) v.w = np[3];
        if (!a.wide_is_x) {   // the narrow operand is the layer input: pre-activation applies to it
          v.x = act_apply<ACT>(v.x); v.y = act_apply<ACT>(v.y); v.z = act_apply<ACT>(v.z); v.w = act_apply<ACT>(v.w);
        }
      }
      s_nar[i] = v;
    }
    __syncthreads();
    if (cok) {
      for (int q = ps; q < a.TRo * a.W; q += 2) {
        const int row = q >> a.logW, col = q & (a.W - 1);
        float4 wv = *reinterpret_cast<const float4*>(a.wide + (img + (long)(r0 + row) * a.W + col) * a.ldw + sc);
        if (a.wide_is_x) {
          wv.x = act_apply<ACT>(sgnw * wv.x); wv.y = act_apply<ACT>(sgnw * wv.y);
          wv.z = act_apply<ACT>(sgnw * wv.z); wv.w = act_apply<ACT>(sgnw * wv.w);
        }
        const float4* nrow = s_nar + (row + kOuterHalo + dh) * LW + col + kOuterHalo;
#pragma unroll
        for (int w = 0; w < KW; ++w) {
          const float4 nv = nrow[a.sgn * (w - a.pw)];
          acc[w][0][0] += wv.x * nv.x; acc[w][0][1] += wv.x * nv.y; acc[w][0][2] += wv.x * nv.z; acc[w][0][3] += wv.x * nv.w;
          acc[w][1][0] += wv.y * nv.x; acc[w][1][1] += wv.y * nv.y; acc[w][1][2] += wv.y * nv.z; acc[w][1][3] += wv.y * nv.w;
          acc[w][2][0] += wv.z * nv.x; acc[w][2][1] += wv.z * nv.y; acc[w][2][2] += wv.z * nv.z; acc[w][2][3] += wv.z * nv.w;
          acc[w][3][0] += wv.w * nv.x; acc[w][3][1] += wv.w * nv.y; acc[w][3][2] += wv.w * nv.z; acc[w][3][3] += wv.w * nv.w;
        }
      }
    }
  }
  // combine the two pixel streams through LDS, then one thread per (kh, channel quad) writes its taps
  __syncthreads();
  float* red = reinterpret_cast<float*>(s_nar);      // [KH][32][KW*16]
  const int slot = (kh * 32 + cq) * (KW * 16);
  if (ps == 1) {
#pragma unroll
    for (int w = 0; w < KW; ++w)
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) red[slot + (w * 4 + k) * 4 + j] = acc[w][k][j];
  }
  __syncthreads();
  if (ps == 0 && cok) {
    float* out = a.slab + (long)blockIdx.y * a.slab_stride;
#pragma unroll
    for (int w = 0; w < KW; ++w)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (c + k >= a.wideC) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (j < a.J)
            out[(long)(kh * KW + w) * a.sT + (long)(c + k) * a.sC + j * a.sJ] =
                acc[w][k][j] + red[slot + (w * 4 + k) * 4 + j];
      }
  }
}

// The same weight gradient on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32, exact fp32 products): per filter row kh a GEMM
//     dW[(kh, kw)][d][j] = sum_pix  wide[pix][d] * narrow[pix + sgn * ((kh, kw) - pad)][j]
// with M = wide channels, N = (kw, j) (15 of 16 columns for a 5 x 5 RGB layer) and K = pixels.  A lane loads one float4 of
// wide (4 channels of one of the group's 4 pixels: a wave-load is 4 pixels x 256 contiguous bytes) and feeds its elements to
// four MFMAs -- 64 channels per wave, the channel of output row i of MFMA s is 4 i + s -- against one ds_read_b32 per filter
// row of the narrow tile (a lane reads the pixel its column's kw points at).  KH x 4 accumulator tiles per wave; a
// workgroup is 2 channel groups x 2 pixel streams, the streams are added through LDS at the end.  Same arguments, slabs
// and split as conv_outer2_kernel.  5.4 GFLOP for the DCGAN RGB-out layer: 34 us at the fp32 matrix peak.
template <int ACT, int KH>
__global__ __launch_bounds__(256) void conv_outer_mfma_kernel(Outer2Args a) {
  extern __shared__ __attribute__((aligned(16))) float4 s_nm[];   // narrow tile [(TRo + 4)][(W + 4)] float4; then the stream reduction
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cg = wave & 1, ps = wave >> 1;
  const int li = lane & 15, kq = lane >> 4;
  const int d = blockIdx.x * 128 + cg * 64 + 4 * li;     // the lane's channel quad (A operand); rows 4 li + s of the outputs
  const bool cok = d < a.wideC;
  int sc = d;
  float sgnw = 1.f;
  if (a.wide_is_x && cok) {
    GatherA gm;
    gm.cmap = a.cmap; gm.Creal = a.Creal; gm.doubled = a.doubled;
    decode_map(gm, d, a.cmap ? a.cmap[d] : 0, sc, sgnw);
  }
  // the lane's B column: n = li -> (kw, j); columns past KW * J compute garbage that is never stored
  const int ncol = a.KW * a.J;
  const int kwn = li < ncol ? li / a.J : a.KW - 1, jn = li < ncol ? li - kwn * a.J : 0;
  const int LW = a.W + 2 * kOuterHalo;
  const int boff = (kOuterHalo + kq + a.sgn * (kwn - a.pw)) * 4 + jn;   // float index inside a tile row, pixel group at column 0
  f32x4 acc[KH][4];
#pragma unroll
  for (int kh = 0; kh < KH; ++kh)
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) acc[kh][s_] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int units_per_img = a.H / a.TRo;
  const int u0 = blockIdx.y * a.units_per_block;
  int u1 = u0 + a.units_per_block;
  if (u1 > a.units) u1 = a.units;
  const int groups = a.TRo * (a.W >> 2);     // 4-pixel groups of a unit
  const int logW4 = a.logW - 2;
  const float* s_nf = reinterpret_cast<const float*>(s_nm);
  for (int u = u0; u < u1; ++u) {
    const int n = u / units_per_img, r0 = (u - n * units_per_img) * a.TRo;
    const long img = (long)n * a.H * a.W;
    __syncthreads();   // previous unit's tile fully consumed
    for (int i = tid; i < (a.TRo + 2 * kOuterHalo) * LW; i += 256) {
      const int lr = i / LW, lc = i - lr * LW;
      const int r = r0 - kOuterHalo + lr, q = lc - kOuterHalo;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((unsigned)r < (unsigned)a.H && (unsigned)q < (unsigned)a.W) {
        const float* np = a.narrow + (img + (long)r * a.W + q) * a.ldn;
        v.x = np[0];
        if (a.J > 1) v.y = np[1];
        if (a.J > 2) v.z = np[2];
        if (a.J > 3) v.w = np[3];
        if (!a.wide_is_x) {   // the narrow operand is the layer input: pre-activation applies to it
          v.x = act_apply<ACT>(v.x); v.y = act_apply<ACT>(v.y); v.z = act_apply<ACT>(v.z); v.w = act_apply<ACT>(v.w);
        }
      }
      s_nm[i] = v;
    }
    __syncthreads();
    // two groups per iteration: both loads in flight before the first MFMA
    for (int gq = ps; gq < groups; gq += 4) {
      float4 wv[2];
      int row[2], col[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int gi = gq + 2 * h;
        row[h] = gi >> logW4;
        col[h] = (gi & ((a.W >> 2) - 1)) << 2;
        wv[h] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (cok && gi < groups)
          wv[h] = *reinterpret_cast<const float4*>(a.wide + (img + (long)(r0 + row[h]) * a.W + col[h] + kq) * a.ldw + sc);
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (gq + 2 * h >= groups) break;   // block-uniform
        if (a.wide_is_x) {
          wv[h].x = act_apply<ACT>(sgnw * wv[h].x); wv[h].y = act_apply<ACT>(sgnw * wv[h].y);
          wv[h].z = act_apply<ACT>(sgnw * wv[h].z); wv[h].w = act_apply<ACT>(sgnw * wv[h].w);
        }
        const float* nb = s_nf + ((row[h] + kOuterHalo) * LW + col[h]) * 4 + boff;
        float bv[KH];
#pragma unroll
        for (int kh = 0; kh < KH; ++kh) bv[kh] = nb[a.sgn * (kh - a.ph) * LW * 4];
#pragma unroll
        for (int kh = 0; kh < KH; ++kh) {
          acc[kh][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[h].x, bv[kh], acc[kh][0], 0, 0, 0);
          acc[kh][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[h].y, bv[kh], acc[kh][1], 0, 0, 0);
          acc[kh][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[h].z, bv[kh], acc[kh][2], 0, 0, 0);
          acc[kh][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[h].w, bv[kh], acc[kh][3], 0, 0, 0);
        }
      }
    }
  }
  // add the two pixel streams through LDS (stream 1 writes, stream 0 adds and stores)
  __syncthreads();
  f32x4* red = reinterpret_cast<f32x4*>(s_nm) + (size_t)cg * KH * 4 * 64;
  if (ps == 1) {
#pragma unroll
    for (int kh = 0; kh < KH; ++kh)
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) red[(kh * 4 + s_) * 64 + lane] = acc[kh][s_];
  }
  __syncthreads();
  if (ps == 0 && li < ncol) {
    float* out = a.slab + (long)blockIdx.y * a.slab_stride;
#pragma unroll
    for (int kh = 0; kh < KH; ++kh)
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) {
        const f32x4 v = acc[kh][s_] + red[(kh * 4 + s_) * 64 + lane];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int dd = blockIdx.x * 128 + cg * 64 + 4 * (4 * kq + q) + s_;   // output row i = 4 kq + q -> channel 4 i + s
          if (dd < a.wideC) out[(long)(kh * a.KW + kwn) * a.sT + (long)dd * a.sC + jn * a.sJ] = v[q];
        }
      }
  }
}

// out[i] = sum_k slab[k][i], deterministic.  A 256-thread block owns 64 groups of VW consecutive elements; its four
// waves take the splits k = w, w + 4, ... (four independent loads in flight each) and are combined through LDS in
// wave order.  (One thread per element summing all splits in sequence: 33 us for 100 slabs of 72 K floats, 0.85 TB/s,
// 51 launches per DenseNet step.)
template <int VW>
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slab, int nsplit, long n,
                                                          float* __restrict__ out) {
  typedef float VT __attribute__((ext_vector_type(VW)));
  __shared__ VT part[4][64];
  const int e = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long i = ((long)blockIdx.x * 64 + e) * VW;
  VT acc[4];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int q = 0; q < VW; ++q) acc[u][q] = 0.f;
  if (i < n) {
    int k = w;
    for (; k + 12 < nsplit; k += 16) {
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[u] += *reinterpret_cast<const VT*>(slab + (long)(k + 4 * u) * n + i);
    }
    for (; k < nsplit; k += 4) acc[0] += *reinterpret_cast<const VT*>(slab + (long)k * n + i);
  }
  part[w][e] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  __syncthreads();
  if (w == 0 && i < n) *reinterpret_cast<VT*>(out + i) = (part[0][e] + part[1][e]) + (part[2][e] + part[3][e]);
}
static void launch_slab_reduce(const float* slab, int nsplit, long n, float* out, hipStream_t s) {
  const bool v4 = n % 4 == 0 && (reinterpret_cast<uintptr_t>(slab) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;
  if (v4) hipLaunchKernelGGL(slab_reduce_kernel<4>, dim3((unsigned)ceil_div_l(n, 256)), dim3(256), 0, s, slab, nsplit, n, out);
  else hipLaunchKernelGGL(slab_reduce_kernel<1>, dim3((unsigned)ceil_div_l(n, 64)), dim3(256), 0, s, slab, nsplit, n, out);
}

// dx[n,h,w,c] (+)= sum over the 2x2 replicas of dxv[n,2h+i,2w+j,c]  (legacy upsample dgrad)
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
// upsample folding: weight transforms
// ---------------------------------------------------------------------------------------
// Source-pixel offset of filter tap k for output parity p:  floor((p + k - pad) / 2).
__host__ __device__ inline int fold_delta(int p, int k, int pad) {
  const int v = p + k - pad;
  return v >= 0 ? v / 2 : -((1 - v) / 2);
}

struct FoldTab {
  int KH, KW, pad_t, pad_l;
  int nth[2], ntw[2];      // taps per parity along h / w
  int dmin_h[2], dmin_w[2];  // smallest delta per parity
  long woff[kMaxClass];    // class offsets (elements), class = ph*2 + pw
  long total;              // total elements of weff
};

// weff[cls][th][tw][ci][co] = sum of w[kh][kw][ci][co] over the taps folded into (th, tw)
__global__ void fold_weights_kernel(const float* __restrict__ w, FoldTab f, long CkCout,
                                    float* __restrict__ weff) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < f.total;
       i += (long)gridDim.x * blockDim.x) {
    int cls = 0;
#pragma unroll
    for (int c = 1; c < kMaxClass; ++c)
      if (i >= f.woff[c]) cls = c;
    const int ph = cls >> 1, pw = cls & 1;
    const long rem = i - f.woff[cls];
    const int tap = (int)(rem / CkCout);
    const long e = rem - (long)tap * CkCout;
    const int th = tap / f.ntw[pw], tw = tap - th * f.ntw[pw];
    const int dh = f.dmin_h[ph] + th, dw = f.dmin_w[pw] + tw;
    float s = 0.f;
    for (int kh = 0; kh < f.KH; ++kh) {
      if (fold_delta(ph, kh, f.pad_t) != dh) continue;
      for (int kw = 0; kw < f.KW; ++kw)
        if (fold_delta(pw, kw, f.pad_l) == dw) s += w[(long)(kh * f.KW + kw) * CkCout + e];
    }
    weff[i] = s;
  }
}

// dw[kh][kw][ci][co] = sum over the four classes of dweff[cls][tap(kh,kw)][ci][co]
__global__ void unfold_wgrad_kernel(const float* __restrict__ dweff, FoldTab f, long CkCout,
                                    float* __restrict__ dw) {
  const long total = (long)f.KH * f.KW * CkCout;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int k = (int)(i / CkCout);
    const long e = i - (long)k * CkCout;
    const int kh = k / f.KW, kw = k - kh * f.KW;
    float s = 0.f;
#pragma unroll
    for (int cls = 0; cls < kMaxClass; ++cls) {
      const int ph = cls >> 1, pw = cls & 1;
      const int th = fold_delta(ph, kh, f.pad_t) - f.dmin_h[ph];
      const int tw = fold_delta(pw, kw, f.pad_l) - f.dmin_w[pw];
      s += dweff[f.woff[cls] + (long)(th * f.ntw[pw] + tw) * CkCout + e];
    }
    dw[i] = s;
  }
}

// out[c][r] = in[r][c]   (32x32 LDS tiles)
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, int R, int C,
                                                        float* __restrict__ out) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    tile[ty + 8 * i][tx] = (r < R && c < C) ? in[(long)r * C + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, r = r0 + tx;
    if (r < R && c < C) out[(long)c * R + r] = tile[tx][ty + 8 * i];
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
inline int pack_dhw(int dh, int dw) { return (dh << 16) | (dw & 0xffff); }

// 16-byte gathers need aligned quads of effective channels to be contiguous in the source
inline bool map_quads(const otgan_conv_desc* d) { return !doubled_act(d->preact) || d->list_quads != 0; }
struct Geo {
  int Hin, Win, OH, OW, pad_t, pad_l, Ceff, logUp;
  bool fold;
};
int make_geo(const otgan_conv_desc* d, Geo* g) {
  OTGAN_CHECK_ARG(d, "null desc");
  OTGAN_CHECK_ARG(d->N > 0 && d->H > 0 && d->W > 0 && d->C > 0 && d->Cout > 0, "bad conv sizes");
  OTGAN_CHECK_ARG(d->stride == 1 || d->stride == 2, "stride must be 1 or 2");
  OTGAN_CHECK_ARG(d->upsample == 0 || d->upsample == 1, "upsample must be 0 or 1");
  OTGAN_CHECK_ARG(d->KH >= 1 && d->KW >= 1 && d->KH * d->KW <= 25, "filter too large");
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
  // Upsample folding is used whenever the vectorised gathers apply (deterministic in the
  // descriptor: the caller must then supply folded weights, see otgan_layers.h).
  g->fold = d->upsample == 1 && d->stride == 1 && map_quads(d) && g->Ceff % 16 == 0 && d->ldx % 4 == 0 &&
            d->Cout % 4 == 0 && d->ldy % 4 == 0 && d->y_coff % 4 == 0 && d->KH >= 2 && d->KW >= 2;
  return OTGAN_OK;
}

FoldTab make_fold(const otgan_conv_desc* d, const Geo& g) {
  FoldTab f;
  memset(&f, 0, sizeof(f));
  f.KH = d->KH; f.KW = d->KW; f.pad_t = g.pad_t; f.pad_l = g.pad_l;
  for (int p = 0; p < 2; ++p) {
    f.dmin_h[p] = fold_delta(p, 0, g.pad_t);
    f.nth[p] = fold_delta(p, d->KH - 1, g.pad_t) - f.dmin_h[p] + 1;
    f.dmin_w[p] = fold_delta(p, 0, g.pad_l);
    f.ntw[p] = fold_delta(p, d->KW - 1, g.pad_l) - f.dmin_w[p] + 1;
  }
  const long CkCout = (long)g.Ceff * d->Cout;
  long off = 0;
  for (int cls = 0; cls < 4; ++cls) {
    f.woff[cls] = off;
    off += (long)f.nth[cls >> 1] * f.ntw[cls & 1] * CkCout;
  }
  f.total = off;
  return f;
}

// Winograd F(2x2,3x3) applies to folded 5x5 upsampling layers without pre-activation.
inline bool wino_ok(const otgan_conv_desc* d, const Geo& g) {
  return g.fold && d->KH == 5 && d->KW == 5 && d->preact == OTGAN_ACT_NONE && d->C % 32 == 0 &&
         d->Cout % 4 == 0 && d->H % kWinoM == 0 && d->W % kWinoM == 0 && d->H >= kWinoM && d->W >= kWinoM && WINO(winograd_enabled)();
}
inline WinoGeo wino_geo(const otgan_conv_desc* d) {
  WinoGeo w;
  w.N = d->N; w.H = d->H; w.W = d->W; w.Cin = d->C; w.Cout = d->Cout; w.ldx = d->ldx; w.ldy = d->ldy;
  w.y_coff = d->y_coff;
  w.x_amax = d->x_amax; w.dy_amax = d->dy_amax;
  w.w_amax = d->w_amax;
  return w;
}

// 5x5 stride-2 layers (single-tensor input): four 3x3 sub-convolutions in Winograd form.
// Wide 3x3 stride-1 layers (the block-input convolution of a DenseNet block: ops.py DenseBlockFunction) take the same
// three passes with ONE class ("plain", winograd.h): 2.25 instead of 9 products per output.  Narrow ones do not pay
// (the Winograd-domain result is 2.25 x Cout floats per pixel, written and read once): Cout >= 128.
constexpr int wino_plain3_min_ceff() { return 64; }
constexpr int wino_plain3_min_cout() { return 128; }
inline bool wino_plain3_ok(const otgan_conv_desc* d, const Geo& g) {
  return d->stride == 1 && d->upsample == 0 && d->KH == 3 && d->KW == 3 && d->C % 4 == 0 && g.Ceff % 16 == 0 &&
         g.Ceff >= wino_plain3_min_ceff() && d->Cout % 32 == 0 && d->Cout >= wino_plain3_min_cout() && d->H % kWinoM == 0 && d->W % kWinoM == 0 && d->ldx % 4 == 0 &&
         d->ldy % 4 == 0 && d->y_coff % 4 == 0 && WINO(winograd_enabled)();
}
inline bool wino_s2_ok(const otgan_conv_desc* d, const Geo& g) {
  if (wino_plain3_ok(d, g)) return true;
  return d->stride == 2 && d->upsample == 0 && d->KH == 5 && d->KW == 5 && d->C % 4 == 0 && g.Ceff % 32 == 0 &&
         d->Cout % 4 == 0 && d->H % (2 * kWinoM) == 0 && d->W % (2 * kWinoM) == 0 && d->ldx % 4 == 0 && d->ldy % 4 == 0 &&
         d->y_coff % 4 == 0 && WINO(winograd_enabled)();
}
inline WinoS2Geo wino_s2_geo(const otgan_conv_desc* d, const Geo& g) {
  WinoS2Geo w;
  w.N = d->N; w.H = d->H; w.W = d->W; w.C = d->C; w.Ceff = g.Ceff; w.doubled = doubled_act(d->preact) ? 1 : 0;
  w.act = act_kind(d->preact); w.ldx = d->ldx; w.Cout = d->Cout; w.ldy = d->ldy; w.y_coff = d->y_coff;
  w.x_amax = d->x_amax; w.dy_amax = d->dy_amax;
  w.x_amax_count = d->x_amax_count > 1 ? d->x_amax_count : 1;
  w.dy_amax_count = d->dy_amax_count > 1 ? d->dy_amax_count : 1;
  w.y_amax_out = d->y_amax_out; w.dx_amax_out = d->dx_amax_out;
  w.w_amax = d->w_amax;
  w.plain = d->stride == 1 ? 1 : 0;
  return w;
}
inline double wino_s2_blocks(const WinoS2Geo& w) { return w.plain ? kWinoFreq : kWinoS2Blocks; }

// 3x3 on a 2x upsampled image with CReLU (DenseNet generator transitions): forward through Winograd when the caller
// hands over filters prepared from the UN-folded weights (otgan_conv2d_filter_bytes(d, 2) > 0)
inline bool wino_up3_ok(const otgan_conv_desc* d, const Geo& g) {
  return d->upsample == 1 && d->stride == 1 && d->KH == 3 && d->KW == 3 && d->preact == OTGAN_ACT_CRELU && d->C % 4 == 0 &&
         g.Ceff == 2 * d->C && g.Ceff % 32 == 0 && d->Cout % 4 == 0 && (2 * d->H) % kWinoM == 0 && (2 * d->W) % kWinoM == 0 &&
         d->ldx % 4 == 0 && d->ldy % 4 == 0 && d->y_coff % 4 == 0 && WINO(winograd_enabled)();
}
inline WinoUp3Geo wino_up3_geo(const otgan_conv_desc* d, const Geo& g) {
  WinoUp3Geo w;
  w.N = d->N; w.H = d->H; w.W = d->W; w.C = d->C; w.Ceff = g.Ceff; w.ldx = d->ldx; w.Cout = d->Cout; w.ldy = d->ldy;
  w.y_coff = d->y_coff; w.x_amax = d->x_amax;
  w.y_amax_out = d->y_amax_out;
  w.w_amax = d->w_amax;
  return w;
}

// ... and their weight gradient: the one-class ("plain") passes of the strided-layer code on the upsampled grid, x read
// through the upsample by the input transform; un-folded dw directly (no dweff, no unfold pass)
inline bool wino_up3_wgrad_ok(const otgan_conv_desc* d, const Geo& g) {
  return wino_up3_ok(d, g) && d->Cout % 16 == 0 && ((long)d->N * (2 * d->H / kWinoM) * (2 * d->W / kWinoM)) % 32 == 0;
}
inline WinoS2Geo wino_up3_wgrad_geo(const otgan_conv_desc* d, const Geo& g) {
  WinoS2Geo w;
  w.N = d->N; w.H = 2 * d->H; w.W = 2 * d->W; w.C = d->C; w.Ceff = g.Ceff; w.doubled = 1; w.act = 1; w.ldx = d->ldx;
  w.Cout = d->Cout; w.ldy = d->ldy; w.y_coff = d->y_coff; w.x_amax = d->x_amax; w.dy_amax = d->dy_amax;
  w.plain = 1; w.up = 1;
  return w;
}

// ... and their input gradient, when the caller hands over filters prepared from the UN-folded weights (which = 3)
inline bool wino_up3_dgrad_ok(const otgan_conv_desc* d, const Geo& g) {
  return wino_up3_ok(d, g) && d->Cout % 4 == 0;
}

inline int outer_unit_rows(int H) { return H >= 8 ? 8 : H; }

struct WgPlan {
  int outer;       // 0 no, 1 few outputs (Cout <= 4), 2 few inputs (Cin_eff <= 4)
  int chunk, nchunks;
  bool vec, fold, narrow, n16;
  int wide;   // 0, or 160 / 224: one exact column tile for Cout just above 128 / 192 (CfgW160 / CfgW224)
  bool dense16;    // DenseNet growth layer: dense16.hip kernel, nsplit slabs
  int bk;
  int tiles_m, tiles_n, nsplit, kt_per_split, nz;
  long slab_elems;  // elements of one slab (= weff elements when folded)
  long M;           // GEMM K extent (pixels of the row grid)
};
WgPlan plan_wgrad(const otgan_conv_desc* d, const Geo& g) {
  WgPlan p;
  p.wide = 0;
  const int taps = d->KH * d->KW;
  p.dense16 = false;
  if (d->Cout == 16 && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->upsample == 0 && d->C % 4 == 0 &&
      d->ldx % 4 == 0 && d->ldy % 4 == 0 && d->y_coff % 4 == 0 && dense16_enabled()) {
    const Dense16Tiling t = dense16_tiling(d->N, d->H, d->W, g.Ceff);
    if (t.ok) {
      memset(&p, 0, sizeof(p));
      p.dense16 = true;
      p.nsplit = t.nsplit;
      p.slab_elems = (long)taps * g.Ceff * d->Cout;
      p.M = (long)d->N * d->H * d->W;
      return p;
    }
  }
  p.vec = map_quads(d) && (g.Ceff % 4 == 0) && (d->Cout % 4 == 0) && (d->ldy % 4 == 0) && (d->y_coff % 4 == 0) &&
          (d->ldx % 4 == 0) && (g.Ceff >= 32);
  p.fold = g.fold && p.vec;
  p.outer = 0;
  p.chunk = p.nchunks = 0;
  if (d->upsample == 0) {
    if (d->Cout <= 4) p.outer = 1;
    else if (g.Ceff <= 4 && d->preact == OTGAN_ACT_NONE) p.outer = 2;
  }
  if (p.outer) {
    p.fold = false;
    p.M = (long)d->N * g.OH * g.OW;
    p.nchunks = p.M >= 256 * 64 ? 256 : (int)ceil_div_l(p.M, 64);
    if (d->stride == 1) {
      // the all-taps kernel streams its pixels serially per thread: 5 waves per workgroup and 256 workgroups leave
      // ~1 wave per SIMD, nothing to hide the load latency with -> more, shorter chunks
      constexpr int want = 512;
      if (p.M >= (long)want * 64) p.nchunks = want;
    }
    p.chunk = (int)ceil_div_l(p.M, p.nchunks);
    if (d->stride == 1) {
      // whole units of TRo image rows per block (conv_outer2_kernel stages the narrow operand per unit)
      const int unit = outer_unit_rows(d->H) * d->W;
      p.chunk = ceil_div(p.chunk, unit) * unit;
    }
    p.nchunks = (int)ceil_div_l(p.M, p.chunk);
    p.slab_elems = (long)taps * g.Ceff * d->Cout;
    p.nsplit = p.nchunks;
    p.vec = false; p.narrow = false; p.n16 = false; p.bk = 16; p.tiles_m = p.tiles_n = 1; p.nz = taps; p.kt_per_split = 1;
    return p;
  }
  p.narrow = d->Cout <= 32;
  p.n16 = d->Cout <= 16;
  // vector path: 128x128x32 (256x32x16 / 256x16x16 when narrow); scalar path: BK = 16 tiles
  p.wide = 0;
  if (p.vec && d->Cout > 128 && d->Cout <= 160) p.wide = 160;
  if (p.vec && d->Cout > 192 && d->Cout <= 224) p.wide = 224;
  const int BM = p.narrow ? CfgNarrow::BM : 128,
            BN = p.wide ? p.wide : (p.n16 ? CfgN16::BN : (p.narrow ? CfgNarrow::BN : 128));
  p.bk = p.wide == 224 ? CfgW224::BK : ((p.vec && !p.narrow) ? CfgMain::BK : 16);
  if (p.fold) {
    const FoldTab f = make_fold(d, g);
    p.slab_elems = f.total;
    p.nz = 0;
    for (int cls = 0; cls < 4; ++cls) p.nz += f.nth[cls >> 1] * f.ntw[cls & 1];
    p.M = (long)d->N * d->H * d->W;
    p.tiles_m = ceil_div(g.Ceff, BM);
  } else {
    p.slab_elems = (long)taps * g.Ceff * d->Cout;
    p.M = (long)d->N * g.OH * g.OW;
    if (p.vec) {
      p.tiles_m = ceil_div(g.Ceff, BM);
      p.nz = taps;
    } else {
      p.tiles_m = ceil_div(taps * g.Ceff, BM);
      p.nz = 1;
    }
  }
  p.tiles_n = ceil_div(d->Cout, BN);
  const int nkt = (int)ceil_div_l(p.M, p.bk);
  const int blocks = p.tiles_m * p.tiles_n * p.nz;
  // Split the pixel (K) dimension so that the grid fills whole "rounds" of resident workgroups
  // (256 CUs x 2 workgroups of the 128x128x32 tile, x4 of the narrow tile): a 2.25-round grid
  // runs as long as a 3-round one.  Extra splits only cost slab traffic (weight-sized).
  const int per_round = 256 * ((p.vec && !p.narrow) ? 2 : 4);
  int want = 1;
  double best = -1.0;
  for (int ns = 1; ns <= 16; ++ns) {
    if (ns > 1 && nkt / ns < 16) break;
    const long b = (long)blocks * ns;
    const double eff = (double)b / (double)(ceil_div_l(b, per_round) * per_round) - 0.004 * (ns - 1);
    if (eff > best + 1e-9) {
      best = eff;
      want = ns;
    }
  }
  p.kt_per_split = ceil_div(nkt, want);
  p.nsplit = ceil_div(nkt, p.kt_per_split);
  return p;
}

void fill_gather_x(const otgan_conv_desc* d, const Geo& g, const float* x, const int32_t* cmap,
                   int GH, int GW, int sa, int logUp, GatherA* ga) {
  memset(ga, 0, sizeof(*ga));
  ga->x = x;
  ga->ldx = d->ldx;
  ga->H = d->H;
  ga->W = d->W;
  ga->logUp = logUp;
  ga->logGH = ilog2_exact(GH);
  ga->logGW = ilog2_exact(GW);
  ga->Mtot = d->N * GH * GW;
  ga->sa = sa;
  ga->Ck = g.Ceff;
  ga->cmap = cmap;
  ga->Creal = d->C;
  ga->doubled = doubled_act(d->preact) ? 1 : 0;
}

// class table of a plain conv: one class, taps in (kh, kw) order
void single_class(const otgan_conv_desc* d, const Geo& g, ClassTab* ct) {
  memset(ct, 0, sizeof(*ct));
  ct->ncls = 1;
  Taps& t = ct->taps[0];
  t.n = d->KH * d->KW;
  for (int kh = 0; kh < d->KH; ++kh)
    for (int kw = 0; kw < d->KW; ++kw) {
      const int i = kh * d->KW + kw;
      t.dhw[i] = pack_dhw(kh - g.pad_t, kw - g.pad_l);
    }
  ct->zbase[1] = t.n;
}

// class table of a folded (upsample) conv on the SMALL grid: class = output parity.
// boff_stride = distance between consecutive taps of a class in the B operand.
void folded_classes(const FoldTab& f, long boff_stride, const long* woff, ClassTab* ct) {
  memset(ct, 0, sizeof(*ct));
  ct->ncls = 4;
  int z = 0;
  for (int cls = 0; cls < 4; ++cls) {
    const int ph = cls >> 1, pw = cls & 1;
    Taps& t = ct->taps[cls];
    t.n = f.nth[ph] * f.ntw[pw];
    for (int th = 0; th < f.nth[ph]; ++th)
      for (int tw = 0; tw < f.ntw[pw]; ++tw) {
        const int i = th * f.ntw[pw] + tw;
        t.dhw[i] = pack_dhw(f.dmin_h[ph] + th, f.dmin_w[pw] + tw);
        t.boff[i] = (int)(i * boff_stride);
      }
    ct->oa[cls] = ph;
    ct->ob[cls] = pw;
    ct->woff[cls] = woff[cls];
    ct->zbase[cls] = z;
    z += t.n;
  }
  ct->zbase[4] = z;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <class Cfg, bool VEC, int EPI, int ACT>
void launch_igemm3(dim3 grid, hipStream_t s, const GatherA& ga, const ClassTab& ct, const WeightB& wb,
                   const EpiArgs& e) {
  using LA = ConvALoader<Cfg, Cfg::BM, VEC, ACT>;
  using LB = ConvBLoader<Cfg, Cfg::BN, VEC>;
  constexpr size_t lds = sizeof(float) * (2 * LA::FLOATS + 2 * LB::FLOATS);
  ensure_lds<conv_igemm_kernel<Cfg, VEC, EPI, ACT>>(lds);
  hipLaunchKernelGGL((conv_igemm_kernel<Cfg, VEC, EPI, ACT>), grid, dim3(Cfg::THREADS), lds, s, ga, ct, wb, e);
}
// the same launch on the split-precision main loop (vector gathers, BK = 16 configurations)
template <class Cfg, int EPI, int ACT>
void launch_igemm3_x3s(dim3 grid, hipStream_t s, const GatherA& ga, const ClassTab& ct, const WeightB& wb,
                       const EpiArgs& e) {
  static_assert(Cfg::BK == 16 && Cfg::TS == 32, "");
  if (e.amax_a && e.amax_b && (!ga.cmap || (size_t)ga.Ck * sizeof(int) <= kIgemmCmapBytes)) {   // both records at hand: two scaled fp16 pieces, three MFMAs per product
    constexpr size_t lds2 = X2hLds<Cfg>::BYTES + kIgemmCmapBytes;
    ensure_lds<conv_igemm_kernel<Cfg, true, EPI, ACT, 2>>(lds2);
    hipLaunchKernelGGL((conv_igemm_kernel<Cfg, true, EPI, ACT, 2>), grid, dim3(Cfg::THREADS), lds2, s, ga, ct, wb, e);
    return;
  }
  constexpr size_t lds = X3sLds<Cfg>::BYTES;
  ensure_lds<conv_igemm_kernel<Cfg, true, EPI, ACT, 1>>(lds);
  hipLaunchKernelGGL((conv_igemm_kernel<Cfg, true, EPI, ACT, 1>), grid, dim3(Cfg::THREADS), lds, s, ga, ct, wb, e);
}
inline bool igemm_x3s() {
  static const bool on = [] {
    const char* v = getenv("OTGAN_IGEMM_X3");
    return !(v && v[0] == '0');
  }();
  return on;
}

// Picks the tile configuration for an implicit GEMM with `rows` x `cols_tiles128` output
// (cols_tiles = number of 128-wide N tiles, i.e. 64 real channels when paired) and launches.
// vec_ok: all alignment conditions for float4 gathers hold; Ck: channels per tap in K.
template <int EPI, int ACT>
void launch_igemm(bool vec_ok, int Ck, int rows, int ncols, bool paired, int ncls, hipStream_t s,
                  const GatherA& ga, const ClassTab& ct, const WeightB& wb, const EpiArgs& e) {
  const int ntiles = paired ? ceil_div(ncols, 64) : ceil_div(ncols, 128);
  if (!paired && ncols <= 32) {
    if constexpr (EPI != EPI_DG_PAIR) {
      if (ncols <= 16) {
        dim3 grid(ceil_div(rows, CfgN16::BM), 1, ncls);
        if (vec_ok && Ck % 16 == 0) launch_igemm3<CfgN16, true, EPI, ACT>(grid, s, ga, ct, wb, e);
        else launch_igemm3<CfgN16, false, EPI, ACT>(grid, s, ga, ct, wb, e);
      } else {
        dim3 grid(ceil_div(rows, CfgNarrow::BM), 1, ncls);
        if (vec_ok && Ck % 16 == 0) launch_igemm3<CfgNarrow, true, EPI, ACT>(grid, s, ga, ct, wb, e);
        else launch_igemm3<CfgNarrow, false, EPI, ACT>(grid, s, ga, ct, wb, e);
      }
    }
    return;
  }
  if constexpr (EPI == EPI_FWD) {
    // forward of the wide-but-not-256 outputs: one exact column tile instead of two 128-wide ones
    if (!paired && vec_ok && Ck % 32 == 0 && ncols > 128 && ncols <= 160 && (long)ceil_div(rows, 128) * ncls >= 256) {
      dim3 grid(ceil_div(rows, CfgW160::BM), 1, ncls);
      if (igemm_x3s()) launch_igemm3_x3s<CfgW160k16, EPI, ACT>(grid, s, ga, ct, wb, e);
      else launch_igemm3<CfgW160, true, EPI, ACT>(grid, s, ga, ct, wb, e);
      return;
    }
    if (!paired && vec_ok && Ck % 16 == 0 && ncols > 192 && ncols <= 224 && (long)ceil_div(rows, 128) * ncls >= 256) {
      dim3 grid(ceil_div(rows, CfgW224::BM), 1, ncls);
      if (igemm_x3s()) launch_igemm3_x3s<CfgW224, EPI, ACT>(grid, s, ga, ct, wb, e);
      else launch_igemm3<CfgW224, true, EPI, ACT>(grid, s, ga, ct, wb, e);
      return;
    }
  }
  if (vec_ok && Ck % 4 == 0 && igemm_x3s()) {   // (a tap's last K tile is zero-filled past Ck)
    const long tiles128 = (long)ceil_div(rows, 128) * ntiles * ncls;
    if (tiles128 < 512) {
      dim3 grid(ceil_div(rows, CfgSmall16::BM), ntiles, ncls);
      launch_igemm3_x3s<CfgSmall16, EPI, ACT>(grid, s, ga, ct, wb, e);
    } else {
      dim3 grid(ceil_div(rows, CfgMain16::BM), ntiles, ncls);
      launch_igemm3_x3s<CfgMain16, EPI, ACT>(grid, s, ga, ct, wb, e);
    }
    return;
  }
  if (vec_ok && Ck % 32 == 0) {
    const long tiles128 = (long)ceil_div(rows, 128) * ntiles * ncls;
    if (tiles128 < 512) {
      dim3 grid(ceil_div(rows, CfgSmall::BM), ntiles, ncls);
      launch_igemm3<CfgSmall, true, EPI, ACT>(grid, s, ga, ct, wb, e);
    } else {
      dim3 grid(ceil_div(rows, CfgMain::BM), ntiles, ncls);
      launch_igemm3<CfgMain, true, EPI, ACT>(grid, s, ga, ct, wb, e);
    }
    return;
  }
  if (vec_ok && Ck % 16 == 0 && (long)ceil_div(rows, 128) * ntiles * ncls < 512) {
    // few tiles (the 8x8 -> 4x4 DenseNet transition: 64 of the 128 x 128 ones for 256 CUs): half-height tiles
    dim3 grid(ceil_div(rows, CfgSmall16::BM), ntiles, ncls);
    launch_igemm3<CfgSmall16, true, EPI, ACT>(grid, s, ga, ct, wb, e);
    return;
  }
  dim3 grid(ceil_div(rows, CfgMain16::BM), ntiles, ncls);
  if (vec_ok && Ck % 4 == 0) launch_igemm3<CfgMain16, true, EPI, ACT>(grid, s, ga, ct, wb, e);
  else launch_igemm3<CfgMain16, false, EPI, ACT>(grid, s, ga, ct, wb, e);
}

// forward: activation applied in the gather (compile-time ACT)
void launch_fwd(int act, bool vec_ok, int Ck, int rows, int ncols, int ncls, hipStream_t s,
                const GatherA& ga, const ClassTab& ct, const WeightB& wb, const EpiArgs& e) {
  if (act == 1) launch_igemm<EPI_FWD, 1>(vec_ok, Ck, rows, ncols, false, ncls, s, ga, ct, wb, e);
  else if (act == 2) launch_igemm<EPI_FWD, 2>(vec_ok, Ck, rows, ncols, false, ncls, s, ga, ct, wb, e);
  else launch_igemm<EPI_FWD, 0>(vec_ok, Ck, rows, ncols, false, ncls, s, ga, ct, wb, e);
}

template <class Cfg, int ACT, bool VEC>
void launch_wgrad3(dim3 grid, hipStream_t s, const GatherA& ga, const ClassTab& ct, const WgradArgs& a) {
  if constexpr (VEC) {
    using LA = WgALoaderV<Cfg, ACT>;
    using LB = WgBLoaderV<Cfg>;
    constexpr size_t lds = sizeof(float) * (2 * LA::FLOATS + 2 * LB::FLOATS);
    ensure_lds<conv_wgrad_kernel<Cfg, ACT>>(lds);
    hipLaunchKernelGGL((conv_wgrad_kernel<Cfg, ACT>), grid, dim3(Cfg::THREADS), lds, s, ga, ct, a);
  } else {
    using LA = WgALoaderS<Cfg, ACT>;
    using LB = WgBLoaderS<Cfg>;
    constexpr size_t lds = sizeof(float) * (2 * LA::FLOATS + 2 * LB::FLOATS);
    ensure_lds<conv_wgrad_scalar_kernel<Cfg, ACT>>(lds);
    hipLaunchKernelGGL((conv_wgrad_scalar_kernel<Cfg, ACT>), grid, dim3(Cfg::THREADS), lds, s, ga, ct, a);
  }
}
template <class Cfg, bool VEC>
void launch_wgrad(int act, dim3 grid, hipStream_t s, const GatherA& ga, const ClassTab& ct,
                  const WgradArgs& a) {
  if (act == 1) launch_wgrad3<Cfg, 1, VEC>(grid, s, ga, ct, a);
  else if (act == 2) launch_wgrad3<Cfg, 2, VEC>(grid, s, ga, ct, a);
  else launch_wgrad3<Cfg, 0, VEC>(grid, s, ga, ct, a);
}

}  // namespace

extern "C" {

size_t otgan_conv2d_folded_weight_elems(const otgan_conv_desc* d) {
  Geo g;
  if (make_geo(d, &g) != OTGAN_OK || !g.fold) return 0;
  return (size_t)make_fold(d, g).total;
}

int otgan_conv2d_fold_weights_f32(const otgan_conv_desc* d, const float* w, float* weff,
                                  float* weffT, void* stream) {
  Geo g;
  int rc = make_geo(d, &g);
  if (rc) return rc;
  OTGAN_CHECK_ARG(g.fold, "this layer is not folded (otgan_conv2d_folded_weight_elems == 0)");
  OTGAN_CHECK_ARG(w && weff && weffT, "null pointer");
  hipStream_t s = (hipStream_t)stream;
  const FoldTab f = make_fold(d, g);
  const long CkCout = (long)g.Ceff * d->Cout;
  ProfScope ps(OTGAN_PROF_POINTWISE, 0.0, 4.0 * 3 * (double)f.total, s);
  long blocks = ceil_div_l(f.total, 256 * 4);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(fold_weights_kernel, dim3((int)blocks), dim3(256), 0, s, w, f, CkCout, weff);
  for (int cls = 0; cls < 4; ++cls) {
    const int R = f.nth[cls >> 1] * f.ntw[cls & 1] * g.Ceff;  // K of the class
    dim3 grid(ceil_div(d->Cout, 32), ceil_div(R, 32));
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, s, weff + f.woff[cls], R, d->Cout,
                       weffT + f.woff[cls]);
  }
  OTGAN_CHECK_LAUNCH("fold weights");
  return OTGAN_OK;
}

// K splits of the generic forward pass (0 / 1 = none): vector path on the 64 x 128 split-precision tile with fewer than
// 256 tiles, at least 16 channel slices per split, plain output grid (no upsample / fold), columns a multiple of 4
static int igemm_fwd_ksplit(const otgan_conv_desc* d, const Geo& g, bool vec, long Mtot) {
  if (!vec || g.fold || d->upsample || d->Cout % 4 || d->ldy % 4 || d->y_coff % 4 || g.Ceff % 4 || !igemm_x3s()) return 1;
  if (d->Cout <= 32) return 1;
  const long tiles = (long)ceil_div((int)Mtot, 64) * ceil_div(d->Cout, 128);
  // (counting the 128 x 160 / 128 x 224 tiles of the exact-width configurations instead was measured: the 512-tile
  // 32x32 -> 16x16 transition does not gain from a split -- 552 us either way -- and the 128-tile one loses: 229 -> 316 us)
  // Round 4, late: these kernels run one k step per global-load round trip (prefetch depth one), so with two workgroups
  // per compute unit they are bound by memory LATENCY (the 16x16 -> 8x8 DenseNet transition: 512 tiles, 450 k steps of
  // 1 us each, MFMA busy 0.14).  More workgroups in flight hide it: K splits up to 2048 workgroups (was: only below 256
  // tiles, up to 512), also for the shapes of the exact-width column tiles (Cout 129 - 160, 193 - 224), slices of at
  // least 8 x 16 channels.  Measured on the DenseNet step's forward transitions: 41.6 -> 32.2 ms per 18 steps.
  constexpr int thr = 2048, target = 2048, minsl = 8;
  if (tiles >= thr) return 1;
  const int nsl = ceil_div(g.Ceff, 16);
  int ks = (int)(target / tiles);
  if (ks > nsl / minsl) ks = nsl / minsl;
  if (ks > 8) ks = 8;
  return ks < 2 ? 1 : ks;
}

size_t otgan_conv2d_workspace_bytes(const otgan_conv_desc* d, int which) {
  Geo g;
  if (make_geo(d, &g) != OTGAN_OK) return 0;
  if (wino_ok(d, g)) {
    const WinoGeo w = wino_geo(d);
    const size_t fl = which == 0 ? WINO(wino_fwd_ws_floats)(w) : which == 1 ? WINO(wino_dgrad_ws_floats)(w)
                                                                      : WINO(wino_wgrad_ws_floats)(w) + make_fold(d, g).total;
    return align_up(sizeof(float) * fl, 256) + 256;
  }
  size_t s2 = 0;   // the strided Winograd path falls back to the generic one for list inputs: max of both
  if (which == 0 && wino_up3_ok(d, g)) s2 = align_up(sizeof(float) * WINO(wino_up3_fwd_ws_floats)(wino_up3_geo(d, g)), 256) + 256;
  if (which == 1 && wino_up3_dgrad_ok(d, g)) s2 = align_up(sizeof(float) * WINO(wino_s2_dgrad_ws_floats)(wino_up3_wgrad_geo(d, g)), 256) + 256;
  if (which == 2 && wino_up3_wgrad_ok(d, g)) s2 = align_up(sizeof(float) * WINO(wino_s2_wgrad_ws_floats)(wino_up3_wgrad_geo(d, g)), 256) + 256;
  if (wino_s2_ok(d, g)) {
    const WinoS2Geo w = wino_s2_geo(d, g);
    const size_t fl = which == 0 ? WINO(wino_s2_fwd_ws_floats)(w) : which == 1 ? WINO(wino_s2_dgrad_ws_floats)(w)
                                                                         : WINO(wino_s2_wgrad_ws_floats)(w);
    s2 = align_up(sizeof(float) * fl, 256) + 256;
  }
  const size_t recs = 256 + 2 * sizeof(float) * OTGAN_AMAX_RECORD_FLOATS;   // igemm_records()
  if (which == 1) {
    // legacy (un-folded) dgrad through a 2x upsample: gradient on the virtual grid
    const size_t gen = ((d->upsample && !g.fold)
               ? align_up(sizeof(float) * (size_t)d->N * g.Hin * g.Win * d->C, 256)
               : 0) + recs;
    return gen > s2 ? gen : s2;
  }
  if (which == 2) {
    const WgPlan p = plan_wgrad(d, g);
    size_t slabs = (p.nsplit > 1 || p.fold) ? (size_t)p.slab_elems * p.nsplit : 0;
    if (p.fold && p.nsplit > 1) slabs += (size_t)p.slab_elems;  // reduced dweff before unfolding
    const size_t gen = align_up(sizeof(float) * slabs, 256) + 256;
    return gen > s2 ? gen : s2;
  }
  size_t gen = recs;
  if (which == 0) {   // K-split partial sums of a few-tile forward pass behind the scratch records
    const bool vec = map_quads(d) && (g.Ceff % 16 == 0) && (d->ldx % 4 == 0);
    const long Mtot = (long)d->N * g.OH * g.OW;
    const int ks = igemm_fwd_ksplit(d, g, vec, Mtot);
    if (ks > 1) gen = align_up(recs, 256) + sizeof(float) * (size_t)ks * Mtot * d->Cout;
  }
  return s2 > gen ? s2 : gen;
}

// growth layer that the two-scaled-fp16-piece kernel takes (given filters and an x_amax record)
static bool dense16_h2_desc_ok(const otgan_conv_desc* d) {
  return d && d->Cout == 16 && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->upsample == 0 && d->preact == OTGAN_ACT_CRELU &&
         d->list_width == 16 && d->C >= 16 && d->C % 16 == 0 && d->ldx % 4 == 0 && d->y_accumulate && d->ldy % 4 == 0 &&
         dense16_enabled() && dense16_h2_shape_ok(d->N, d->H, d->W);
}
int otgan_dense16_h2_ok(const otgan_conv_desc* d) { return dense16_h2_desc_ok(d) ? 1 : 0; }
size_t otgan_dense16_filter_bytes(int nslices) { return dense16_h2_filter_bytes(nslices); }
int otgan_dense16_prepare_filters_f32(const float* const* wT, const int* nslices, void* const* filters, int count,
                                      void* stream) {
  OTGAN_CHECK_ARG(wT && nslices && filters && count >= 1 && count <= OTGAN_DENSE16_MAX_BATCH, "1 <= count <= %d layers",
                  OTGAN_DENSE16_MAX_BATCH);
  for (int i = 0; i < count; ++i)
    OTGAN_CHECK_ARG(wT[i] && filters[i] && nslices[i] >= 1 && aligned16(wT[i]) && aligned16(filters[i]), "null / misaligned layer %d", i);
  const int rc = dense16_h2_prepare(wT, nslices, filters, count, (hipStream_t)stream);
  OTGAN_CHECK_LAUNCH("dense16 prepare filters");
  return rc;
}

size_t otgan_dense16_bwd_filter_bytes(int npairs) { return dense16_h2_bwd_filter_bytes(npairs); }
int otgan_dense16_prepare_bwd_filters_f32(const otgan_dense16_bwd_pair* pairs, int npairs, const void* const* all_fwd_filters,
                                          int nall, void* stream) {
  OTGAN_CHECK_ARG(pairs && npairs >= 1 && npairs <= 4096 && all_fwd_filters && nall >= 1 && nall <= 16, "bad pair / layer counts");
  std::vector<Dense16BwdPair> v((size_t)npairs);
  for (int i = 0; i < npairs; ++i) {
    const otgan_dense16_bwd_pair& p = pairs[i];
    OTGAN_CHECK_ARG(p.w && p.fwd_filters && p.filters && aligned16(p.w) && aligned16(p.filters) && p.nslices_src >= 1 &&
                        p.slice_index >= 0 && p.slice_index < p.nslices_src && p.pair_index >= 0,
                    "bad pair %d", i);
    v[i] = Dense16BwdPair{p.w, p.fwd_filters, p.filters, p.nslices_src, p.slice_index, p.pair_index};
  }
  for (int i = 0; i < nall; ++i) OTGAN_CHECK_ARG(all_fwd_filters[i], "null forward buffer %d", i);
  const int rc = dense16_h2_bwd_prepare(v.data(), npairs, all_fwd_filters, nall, (hipStream_t)stream);
  OTGAN_CHECK_LAUNCH("dense16 prepare bwd filters");
  return rc;
}
int otgan_dense16_bwd_slice_f32(int N, int H, int W, int npairs, const float* g, int ldg, const void* filters, const float* x,
                                int ldx, float* dx, const float* rec0, int nrec0, const float* rec1, int nrec1,
                                float* amax_out, void* stream) {
  OTGAN_CHECK_ARG(g && filters && x && dx && rec0 && nrec0 >= 1 && (rec1 || nrec1 == 0), "null pointer");
  OTGAN_CHECK_ARG(npairs >= 1 && ldg % 4 == 0 && ldx % 4 == 0 && aligned16(g) && aligned16(x) && aligned16(dx) && aligned16(filters),
                  "strides multiples of 4, 16-byte aligned buffers");
  OTGAN_CHECK_ARG(dense16_enabled() && dense16_h2_shape_ok(N, H, W), "geometry not taken by the fp16 x 2 growth kernels");
  ProfScope ps(OTGAN_PROF_CONV_DGRAD, 2.0 * (double)N * H * W * 9.0 * 16.0 * npairs * 32.0, 0.0, (hipStream_t)stream);
  const int rc = dense16_bwd_h2(N, H, W, npairs, g, ldg, filters, x, ldx, dx, rec0, nrec0, rec1, nrec1, (hipStream_t)stream, amax_out);
  OTGAN_CHECK_LAUNCH("dense16 bwd slice");
  return rc;
}

int otgan_dense16_chain_fwd_f32(int N, int H, int W, int nslices, float* buf_group, int ld, const void* const* filters,
                                float* records, void* stream) {
  OTGAN_CHECK_ARG(buf_group && filters && records && nslices >= 2 && nslices <= 17 && ld % 4 == 0 && aligned16(buf_group),
                  "bad chain arguments");
  OTGAN_CHECK_ARG(dense16_enabled() && dense16_h2_shape_ok(N, H, W), "geometry not taken by the fp16 x 2 growth kernels");
  hipStream_t s = (hipStream_t)stream;
  const int R = OTGAN_AMAX_RECORD_FLOATS;
  for (int j = 1; j < nslices; ++j)
    OTGAN_CHECK_ARG(filters[j - 1] && aligned16(filters[j - 1]), "null / misaligned filters of chain layer %d", j);
  // one launch for the whole chain where a workgroup covers an image (round 6: dense16_chain_fwd_h2_kernel; OTGAN_DENSE16_CHAIN=0,
  // a test knob, keeps the launch per layer)
  static const bool one_launch = [] { const char* e = getenv("OTGAN_DENSE16_CHAIN"); return !(e && e[0] == '0'); }();
  if (one_launch && H == W && (W == 8 || W == 16)) {
    double flop = 0.0;
    for (int j = 1; j < nslices; ++j) flop += 2.0 * (double)N * H * W * 9.0 * 32.0 * j * 16.0;
    ProfScope ps(OTGAN_PROF_CONV_FWD, flop, 0.0, s);
    if (dense16_chain_fwd_h2(N, H, W, nslices, buf_group, ld, filters, records, s)) {
      OTGAN_CHECK_LAUNCH("dense16 chain fwd (one launch)");
      return OTGAN_OK;
    }
  }
  for (int j = 1; j < nslices; ++j) {
    ProfScope ps(OTGAN_PROF_CONV_FWD, 2.0 * (double)N * H * W * 9.0 * 32.0 * j * 16.0, 0.0, s);
    const int rc = dense16_fwd_h2(N, H, W, j, buf_group, ld, filters[j - 1], records, 1 + j, buf_group, ld, 16 * j, s,
                                  records + (size_t)(1 + j) * R);
    if (rc) return rc;
  }
  OTGAN_CHECK_LAUNCH("dense16 chain fwd");
  return OTGAN_OK;
}
int otgan_dense16_chain_bwd_f32(int N, int H, int W, int nslices, float* g_group, int ldg, const float* x_group, int ldx,
                                const void* const* filters, const float* rec0, float* slice_records, void* stream) {
  OTGAN_CHECK_ARG(g_group && x_group && filters && rec0 && slice_records && nslices >= 2 && nslices <= 17 && ldg % 4 == 0 &&
                      ldx % 4 == 0 && aligned16(g_group) && aligned16(x_group),
                  "bad chain arguments");
  OTGAN_CHECK_ARG(dense16_enabled() && dense16_h2_shape_ok(N, H, W), "geometry not taken by the fp16 x 2 growth kernels");
  hipStream_t s = (hipStream_t)stream;
  const int R = OTGAN_AMAX_RECORD_FLOATS;
  for (int c = nslices - 2; c >= 0; --c) OTGAN_CHECK_ARG(filters[c] && aligned16(filters[c]), "null / misaligned filters of slice %d", c);
  static const bool one_launch = [] { const char* e = getenv("OTGAN_DENSE16_CHAIN"); return !(e && e[0] == '0'); }();
  if (one_launch && H == W && W == 8) {      // round 6: the slices last to first inside one workgroup per image (8 x 8: dense16.hip)
    double flop = 0.0;
    for (int c = nslices - 2; c >= 0; --c) flop += 2.0 * (double)N * H * W * 9.0 * 16.0 * (nslices - 1 - c) * 32.0;
    ProfScope ps(OTGAN_PROF_CONV_DGRAD, flop, 0.0, s);
    if (dense16_chain_bwd_h2(N, H, W, nslices, g_group, ldg, x_group, ldx, filters, rec0, slice_records, s)) {
      OTGAN_CHECK_LAUNCH("dense16 chain bwd (one launch)");
      return OTGAN_OK;
    }
  }
  for (int c = nslices - 2; c >= 0; --c) {
    const int nsl = nslices - 1 - c;
    ProfScope ps(OTGAN_PROF_CONV_DGRAD, 2.0 * (double)N * H * W * 9.0 * 16.0 * nsl * 32.0, 0.0, s);
    const int rc = dense16_bwd_h2(N, H, W, nsl, g_group + 16 * (c + 1), ldg, filters[c], x_group + 16 * c, ldx, g_group + 16 * c,
                                  rec0, 1, slice_records + (size_t)(c + 1) * R, nsl, s, slice_records + (size_t)c * R);
    if (rc) return rc;
  }
  OTGAN_CHECK_LAUNCH("dense16 chain bwd");
  return OTGAN_OK;
}

size_t otgan_conv2d_filter_bytes(const otgan_conv_desc* d, int which) {
  Geo g;
  if (make_geo(d, &g) != OTGAN_OK || which < 0 || which > 3) return 0;
  if (wino_s2_ok(d, g)) return which < 2 ? sizeof(float) * WINO(wino_s2_filter_floats)(wino_s2_geo(d, g), which) : 0;
  if (wino_ok(d, g)) return sizeof(float) * WINO(wino_filter_floats)(wino_geo(d), which);   // 2, 3: from un-folded weights
  if (wino_up3_ok(d, g)) {   // from the un-folded weights: forward (2), input gradient (3)
    if (which == 2) return sizeof(float) * WINO(wino_up3_filter_floats)(wino_up3_geo(d, g));
    if (which == 3 && wino_up3_dgrad_ok(d, g)) return sizeof(float) * WINO(wino_s2_filter_floats)(wino_up3_wgrad_geo(d, g), 1);
    return 0;
  }
  return 0;
}

size_t otgan_conv2d_operand_bytes(const otgan_conv_desc* d) {
  Geo g;
  if (make_geo(d, &g) != OTGAN_OK) return 0;
  if (wino_s2_ok(d, g)) return sizeof(float) * WINO(wino_s2_x_operand_floats)(wino_s2_geo(d, g));
  if (wino_ok(d, g)) return sizeof(float) * WINO(wino_x_operand_floats)(wino_geo(d));
  if (wino_up3_ok(d, g) && wino_up3_wgrad_ok(d, g)) return sizeof(float) * WINO(wino_s2_x_operand_floats)(wino_up3_wgrad_geo(d, g));
  return 0;
}
// the caller's operand buffer, when this layer's forward and weight gradient share one
static inline float* shared_x_operand(const otgan_conv_desc* d) {
  return (d->x_operand && otgan_conv2d_operand_bytes(d) > 0) ? (float*)d->x_operand : nullptr;
}

int otgan_absmax_f32(const float* x, long rows, int C, long ld, float* record, void* stream) {
  OTGAN_CHECK_ARG(x && record && aligned16(x) && aligned16(record), "null or misaligned pointer");
  OTGAN_CHECK_ARG(rows >= 1 && C >= 4 && C % 4 == 0 && (rows == 1 || (ld >= C && ld % 4 == 0)), "rows >= 1, C and ld multiples of 4, ld >= C");
  WINO(wino_absmax)(x, rows, C, ld, record, (hipStream_t)stream, false);
  OTGAN_CHECK_LAUNCH("absmax");
  return OTGAN_OK;
}

// the implicit-GEMM input-gradient epilogue writes dx itself (no pooling pass behind it: no upsample) and every element of
// dx once (all C columns: the kernel's column blocks cover them)
static bool igemm_dgrad_amax_ok(const otgan_conv_desc* d, int lddx) {
  return !d->upsample && d->C % 4 == 0 && lddx % 4 == 0;
}
int otgan_conv2d_amax_fused(const otgan_conv_desc* d, int which) {
  Geo g;
  if (!d || make_geo(d, &g)) return 0;
  if (which == 0) {
    if (d->Cout % 4 || d->ldy % 4 || d->y_coff % 4) return 0;
    if (wino_s2_ok(d, g)) return 1;                                   // the output transform of the strided / wide 3x3 passes
    if (wino_up3_ok(d, g) && !wino_ok(d, g) && !d->y_accumulate) return 1;   // ... and of the 3x3 upsampling layers (round 4; with prepared filters)
    if (d->y_accumulate) return d->Cout == 16;                        // growth layers (dense16 kernels); else reduced afterwards
    // RGB-in layer (launch_rgbin_fwd)
    if (d->preact == OTGAN_ACT_NONE && d->C == 3 && d->stride == 1 && d->upsample == 0 && d->KH == d->KW &&
        (d->KH == 5 || d->KH == 3) && d->Cout % 128 == 0 && d->W >= 16 && d->W <= 64 && 256 / d->W >= 4 &&
        256 / d->W <= d->H && d->H % (256 / d->W) == 0)
      return 1;
    // the implicit-GEMM kernel's epilogue (every layer that is neither folded, nor Winograd, nor a few-output layer)
    return !wino_ok(d, g) && !g.fold && !wino_up3_ok(d, g) && d->Cout > 4 ? 1 : 0;
  }
  if (which == 1) {
    if (d->C % 4) return 0;
    if (wino_s2_ok(d, g)) return 1;
    if (wino_up3_dgrad_ok(d, g) && !wino_ok(d, g)) return 1;         // 3x3 upsampling layers (with prepared filters), round 4
    if (d->Cout == 3 && d->stride == 1 && d->upsample == 0 && d->KH == d->KW && (d->KH == 3 || d->KH == 5) &&
        d->W % kFdTW == 0 && d->H % kFdTH == 0)
      return 1;                                                      // RGB-out layer (conv_fewout_dgrad_kernel), round 4
    // implicit-GEMM input gradient (round 4): everything that is not a Winograd / few-channel / folded pass
    return !wino_ok(d, g) && !g.fold && !wino_up3_dgrad_ok(d, g) && !d->upsample && d->C > 4 && d->Cout > 4 &&
           !(d->Cout == 16 && d->KH == 3 && d->KW == 3 && d->stride == 1) ? 1 : 0;
  }
  return 0;
}

int otgan_conv2d_prepare_filters_f32(const otgan_conv_desc* d, int which, const float* w, void* filters,
                                     size_t filter_bytes, void* stream) {
  Geo g;
  int rc = make_geo(d, &g);
  if (rc) return rc;
  const size_t need = otgan_conv2d_filter_bytes(d, which);
  OTGAN_CHECK_ARG(need > 0, "this layer / pass has no Winograd-domain filters");
  OTGAN_CHECK_ARG(w && filters && aligned16(w) && aligned16(filters), "null or misaligned pointer");
  if (filter_bytes < need) {
    otgan_set_error("conv2d filter buffer too small: need %zu, got %zu", need, filter_bytes);
    return OTGAN_ERR_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  if (wino_s2_ok(d, g)) {
    rc = WINO(wino_s2_prepare_filters)(wino_s2_geo(d, g), which, w, (float*)filters, s);
  } else if (!wino_ok(d, g) && wino_up3_ok(d, g)) {
    if (which == 3) rc = WINO(wino_s2_prepare_filters)(wino_up3_wgrad_geo(d, g), 1, w, (float*)filters, s);
    else rc = WINO(wino_up3_prepare_filters)(wino_up3_geo(d, g), w, (float*)filters, s);
  } else {
    const FoldTab f = make_fold(d, g);
    rc = WINO(wino_prepare_filters)(wino_geo(d), which, w, f.woff[1] - f.woff[0], (float*)filters, s);
  }
  OTGAN_CHECK_LAUNCH("conv2d prepare filters");
  return rc;
}

static int conv2d_fwd_impl(const otgan_conv_desc* d, const float* x, const int32_t* cmap, const float* wT,
                           const float* filters, const float* bias, float* y, void* workspace,
                           size_t workspace_bytes, void* stream);

int otgan_conv2d_fwd_f32(const otgan_conv_desc* d, const float* x, const int32_t* cmap,
                         const float* wT, const float* bias, float* y, void* workspace,
                         size_t workspace_bytes, void* stream) {
  return conv2d_fwd_impl(d, x, cmap, wT, nullptr, bias, y, workspace, workspace_bytes, stream);
}

int otgan_conv2d_fwd_pf_f32(const otgan_conv_desc* d, const float* x, const int32_t* cmap,
                            const float* wT, const void* filters, const float* bias, float* y,
                            void* workspace, size_t workspace_bytes, void* stream) {
  OTGAN_CHECK_ARG(filters == nullptr || aligned16(filters), "misaligned filter buffer");
  return conv2d_fwd_impl(d, x, cmap, wT, (const float*)filters, bias, y, workspace, workspace_bytes, stream);
}

int otgan_conv2d_glu_fused(const otgan_conv_desc* d) {
  Geo g;
  if (!d || make_geo(d, &g) != OTGAN_OK) return 0;
  return wino_ok(d, g) && d->Cout % 8 == 0 && d->y_coff == 0 && d->ldy == d->Cout && !d->y_accumulate;
}

static int conv2d_fwd_body(const otgan_conv_desc* d, const float* x, const int32_t* cmap, const float* wT,
                           const float* filters, const float* bias, float* y, void* workspace,
                           size_t workspace_bytes, void* stream);
static int conv2d_fwd_impl(const otgan_conv_desc* d, const float* x, const int32_t* cmap, const float* wT,
                           const float* filters, const float* bias, float* y, void* workspace,
                           size_t workspace_bytes, void* stream) {
  if (d && d->y_amax_out) {
    OTGAN_CHECK_ARG(d->Cout % 4 == 0 && d->ldy % 4 == 0 && d->y_coff % 4 == 0 && aligned16(y),
                    "y_amax_out: needs Cout, ldy, y_coff multiples of 4 and a 16-byte aligned y");
  }
  if (d && d->glu_out) {
    OTGAN_CHECK_ARG(otgan_conv2d_glu_fused(d) && cmap == nullptr && aligned16(d->glu_out),
                    "glu_out: only where otgan_conv2d_glu_fused(d) says so (single-tensor input, 16-byte aligned glu_out)");
  }
  g_amax_written = false;
  const int rc = conv2d_fwd_body(d, x, cmap, wT, filters, bias, y, workspace, workspace_bytes, stream);
  if (rc == OTGAN_OK && d->y_amax_out && !g_amax_written) {
    // this pass has no fused record: one reduction over the output it just wrote (same value)
    Geo g;
    make_geo(d, &g);
    WINO(wino_absmax)(y + d->y_coff, (long)d->N * g.OH * g.OW, d->Cout, d->ldy, d->y_amax_out, (hipStream_t)stream, true);
    OTGAN_CHECK_LAUNCH("conv2d fwd (amax of y)");
  }
  return rc;
}
static int conv2d_fwd_body(const otgan_conv_desc* d, const float* x, const int32_t* cmap, const float* wT,
                           const float* filters, const float* bias, float* y, void* workspace,
                           size_t workspace_bytes, void* stream) {
  Geo g;
  int rc = make_geo(d, &g);
  if (rc) return rc;
  OTGAN_CHECK_ARG(x && wT && y, "null pointer");
  hipStream_t s = (hipStream_t)stream;
  const bool growth16 = d->Cout == 16 && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->upsample == 0 && d->C % 8 == 0 &&
                        d->ldx % 4 == 0 && aligned16(x) && aligned16(wT) && aligned16(cmap) && dense16_enabled();
  const bool wino_s2_fwd_taken = wino_s2_ok(d, g) && cmap == nullptr && aligned16(x) && aligned16(wT) && aligned16(y) &&
                                 aligned16(bias) && aligned16(workspace) && workspace &&
                                 workspace_bytes >= otgan_conv2d_workspace_bytes(d, 0);
  OTGAN_CHECK_ARG(!d->y_accumulate || growth16 || (wino_s2_fwd_taken && wino_plain3_ok(d, g)),
                  "y_accumulate: only the 3x3 stride-1 growth layers (16 outputs) and the wide 3x3 stride-1 layers on the "
                  "Winograd path accumulate");
  GatherA ga;
  ClassTab ct;
  WeightB wb;
  memset(&wb, 0, sizeof(wb));
  wb.w = wT;
  wb.Nvalid = d->Cout;
  wb.Ck = g.Ceff;
  EpiArgs e;
  memset(&e, 0, sizeof(e));
  e.out = y; e.ldo = d->ldy; e.coff = d->y_coff;
  e.OHf = g.OH; e.OWf = g.OW;
  e.bias = bias; e.ncols = d->Cout;
  double flops;
  bool vec;
  if (wino_s2_fwd_taken) {
    WinoS2Geo w = wino_s2_geo(d, g);
    w.y_accumulate = d->y_accumulate;
    w.x_op = shared_x_operand(d);
    ProfScope ps(OTGAN_PROF_CONV_FWD, 2.0 * wino_s2_blocks(w) * (double)wino_s2_tiles(w) * g.Ceff * d->Cout, 0.0, s);
    rc = WINO(wino_s2_fwd)(w, x, wT, bias, y, (float*)workspace, s, filters);
    g_amax_written = w.y_amax_out != nullptr;    // (with y_accumulate: the sums it wrote)
    OTGAN_CHECK_LAUNCH("conv2d fwd (winograd, stride 2)");
    return rc;
  }
  if (filters && wino_up3_ok(d, g) && !wino_ok(d, g)) {
    // prepared filters of a 3x3 upsampling layer are made from the UN-folded weights and the caller may pass the
    // un-folded wT with them: nothing below may run with those, so a failed precondition is an error, not a reroute
    // (as in the input gradient)
    OTGAN_CHECK_ARG(cmap == nullptr && aligned16(x) && aligned16(y) && aligned16(bias) && aligned16(workspace) && workspace &&
                        workspace_bytes >= otgan_conv2d_workspace_bytes(d, 0),
                    "winograd forward of a 3x3 upsampling layer (prepared filters given): single-tensor input, 16-byte "
                    "aligned operands, workspace of otgan_conv2d_workspace_bytes(d, 0)");
    WinoUp3Geo w = wino_up3_geo(d, g);
    w.x_op = shared_x_operand(d);
    ProfScope ps(OTGAN_PROF_CONV_FWD, 2.0 * kWinoFreq * (double)wino_up3_tiles(w) * g.Ceff * d->Cout, 0.0, s);
    rc = WINO(wino_up3_fwd)(w, x, bias, y, (float*)workspace, s, filters);
    g_amax_written = w.y_amax_out != nullptr;
    OTGAN_CHECK_LAUNCH("conv2d fwd (winograd, 3x3 on upsampled input)");
    return rc;
  }
  OTGAN_CHECK_ARG(wino_ok(d, g) || shared_x_operand(d) == nullptr,
                  "x_operand given, but this forward call cannot take the Winograd path that writes it (list input, alignment, "
                  "workspace, or no prepared filters for a 3x3 upsampling layer)");
  if (wino_ok(d, g)) {
    OTGAN_CHECK_ARG(aligned16(x) && aligned16(wT) && aligned16(y) && aligned16(bias) && aligned16(workspace),
                    "winograd conv needs 16-byte aligned operands");
    const size_t need = otgan_conv2d_workspace_bytes(d, 0);
    if (!workspace || workspace_bytes < need) {
      otgan_set_error("conv2d fwd workspace too small: need %zu, got %zu", need, workspace_bytes);
      return OTGAN_ERR_WORKSPACE;
    }
    const FoldTab f = make_fold(d, g);
    WinoGeo w = wino_geo(d);
    w.x_op = shared_x_operand(d);
    w.glu_out = d->glu_out;
    w.glu_amax = d->glu_out ? d->glu_amax_out : nullptr;
    // executed FLOP: 16 GEMMs of tiles x 4*Cout x Cin
    ProfScope ps(OTGAN_PROF_CONV_FWD, 2.0 * kWinoFreq * (double)wino_tiles(w) * 4.0 * d->Cout * d->C, 0.0, s);
    rc = WINO(wino_fwd)(w, x, wT, f.woff[1] - f.woff[0], bias, y, (float*)workspace, s, filters);
    OTGAN_CHECK_LAUNCH("conv2d fwd (winograd)");
    return rc;
  }
  if (g.fold) {
    OTGAN_CHECK_ARG(aligned16(x) && aligned16(wT), "folded conv needs 16-byte aligned operands");
    // rows = SMALL-grid pixels, one class per output parity, folded taps; wT = weffT
    const FoldTab f = make_fold(d, g);
    fill_gather_x(d, g, x, cmap, d->H, d->W, 1, 0, &ga);
    folded_classes(f, g.Ceff, f.woff, &ct);
    // per class the transposed block is [Cout][ntaps*Ceff]
    e.so = 2;
    vec = true;
    flops = 0;
    for (int cls = 0; cls < 4; ++cls) flops += 2.0 * ga.Mtot * (double)ct.taps[cls].n * g.Ceff * d->Cout;
    ProfScope ps(OTGAN_PROF_CONV_FWD, flops, 0.0, s);
    const int act = act_kind(d->preact);
    const bool same_k = ct.taps[0].n == ct.taps[1].n && ct.taps[0].n == ct.taps[2].n &&
                        ct.taps[0].n == ct.taps[3].n;
    if (same_k) {
      // the usual case (odd square filters): one launch, blockIdx.z = output parity class
      wb.ldbn = (long)ct.taps[0].n * g.Ceff;
      launch_fwd(act, vec, g.Ceff, ga.Mtot, d->Cout, 4, s, ga, ct, wb, e);
    } else {
      for (int cls = 0; cls < 4; ++cls) {
        ClassTab one;
        memset(&one, 0, sizeof(one));
        one.ncls = 1;
        one.taps[0] = ct.taps[cls];
        one.oa[0] = ct.oa[cls];
        one.ob[0] = ct.ob[cls];
        one.woff[0] = ct.woff[cls];
        one.zbase[1] = one.taps[0].n;
        wb.ldbn = (long)ct.taps[cls].n * g.Ceff;
        launch_fwd(act, vec, g.Ceff, ga.Mtot, d->Cout, 1, s, ga, one, wb, e);
      }
    }
    OTGAN_CHECK_LAUNCH("conv2d fwd (folded)");
    return OTGAN_OK;
  }
  fill_gather_x(d, g, x, cmap, g.OH, g.OW, d->stride, g.logUp, &ga);
  single_class(d, g, &ct);
  for (int t = 0; t < ct.taps[0].n; ++t) ct.taps[0].boff[t] = t * g.Ceff;
  const int Ktot = ct.taps[0].n * g.Ceff;
  if (cmap == nullptr && d->preact == OTGAN_ACT_NONE) {
    // RGB-in layer: lanes along the output channels, weights in registers
    ProfScope ps(OTGAN_PROF_CONV_FWD, 2.0 * ga.Mtot * (double)Ktot * d->Cout, 0.0, s);
    if (launch_rgbin_fwd(d, g.pad_t, g.pad_l, x, wT, bias, y, s)) {
      OTGAN_CHECK_LAUNCH("conv2d fwd (RGB in)");
      return OTGAN_OK;
    }
  }
  if (d->Cout <= 4 && d->upsample == 0 && g.Ceff % 4 == 0 && d->ldx % 4 == 0 && aligned16(x) &&
      aligned16(wT)) {
    // RGB-out layer: vector-ALU kernel, thread = pixel (an MFMA tile would idle 29 of 32 columns)
    FewOutArgs fa;
    memset(&fa, 0, sizeof(fa));
    fa.w = wT; fa.sJ = Ktot;
    fa.out = y; fa.ldo = d->ldy; fa.coff = d->y_coff; fa.J = d->Cout;
    fa.so = 1; fa.OHf = g.OH; fa.OWf = g.OW; fa.bias = bias;
    ProfScope ps(OTGAN_PROF_CONV_FWD, 2.0 * ga.Mtot * (double)Ktot * d->Cout, 0.0, s);
    const dim3 grid(ceil_div(ga.Mtot, 64));
    const int act = act_kind(d->preact);
    if (act == 1) {
      if (!launch_fewout_mfma<1>(ga, ct.taps[0], fa, d->KH, d->KW, s) && !launch_fewout_tile<1>(ga, ct.taps[0], fa, s))
        hipLaunchKernelGGL(conv_fewout_kernel<1>, grid, dim3(256), 0, s, ga, ct.taps[0], fa);
    } else if (act == 2) {
      if (!launch_fewout_mfma<2>(ga, ct.taps[0], fa, d->KH, d->KW, s) && !launch_fewout_tile<2>(ga, ct.taps[0], fa, s))
        hipLaunchKernelGGL(conv_fewout_kernel<2>, grid, dim3(256), 0, s, ga, ct.taps[0], fa);
    } else {
      if (!launch_fewout_mfma<0>(ga, ct.taps[0], fa, d->KH, d->KW, s) && !launch_fewout_tile<0>(ga, ct.taps[0], fa, s))
        hipLaunchKernelGGL(conv_fewout_kernel<0>, grid, dim3(256), 0, s, ga, ct.taps[0], fa);
    }
    OTGAN_CHECK_LAUNCH("conv2d fwd (few outputs)");
    return OTGAN_OK;
  }
  if (growth16 && filters && dense16_h2_desc_ok(d) && d->x_amax && aligned16(filters) && !bias && aligned16(y)) {
    // the chain of a split dense block on two scaled fp16 pieces (dense16.hip, round 4)
    ProfScope ps(OTGAN_PROF_CONV_FWD, 2.0 * ga.Mtot * (double)Ktot * d->Cout, 0.0, s);
    rc = dense16_fwd_h2(d->N, d->H, d->W, d->C / 16, x, d->ldx, filters, d->x_amax, d->x_amax_count > 1 ? d->x_amax_count : 1,
                        y, d->ldy, d->y_coff, s, d->y_amax_out);
    g_amax_written = d->y_amax_out != nullptr;
    OTGAN_CHECK_LAUNCH("conv2d fwd (dense16, fp16 x 2)");
    return rc;
  }
  if (growth16) {
    // DenseNet growth layer: LDS-free streaming MFMA kernel (dense16.hip)
    Dense16Geo dg;
    dg.N = d->N; dg.H = d->H; dg.W = d->W; dg.logH = ilog2_exact(d->H); dg.logW = ilog2_exact(d->W);
    dg.C = d->C; dg.Ceff = g.Ceff; dg.doubled = doubled_act(d->preact) ? 1 : 0;
    dg.act = act_kind(d->preact); dg.ldx = d->ldx; dg.cmap = cmap;
    ProfScope ps(OTGAN_PROF_CONV_FWD, 2.0 * ga.Mtot * (double)Ktot * d->Cout, 0.0, s);
    rc = dense16_fwd(dg, x, wT, bias, y, d->ldy, d->y_coff, d->y_accumulate, s, d->y_amax_out);
    g_amax_written = d->y_amax_out != nullptr;
    OTGAN_CHECK_LAUNCH("conv2d fwd (dense16)");
    return rc;
  }
  wb.ldbn = Ktot;
  e.so = 1;
  vec = map_quads(d) && (g.Ceff % 16 == 0) && (d->ldx % 4 == 0) && aligned16(x) && aligned16(wT);
  flops = 2.0 * ga.Mtot * (double)Ktot * d->Cout;
  ProfScope ps(OTGAN_PROF_CONV_FWD, flops, 0.0, s);
  const int act = act_kind(d->preact);
  if (vec && (long)Ktot * d->Cout < (1L << 31))
    igemm_records(e, d->x_amax, x, (long)d->N * d->H * d->W, d->C, d->ldx, d->w_amax, wT, (long)Ktot * d->Cout, act == 2,
                  workspace, workspace_bytes, 0, s);
  e.amax_out = d->y_amax_out;
  // K split of few-tile forward passes (round 4: the 8 x 8 -> 4 x 4 DenseNet transition is 128 tiles of 64 x 128 with
  // K = 8208 -- one tile per second compute unit, 472 us at 9 % of the matrix pipe): channel slices over blockIdx.z,
  // partial sums behind the amax scratch records in the workspace, a float4 finish kernel with bias and amax record
  const int ks = igemm_fwd_ksplit(d, g, vec, ga.Mtot);
  if (ks > 1) {
    const size_t at = align_up(256 + 2 * sizeof(float) * OTGAN_AMAX_RECORD_FLOATS, 256);
    const size_t need = at + sizeof(float) * (size_t)ks * ga.Mtot * d->Cout;
    if (workspace && workspace_bytes >= need && aligned16(workspace) && aligned16(y) && aligned16(bias)) {
      e.ksplit = ks;
      e.partial = reinterpret_cast<float*>(static_cast<char*>(workspace) + at);
      launch_fwd(act, vec, g.Ceff, ga.Mtot, d->Cout, ks, s, ga, ct, wb, e);
      const long total4 = (long)ga.Mtot * (d->Cout / 4);
      const int blocks = (int)(ceil_div_l(total4, 256) < 2048 ? ceil_div_l(total4, 256) : 2048);
      hipLaunchKernelGGL(igemm_splitk_finish_kernel, dim3(blocks), dim3(256), 0, s, e.partial, ks, (long)ga.Mtot, d->Cout, bias, y,
                         d->ldy, d->y_coff, d->y_amax_out);
      g_amax_written = d->y_amax_out != nullptr;
      OTGAN_CHECK_LAUNCH("conv2d fwd (K split)");
      return OTGAN_OK;
    }
  }
  launch_fwd(act, vec, g.Ceff, ga.Mtot, d->Cout, 1, s, ga, ct, wb, e);
  g_amax_written = e.amax_out != nullptr;
  OTGAN_CHECK_LAUNCH("conv2d fwd");
  return OTGAN_OK;
}

static int conv2d_dgrad_impl(const otgan_conv_desc* d, const float* dy, const float* w, const float* filters,
                             const float* x, const int32_t* inv, float* dx, int lddx, int accumulate,
                             void* workspace, size_t workspace_bytes, void* stream);

int otgan_conv2d_dgrad_f32(const otgan_conv_desc* d, const float* dy, const float* w,
                           const float* x, const int32_t* inv, float* dx, int lddx,
                           int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
  return conv2d_dgrad_impl(d, dy, w, nullptr, x, inv, dx, lddx, accumulate, workspace, workspace_bytes, stream);
}

int otgan_conv2d_dgrad_pf_f32(const otgan_conv_desc* d, const float* dy, const float* w, const void* filters,
                              const float* x, const int32_t* inv, float* dx, int lddx,
                              int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
  OTGAN_CHECK_ARG(filters == nullptr || aligned16(filters), "misaligned filter buffer");
  return conv2d_dgrad_impl(d, dy, w, (const float*)filters, x, inv, dx, lddx, accumulate, workspace, workspace_bytes,
                           stream);
}

static int conv2d_dgrad_body(const otgan_conv_desc* d, const float* dy, const float* w, const float* filters,
                             const float* x, const int32_t* inv, float* dx, int lddx, int accumulate,
                             void* workspace, size_t workspace_bytes, void* stream);
static int conv2d_dgrad_impl(const otgan_conv_desc* d, const float* dy, const float* w, const float* filters,
                             const float* x, const int32_t* inv, float* dx, int lddx, int accumulate,
                             void* workspace, size_t workspace_bytes, void* stream) {
  if (d && d->dx_amax_out) {
    OTGAN_CHECK_ARG(d->C % 4 == 0 && lddx % 4 == 0 && aligned16(dx),
                    "dx_amax_out: needs C and lddx multiples of 4 and a 16-byte aligned dx");
  }
  g_amax_written = false;
  const int rc = conv2d_dgrad_body(d, dy, w, filters, x, inv, dx, lddx, accumulate, workspace, workspace_bytes, stream);
  if (rc == OTGAN_OK && d->dx_amax_out && !g_amax_written) {
    WINO(wino_absmax)(dx, (long)d->N * d->H * d->W, d->C, lddx, d->dx_amax_out, (hipStream_t)stream, true);
    OTGAN_CHECK_LAUNCH("conv2d dgrad (amax of dx)");
  }
  return rc;
}
static int conv2d_dgrad_body(const otgan_conv_desc* d, const float* dy, const float* w, const float* filters,
                             const float* x, const int32_t* inv, float* dx, int lddx, int accumulate,
                             void* workspace, size_t workspace_bytes, void* stream) {
  Geo g;
  int rc = make_geo(d, &g);
  if (rc) return rc;
  OTGAN_CHECK_ARG(dy && w && dx && lddx >= d->C, "null pointer / bad lddx");
  const int kind = act_kind(d->preact);
  OTGAN_CHECK_ARG(kind == 0 || x, "dgrad through a pre-activation needs the layer input x");
  hipStream_t s = (hipStream_t)stream;
  const bool paired = doubled_act(d->preact);
  // A operand: dy, one K block per (tap, co)
  GatherA ga;
  memset(&ga, 0, sizeof(ga));
  ga.x = dy + d->y_coff;
  ga.ldx = d->ldy;
  ga.H = g.OH; ga.W = g.OW; ga.logUp = 0;
  ga.Ck = d->Cout;
  ga.cmap = nullptr; ga.Creal = d->Cout; ga.doubled = 0;
  WeightB wb;
  memset(&wb, 0, sizeof(wb));
  wb.w = w;
  wb.ldbn = d->Cout;
  wb.Ck = d->Cout;
  wb.paired = paired ? 1 : 0;
  wb.Creal = d->C;
  wb.inv = inv;
  wb.Nvalid = g.Ceff;
  EpiArgs e;
  memset(&e, 0, sizeof(e));
  e.ncols = d->C;
  e.xsrc = x; e.ldxs = d->ldx; e.xH = d->H; e.xW = d->W;
  e.act = kind;
  // float4 gathers of dy / w along the output channels; a tap's last K tile is zero-filled past Cout
  const bool vec = (d->Cout % 4 == 0) && (d->ldy % 4 == 0) && (d->y_coff % 4 == 0) &&
                   aligned16(dy) && aligned16(w);
  ClassTab ct;
  memset(&ct, 0, sizeof(ct));
  float* target = dx;
  bool pool = false;
  double flops = 0;
  if (d->C <= 4 && kind == 0 && d->stride == 1 && d->upsample == 0 && d->Cout % 4 == 0 &&
      d->ldy % 4 == 0 && d->y_coff % 4 == 0 && aligned16(dy) && aligned16(w)) {
    // RGB-in layer: dx[pix][ci<4] = sum_{tap, co} dy[pix - tap][co] * W[tap][ci][co] on the VALU
    ga.logGH = ilog2_exact(g.Hin);
    ga.logGW = ilog2_exact(g.Win);
    ga.Mtot = d->N * g.Hin * g.Win;
    ga.sa = 1;
    Taps& t = ct.taps[0];
    t.n = d->KH * d->KW;
    for (int kh = 0; kh < d->KH; ++kh)
      for (int kw = 0; kw < d->KW; ++kw) {
        const int i = kh * d->KW + kw;
        t.dhw[i] = pack_dhw(g.pad_t - kh, g.pad_l - kw);
        t.boff[i] = i * g.Ceff * d->Cout;
      }
    FewOutArgs fa;
    memset(&fa, 0, sizeof(fa));
    fa.w = w; fa.sJ = d->Cout;
    fa.out = dx; fa.ldo = lddx; fa.coff = 0; fa.J = d->C;
    fa.so = 1; fa.OHf = g.Hin; fa.OWf = g.Win; fa.accumulate = accumulate;
    ProfScope ps(OTGAN_PROF_CONV_DGRAD, 2.0 * ga.Mtot * (double)t.n * d->Cout * d->C, 0.0, s);
    if (!launch_fewout_mfma<0>(ga, t, fa, d->KH, d->KW, s) && !launch_fewout_tile<0>(ga, t, fa, s))
      hipLaunchKernelGGL(conv_fewout_kernel<0>, dim3(ceil_div(ga.Mtot, 64)), dim3(256), 0, s, ga, t, fa);
    OTGAN_CHECK_LAUNCH("conv2d dgrad (few inputs)");
    return OTGAN_OK;
  }
  if (d->Cout == 3 && d->stride == 1 && d->upsample == 0 && d->KH == d->KW && (d->KH == 3 || d->KH == 5) && d->C % 4 == 0 &&
      lddx % 4 == 0 && aligned16(dx) && (kind == 0 || (aligned16(x) && d->ldx % 4 == 0)) && (inv == nullptr || paired) &&
      d->W % kFdTW == 0 && d->H % kFdTH == 0) {
    // RGB-out layer: streaming kernel, lanes along the input channels
    FewDgArgs fa;
    fa.dy = dy + d->y_coff; fa.ldy = d->ldy;
    fa.w = w; fa.inv = inv;
    fa.x = kind ? x : nullptr; fa.ldx = d->ldx;
    fa.dx = dx; fa.lddx = lddx;
    fa.N = d->N; fa.H = d->H; fa.W = d->W; fa.C = d->C; fa.Ceff = g.Ceff;
    fa.pad_t = g.pad_t; fa.pad_l = g.pad_l; fa.act = kind; fa.accumulate = accumulate;
    fa.amax = d->dx_amax_out;
    g_amax_written = fa.amax != nullptr;
    const dim3 grid(d->N, d->W / kFdTW, ceil_div(d->C, kFdCH));
    ProfScope ps(OTGAN_PROF_CONV_DGRAD, 2.0 * (double)d->N * d->H * d->W * d->KH * d->KW * d->Cout * g.Ceff, 0.0, s);
    if (d->KH == 3) {
      if (paired) hipLaunchKernelGGL((conv_fewout_dgrad_kernel<3, true, 3>), grid, dim3(256), 0, s, fa);
      else hipLaunchKernelGGL((conv_fewout_dgrad_kernel<3, false, 3>), grid, dim3(256), 0, s, fa);
    } else {
      if (paired) hipLaunchKernelGGL((conv_fewout_dgrad_kernel<5, true, 3>), grid, dim3(256), 0, s, fa);
      else hipLaunchKernelGGL((conv_fewout_dgrad_kernel<5, false, 3>), grid, dim3(256), 0, s, fa);
    }
    OTGAN_CHECK_LAUNCH("conv2d dgrad (few outputs)");
    return OTGAN_OK;
  }
  if (d->Cout == 16 && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->upsample == 0 && d->ldy % 4 == 0 &&
      d->y_coff % 4 == 0 && d->C % 4 == 0 && d->ldx % 4 == 0 && lddx % 4 == 0 && aligned16(dy) && aligned16(w) &&
      aligned16(x) && aligned16(dx) && dense16_enabled() && dense16_tiling(d->N, d->H, d->W, g.Ceff).ok) {
    // DenseNet growth layer: dy tile in LDS, weights streamed, pos/neg halves combined in registers
    Dense16Geo dg;
    dg.N = d->N; dg.H = d->H; dg.W = d->W; dg.logH = ilog2_exact(d->H); dg.logW = ilog2_exact(d->W);
    dg.C = d->C; dg.Ceff = g.Ceff; dg.doubled = paired ? 1 : 0;
    dg.act = kind; dg.ldx = d->ldx; dg.cmap = nullptr;
    ProfScope ps(OTGAN_PROF_CONV_DGRAD, 2.0 * (double)d->N * d->H * d->W * 9.0 * d->Cout * g.Ceff, 0.0, s);
    rc = dense16_dgrad(dg, dy, d->ldy, d->y_coff, w, x, inv, dx, lddx, accumulate, s);
    OTGAN_CHECK_LAUNCH("conv2d dgrad (dense16)");
    return rc;
  }
  if (filters && !wino_ok(d, g) && wino_up3_dgrad_ok(d, g)) {
    // prepared filters of a 3x3 upsampling layer exist only for the Winograd input gradient, and they (and `w`) are
    // made from the UN-folded weights: nothing else below may run with them
    OTGAN_CHECK_ARG(inv == nullptr && lddx % 4 == 0 && aligned16(dy) && aligned16(dx) && aligned16(x) && aligned16(workspace) &&
                        workspace && workspace_bytes >= otgan_conv2d_workspace_bytes(d, 1),
                    "winograd dgrad of a 3x3 upsampling layer: single-tensor input, 16-byte aligned operands, workspace");
    WinoS2Geo wg = wino_up3_wgrad_geo(d, g);
    wg.dx_amax_out = d->dx_amax_out;       // (round 4: a stored pixel is written by one thread of the output transform)
    ProfScope ps(OTGAN_PROF_CONV_DGRAD, 2.0 * kWinoFreq * (double)wino_s2_tiles(wg) * g.Ceff * d->Cout, 0.0, s);
    rc = WINO(wino_s2_dgrad)(wg, dy, w, x, dx, lddx, accumulate, (float*)workspace, s, filters);
    g_amax_written = wg.dx_amax_out != nullptr;
    OTGAN_CHECK_LAUNCH("conv2d dgrad (winograd, 3x3 on upsampled input)");
    return rc;
  }
  if (wino_s2_ok(d, g) && inv == nullptr && lddx % 4 == 0 && aligned16(dy) && aligned16(w) && aligned16(dx) &&
      aligned16(x) && aligned16(workspace) && workspace && workspace_bytes >= otgan_conv2d_workspace_bytes(d, 1)) {
    const WinoS2Geo wg = wino_s2_geo(d, g);
    ProfScope ps(OTGAN_PROF_CONV_DGRAD, 2.0 * wino_s2_blocks(wg) * (double)wino_s2_tiles(wg) * g.Ceff * d->Cout, 0.0, s);
    rc = WINO(wino_s2_dgrad)(wg, dy, w, x, dx, lddx, accumulate, (float*)workspace, s, filters);
    g_amax_written = wg.dx_amax_out != nullptr;      // (round 4: with `accumulate` the output kernel tracks the sums it leaves)
    OTGAN_CHECK_LAUNCH("conv2d dgrad (winograd, stride 2)");
    return rc;
  }
  if (wino_ok(d, g)) {
    OTGAN_CHECK_ARG(aligned16(dy) && aligned16(w) && aligned16(dx) && aligned16(workspace) && lddx % 4 == 0,
                    "winograd dgrad needs 16-byte aligned operands");
    const size_t need = otgan_conv2d_workspace_bytes(d, 1);
    if (!workspace || workspace_bytes < need) {
      otgan_set_error("conv2d dgrad workspace too small: need %zu, got %zu", need, workspace_bytes);
      return OTGAN_ERR_WORKSPACE;
    }
    const FoldTab f = make_fold(d, g);
    const WinoGeo wg = wino_geo(d);
    ProfScope ps(OTGAN_PROF_CONV_DGRAD, 2.0 * kWinoFreq * (double)wino_tiles(wg) * 4.0 * d->Cout * d->C, 0.0, s);
    rc = WINO(wino_dgrad)(wg, dy, w, f.woff[1] - f.woff[0], dx, lddx, accumulate, (float*)workspace, s, filters);
    OTGAN_CHECK_LAUNCH("conv2d dgrad (winograd)");
    return rc;
  }
  if (g.fold) {
    // gradient w.r.t. the SMALL input directly: rows = small pixels, K = all 4 classes'
    // folded taps; dy is read at (2(a - dh) + ph, 2(b - dw) + pw); w = weff.
    const FoldTab f = make_fold(d, g);
    ga.logGH = ilog2_exact(d->H);
    ga.logGW = ilog2_exact(d->W);
    ga.Mtot = d->N * d->H * d->W;
    ga.sa = 2;
    ct.ncls = 1;
    Taps& t = ct.taps[0];
    int n = 0;
    const long CkCout = (long)g.Ceff * d->Cout;
    for (int cls = 0; cls < 4; ++cls) {
      const int ph = cls >> 1, pw = cls & 1;
      for (int th = 0; th < f.nth[ph]; ++th)
        for (int tw = 0; tw < f.ntw[pw]; ++tw) {
          const int dh = f.dmin_h[ph] + th, dw = f.dmin_w[pw] + tw;
          t.dhw[n] = pack_dhw(-2 * dh + ph, -2 * dw + pw);
          t.boff[n] = (int)(f.woff[cls] + (long)(th * f.ntw[pw] + tw) * CkCout);
          ++n;
        }
    }
    t.n = n;
    ct.zbase[1] = n;
    e.out = dx; e.ldo = lddx; e.coff = 0;
    e.so = 1; e.OHf = d->H; e.OWf = d->W;
    e.logUpX = 0;
    e.accumulate = accumulate;
    flops = 2.0 * ga.Mtot * (double)n * d->Cout * g.Ceff;
  } else {
    const size_t need = otgan_conv2d_workspace_bytes(d, 1);
    int tld = lddx, tacc = accumulate;
    if (d->upsample) {
      if (!workspace || workspace_bytes < need) {
        otgan_set_error("conv2d dgrad workspace too small: need %zu, got %zu", need, workspace_bytes);
        return OTGAN_ERR_WORKSPACE;
      }
      target = (float*)workspace;  // gradient on the virtual (upsampled) grid, dense [.,C]
      tld = d->C;
      tacc = 0;
      pool = true;
    }
    const int st = d->stride;
    const int GH = g.Hin / st, GW = g.Win / st;
    ga.logGH = ilog2_exact(GH);
    ga.logGW = ilog2_exact(GW);
    OTGAN_CHECK_ARG(ga.logGH >= 0 && ga.logGW >= 0, "dgrad grid must be a power of two");
    ga.Mtot = d->N * GH * GW;
    ga.sa = 1;
    ct.ncls = st * st;
    int z = 0;
    for (int cls = 0; cls < ct.ncls; ++cls) {
      const int ph = cls / st, pw = cls % st;
      Taps& t = ct.taps[cls];
      int nt = 0;
      for (int kh = 0; kh < d->KH; ++kh) {
        if (((ph + g.pad_t - kh) % st + st) % st != 0) continue;
        for (int kw = 0; kw < d->KW; ++kw) {
          if (((pw + g.pad_l - kw) % st + st) % st != 0) continue;
          // oh = (ih + pad - kh) / stride, ih = a*st + ph
          const int nh = ph + g.pad_t - kh, nw = pw + g.pad_l - kw;
          t.dhw[nt] = pack_dhw(nh >= 0 ? nh / st : -((-nh) / st), nw >= 0 ? nw / st : -((-nw) / st));
          t.boff[nt] = (kh * d->KW + kw) * g.Ceff * d->Cout;
          ++nt;
        }
      }
      t.n = nt;
      ct.oa[cls] = ph;
      ct.ob[cls] = pw;
      ct.zbase[cls] = z;
      z += nt;
      flops += 2.0 * ga.Mtot * (double)nt * d->Cout * g.Ceff;
    }
    ct.zbase[ct.ncls] = z;
    e.out = target; e.ldo = tld; e.coff = 0;
    e.so = st; e.OHf = g.Hin; e.OWf = g.Win;
    e.logUpX = g.logUp;
    e.accumulate = tacc;
  }
  if (!pool && d->dx_amax_out && igemm_dgrad_amax_ok(d, lddx)) {
    e.amax_out = d->dx_amax_out;
    g_amax_written = true;
  }
  {
    ProfScope ps(OTGAN_PROF_CONV_DGRAD, flops, 0.0, s);
    if (vec && !g.fold) {
      const size_t used = pool ? sizeof(float) * (size_t)d->N * g.Hin * g.Win * d->C : 0;
      const long welems = (long)d->KH * d->KW * g.Ceff * d->Cout;
      if (welems < (1L << 31))
        igemm_records(e, d->dy_amax, dy + d->y_coff, (long)d->N * g.OH * g.OW, d->Cout, d->ldy, d->w_amax, w, welems, 0, workspace,
                      workspace_bytes, used, s);
    }
    if (paired) launch_igemm<EPI_DG_PAIR, 0>(vec, d->Cout, ga.Mtot, d->C, true, ct.ncls, s, ga, ct, wb, e);
    else if (kind) launch_igemm<EPI_DG_ACT, 0>(vec, d->Cout, ga.Mtot, d->C, false, ct.ncls, s, ga, ct, wb, e);
    else launch_igemm<EPI_DG_PLAIN, 0>(vec, d->Cout, ga.Mtot, d->C, false, ct.ncls, s, ga, ct, wb, e);
  }
  OTGAN_CHECK_LAUNCH("conv2d dgrad");
  if (pool) {
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
  OTGAN_CHECK_ARG(!p.vec || (aligned16(x) && aligned16(dy)), "operands must be 16-byte aligned");
  const size_t need = otgan_conv2d_workspace_bytes(d, 2);
  if ((p.nsplit > 1 || p.fold) && (!workspace || workspace_bytes < need)) {
    otgan_set_error("conv2d wgrad workspace too small: need %zu, got %zu", need, workspace_bytes);
    return OTGAN_ERR_WORKSPACE;
  }
  if (wino_up3_wgrad_ok(d, g) && cmap == nullptr && aligned16(x) && aligned16(dy) && aligned16(dw) && aligned16(workspace) &&
      workspace && workspace_bytes >= otgan_conv2d_workspace_bytes(d, 2)) {
    WinoS2Geo wg = wino_up3_wgrad_geo(d, g);
    wg.x_op = shared_x_operand(d);
    ProfScope ps(OTGAN_PROF_CONV_WGRAD, 2.0 * kWinoFreq * (double)wino_s2_tiles(wg) * g.Ceff * d->Cout, 0.0, s);
    rc = WINO(wino_s2_wgrad)(wg, x, dy, dw, (float*)workspace, s);
    OTGAN_CHECK_LAUNCH("conv2d wgrad (winograd, 3x3 on upsampled input)");
    return rc;
  }
  if (wino_s2_ok(d, g) && cmap == nullptr && aligned16(x) && aligned16(dy) && aligned16(dw) && aligned16(workspace) &&
      workspace && workspace_bytes >= otgan_conv2d_workspace_bytes(d, 2)) {
    WinoS2Geo wg = wino_s2_geo(d, g);
    wg.x_op = shared_x_operand(d);
    ProfScope ps(OTGAN_PROF_CONV_WGRAD, 2.0 * wino_s2_blocks(wg) * (double)wino_s2_tiles(wg) * g.Ceff * d->Cout, 0.0, s);
    rc = WINO(wino_s2_wgrad)(wg, x, dy, dw, (float*)workspace, s);
    OTGAN_CHECK_LAUNCH("conv2d wgrad (winograd, stride 2)");
    return rc;
  }
  OTGAN_CHECK_ARG(wino_ok(d, g) || shared_x_operand(d) == nullptr,
                  "x_operand given, but this weight-gradient call cannot take the Winograd path that reads it (list input, "
                  "alignment, workspace)");
  if (wino_ok(d, g)) {
    OTGAN_CHECK_ARG(aligned16(x) && aligned16(dy) && aligned16(dw) && aligned16(workspace),
                    "winograd wgrad needs 16-byte aligned operands");
    const size_t need = otgan_conv2d_workspace_bytes(d, 2);
    if (!workspace || workspace_bytes < need) {
      otgan_set_error("conv2d wgrad workspace too small: need %zu, got %zu", need, workspace_bytes);
      return OTGAN_ERR_WORKSPACE;
    }
    const FoldTab f = make_fold(d, g);
    WinoGeo wg = wino_geo(d);
    wg.x_op = shared_x_operand(d);
    float* ws = (float*)workspace;
    float* dweff = ws + WINO(wino_wgrad_ws_floats)(wg);
    {
      ProfScope ps(OTGAN_PROF_CONV_WGRAD, 2.0 * kWinoFreq * (double)wino_tiles(wg) * 4.0 * d->Cout * d->C, 0.0, s);
      // 5 x 5 'SAME' (the DCGAN generator): the adjoint filter transform writes the un-folded gradient itself (round 4;
      // OTGAN_WINO_UNFOLD_FUSED=0: the two kernels of round 3)
      static const bool fuse_off = getenv("OTGAN_WINO_UNFOLD_FUSED") && getenv("OTGAN_WINO_UNFOLD_FUSED")[0] == '0';
      const bool fuse = !fuse_off && d->KH == 5 && d->KW == 5 && g.pad_t == 2 && g.pad_l == 2;
      rc = WINO(wino_wgrad)(wg, x, dy, dweff, f.woff[1] - f.woff[0], ws, s, fuse ? dw : nullptr);
      if (rc) return rc;
      if (fuse) {
        OTGAN_CHECK_LAUNCH("conv2d wgrad (winograd, un-folded)");
        return OTGAN_OK;
      }
    }
    OTGAN_CHECK_LAUNCH("conv2d wgrad (winograd)");
    const long CkCout = (long)g.Ceff * d->Cout;
    const long total = (long)d->KH * d->KW * CkCout;
    long blocks = ceil_div_l(total, 256 * 4);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(unfold_wgrad_kernel, dim3((int)blocks), dim3(256), 0, s, (const float*)dweff, f, CkCout, dw);
    OTGAN_CHECK_LAUNCH("unfold_wgrad");
    return OTGAN_OK;
  }
  if (p.dense16) {
    OTGAN_CHECK_ARG(aligned16(x) && aligned16(dy) && aligned16(cmap), "operands must be 16-byte aligned");
    Dense16Geo dg;
    dg.N = d->N; dg.H = d->H; dg.W = d->W; dg.logH = ilog2_exact(d->H); dg.logW = ilog2_exact(d->W);
    dg.C = d->C; dg.Ceff = g.Ceff; dg.doubled = doubled_act(d->preact) ? 1 : 0;
    dg.act = act_kind(d->preact); dg.ldx = d->ldx; dg.cmap = cmap;
    float* slabs = p.nsplit > 1 ? (float*)workspace : dw;
    {
      ProfScope ps(OTGAN_PROF_CONV_WGRAD, 2.0 * (double)p.M * (double)p.slab_elems, 0.0, s);
      // round 4: with amax records of the x slices and of dy the kernel runs on the fp16 matrix pipe
      const bool recs = d->x_amax && d->dy_amax;
      rc = dense16_wgrad(dg, x, dy, d->ldy, d->y_coff, slabs, s, recs ? d->x_amax : nullptr, d->x_amax_count > 1 ? d->x_amax_count : 1,
                         recs ? d->dy_amax : nullptr, d->dy_amax_count > 1 ? d->dy_amax_count : 1);
      if (rc) return rc;
    }
    OTGAN_CHECK_LAUNCH("conv2d wgrad (dense16)");
    if (p.nsplit > 1) {
      launch_slab_reduce((const float*)slabs, p.nsplit, p.slab_elems, dw, s);
      OTGAN_CHECK_LAUNCH("slab_reduce");
    }
    return OTGAN_OK;
  }
  GatherA ga;
  ClassTab ct;
  if (p.outer) {
    single_class(d, g, &ct);
    OuterArgs oa;
    memset(&oa, 0, sizeof(oa));
    oa.H = d->H; oa.W = d->W;
    oa.logGH = ilog2_exact(g.OH); oa.logGW = ilog2_exact(g.OW);
    oa.Mtot = (int)p.M; oa.sa = d->stride;
    oa.cmap = cmap; oa.Creal = d->C; oa.doubled = doubled_act(d->preact) ? 1 : 0;
    oa.chunk = p.chunk;
    oa.slab = (float*)workspace; oa.slab_stride = p.slab_elems;
    oa.sT = (long)g.Ceff * d->Cout;
    if (p.outer == 1) {   // few outputs: wide = act(x) shifted by the tap, narrow = dy
      oa.wide = x; oa.ldw = d->ldx; oa.wideC = g.Ceff; oa.wide_shift = 1;
      oa.narrow = dy + d->y_coff; oa.ldn = d->ldy; oa.J = d->Cout;
      oa.sC = d->Cout; oa.sJ = 1;
    } else {              // few inputs: wide = dy, narrow = x shifted by the tap
      oa.wide = dy + d->y_coff; oa.ldw = d->ldy; oa.wideC = d->Cout; oa.wide_shift = 0;
      oa.narrow = x; oa.ldn = d->ldx; oa.J = d->C;
      oa.sC = 1; oa.sJ = d->Cout;
    }
    const dim3 grid(ceil_div(oa.wideC, 128), p.nchunks, ct.taps[0].n);
    const int act = act_kind(d->preact);
    if (d->stride == 1 && (d->KW == 5 || d->KW == 3) && d->KH <= 5 && g.pad_t <= kOuterHalo && g.pad_l <= kOuterHalo &&
        d->KH - 1 - g.pad_t <= kOuterHalo && d->KW - 1 - g.pad_l <= kOuterHalo && d->H % outer_unit_rows(d->H) == 0 &&
        oa.ldw % 4 == 0 && aligned16(oa.wide) && (oa.wideC % 4 == 0)) {
      Outer2Args o2;
      memset(&o2, 0, sizeof(o2));
      o2.wide = oa.wide; o2.ldw = oa.ldw; o2.wideC = oa.wideC;
      o2.narrow = oa.narrow; o2.ldn = oa.ldn; o2.J = oa.J;
      o2.wide_is_x = oa.wide_shift; o2.sgn = oa.wide_shift ? -1 : 1;
      o2.N = d->N; o2.H = d->H; o2.W = d->W; o2.logW = ilog2_exact(d->W);
      o2.KH = d->KH; o2.KW = d->KW; o2.ph = g.pad_t; o2.pw = g.pad_l;
      o2.cmap = cmap; o2.Creal = d->C; o2.doubled = oa.doubled;
      o2.TRo = outer_unit_rows(d->H);
      o2.units = d->N * (d->H / o2.TRo);
      o2.units_per_block = p.chunk / (o2.TRo * d->W);
      o2.slab = oa.slab; o2.slab_stride = oa.slab_stride; o2.sT = oa.sT; o2.sC = oa.sC; o2.sJ = oa.sJ;
      const dim3 g2(ceil_div(oa.wideC, 128), p.nchunks), blk(32 * d->KH * 2);
      size_t lds = sizeof(float4) * (o2.TRo + 2 * kOuterHalo) * (d->W + 2 * kOuterHalo);
      const size_t red = sizeof(float) * (size_t)d->KH * 32 * d->KW * 16;
      if (red > lds) lds = red;
      if (d->KH == d->KW && d->KW * oa.J <= 16 && d->W % 4 == 0 && (o2.TRo * d->W) % 16 == 0) {
        // fp32 matrix pipe: 2 channel groups x 2 pixel streams per workgroup
        size_t l2 = sizeof(float4) * (o2.TRo + 2 * kOuterHalo) * (d->W + 2 * kOuterHalo);
        const size_t r2 = sizeof(float4) * 2 * (size_t)d->KH * 4 * 64;
        if (r2 > l2) l2 = r2;
        ProfScope ps(OTGAN_PROF_CONV_WGRAD, 2.0 * (double)p.M * (double)p.slab_elems, 0.0, s);
#define OTGAN_OUTERM(ACT_)                                                                              \
  do {                                                                                                  \
    if (d->KW == 5) hipLaunchKernelGGL((conv_outer_mfma_kernel<ACT_, 5>), g2, dim3(256), l2, s, o2);    \
    else hipLaunchKernelGGL((conv_outer_mfma_kernel<ACT_, 3>), g2, dim3(256), l2, s, o2);               \
  } while (0)
        if (act == 1) OTGAN_OUTERM(1);
        else if (act == 2) OTGAN_OUTERM(2);
        else OTGAN_OUTERM(0);
#undef OTGAN_OUTERM
      } else {
        ProfScope ps(OTGAN_PROF_CONV_WGRAD, 2.0 * (double)p.M * (double)p.slab_elems, 0.0, s);
#define OTGAN_OUTER2(ACT_)                                                                              \
  do {                                                                                                  \
    if (d->KW == 5) hipLaunchKernelGGL((conv_outer2_kernel<ACT_, 5>), g2, blk, lds, s, o2);             \
    else hipLaunchKernelGGL((conv_outer2_kernel<ACT_, 3>), g2, blk, lds, s, o2);                        \
  } while (0)
        if (act == 1) OTGAN_OUTER2(1);
        else if (act == 2) OTGAN_OUTER2(2);
        else OTGAN_OUTER2(0);
#undef OTGAN_OUTER2
      }
      OTGAN_CHECK_LAUNCH("conv2d wgrad (few channels, all taps)");
    } else {
      ProfScope ps(OTGAN_PROF_CONV_WGRAD, 2.0 * (double)p.M * (double)p.slab_elems, 0.0, s);
      if (act == 1) hipLaunchKernelGGL(conv_outer_kernel<1>, grid, dim3(256), 0, s, oa, ct.taps[0]);
      else if (act == 2) hipLaunchKernelGGL(conv_outer_kernel<2>, grid, dim3(256), 0, s, oa, ct.taps[0]);
      else hipLaunchKernelGGL(conv_outer_kernel<0>, grid, dim3(256), 0, s, oa, ct.taps[0]);
      OTGAN_CHECK_LAUNCH("conv2d wgrad (few channels)");
    }
    launch_slab_reduce((const float*)workspace, p.nchunks, p.slab_elems, dw, s);
    OTGAN_CHECK_LAUNCH("slab_reduce");
    return OTGAN_OK;
  }
  WgradArgs a;
  memset(&a, 0, sizeof(a));
  a.dy = dy + d->y_coff;
  a.ldy = d->ldy;
  a.Cout = d->Cout;
  a.kt_per_split = p.kt_per_split;
  a.tiles_m = p.tiles_m;
  a.tiles_n = p.tiles_n;
  a.slab_stride = p.slab_elems;
  FoldTab f;
  memset(&f, 0, sizeof(f));
  float* slabs = (float*)workspace;
  float* dweff = nullptr;
  if (p.fold) {
    f = make_fold(d, g);
    fill_gather_x(d, g, x, cmap, d->H, d->W, 1, 0, &ga);
    folded_classes(f, (long)g.Ceff * d->Cout, f.woff, &ct);
    a.so = 2; a.OHf = g.OH; a.OWf = g.OW;
    dweff = p.nsplit > 1 ? slabs + (size_t)p.slab_elems * p.nsplit : slabs;
    a.slab = slabs;
  } else {
    fill_gather_x(d, g, x, cmap, g.OH, g.OW, d->stride, g.logUp, &ga);
    single_class(d, g, &ct);
    a.so = 1; a.OHf = g.OH; a.OWf = g.OW;
    a.slab = p.nsplit > 1 ? slabs : dw;
  }
  const int act = act_kind(d->preact);
  dim3 grid(p.tiles_m * p.tiles_n * p.nz, p.nsplit, 1);
  {
    ProfScope ps(OTGAN_PROF_CONV_WGRAD, 2.0 * (double)p.M * (double)p.slab_elems, 0.0, s);
    if (p.vec) {
      if (p.n16) launch_wgrad<CfgN16, true>(act, grid, s, ga, ct, a);
      else if (p.narrow) launch_wgrad<CfgNarrow, true>(act, grid, s, ga, ct, a);
      else if (p.wide == 160) launch_wgrad<CfgW160, true>(act, grid, s, ga, ct, a);
      else if (p.wide == 224) launch_wgrad<CfgW224, true>(act, grid, s, ga, ct, a);
      else launch_wgrad<CfgMain, true>(act, grid, s, ga, ct, a);
    } else {
      if (p.n16) launch_wgrad<CfgN16, false>(act, grid, s, ga, ct, a);
      else if (p.narrow) launch_wgrad<CfgNarrow, false>(act, grid, s, ga, ct, a);
      else launch_wgrad<CfgMain16, false>(act, grid, s, ga, ct, a);
    }
  }
  OTGAN_CHECK_LAUNCH("conv2d wgrad");
  if (p.nsplit > 1) {
    launch_slab_reduce((const float*)slabs, p.nsplit, p.slab_elems, p.fold ? dweff : dw, s);
    OTGAN_CHECK_LAUNCH("slab_reduce");
  }
  if (p.fold) {
    const long CkCout = (long)g.Ceff * d->Cout;
    const long total = (long)d->KH * d->KW * CkCout;
    long blocks = ceil_div_l(total, 256 * 4);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(unfold_wgrad_kernel, dim3((int)blocks), dim3(256), 0, s, (const float*)dweff, f,
                       CkCout, dw);
    OTGAN_CHECK_LAUNCH("unfold_wgrad");
  }
  return OTGAN_OK;
}

}  // extern "C"
