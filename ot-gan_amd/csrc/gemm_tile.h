// gemm_tile.h -- the fp32 MFMA block-GEMM engine shared by every contraction on the
// OT-GAN hot path (cost Gram blocks, plan application, implicit-GEMM conv fwd/dgrad/wgrad).
//
// Design (gfx950 / CDNA4, wave64):
//   * v_mfma_f32_32x32x2_f32: exact-fp32 matrix FMA (k-ordered fmaf chain), 64 FLOP/clk/SIMD,
//     157.3 TFLOP/s chip peak.  One instruction = 64 cycles on a SIMD, so a wave needs only
//     ONE dword of A and ONE dword of B per MFMA per lane; with a MTxNT register tile the LDS
//     read rate is (MT+NT)/(MT*NT) dwords per MFMA -- far below the LDS limit.  The engine is
//     therefore built for simplicity of the operand path, not for LDS bandwidth:
//       - operand tiles live in LDS as [BK][rows + pad] (row index contiguous),
//       - fragments are read with conflict-free ds_read_b32 (lane = row),
//       - global->LDS staging goes through registers so loaders can apply on-the-fly
//         transforms (CReLU / sign / zero padding / gathers) before the data reaches LDS,
//       - double-buffered LDS, one barrier per BK step; the loads for step k+1 are issued
//         before the MFMAs of step k and written to LDS after them (latency hidden under
//         BK/2 * MT*NT * 64 cycles of matrix work).
//   * MFMA fragment maps (32x32x2 f32):  A: lane l holds A[i=l&31][k=l>>5];
//     B: lane l holds B[k=l>>5][j=l&31];  D reg r of lane l is
//     D[row=(r&3)+8*(r>>2)+4*(l>>5)][col=l&31].
//
// A "loader" is a struct with:  static constexpr int LD, FLOATS;  void load(int kt);
// void store(float* lds_tile) const;   load() pulls tile kt into registers, store() writes
// it to an LDS tile laid out [BK][LD].
#pragma once
#include "common.h"

// TS = MFMA tile size: 32 (v_mfma_f32_32x32x2_f32, 16 accumulator registers) or
// 16 (v_mfma_f32_16x16x4_f32, 4 accumulator registers; for outputs only 16 columns wide).
template <int WM_, int WN_, int MT_, int NT_, int BK_, int TS_ = 32>
struct GemmCfg {
  static constexpr int WM = WM_, WN = WN_, MT = MT_, NT = NT_, BK = BK_, TS = TS_;
  static constexpr int THREADS = 64 * WM * WN;
  static constexpr int BM = WM * MT * TS;
  static constexpr int BN = WN * NT * TS;
  static constexpr int ACC = TS == 32 ? 16 : 4;           // accumulator registers per tile
  static constexpr int KSTEP = TS == 32 ? 2 : 4;          // k consumed per MFMA
  typedef float acc_t __attribute__((ext_vector_type(TS == 32 ? 16 : 4)));
  // accumulator register r of lane -> (row, col) inside the TS x TS tile
  __device__ static __forceinline__ int acc_row(int lane, int r) {
    return TS == 32 ? (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) : 4 * (lane >> 4) + r;
  }
  __device__ static __forceinline__ int acc_col(int lane) { return TS == 32 ? (lane & 31) : (lane & 15); }
};

// LDS row padding: chosen so that the scattered ds_write_b32 of a k-contiguous loader
// (lanes = {k-chunk, row}) hit distinct banks.  BK=32: 8 chunks -> pad 1; BK=16: 4 chunks -> pad 2.
template <int BK>
struct KPad {
  static constexpr int value = (BK == 32) ? 1 : 2;
};

template <class Cfg, class LA, class LB>
__device__ __forceinline__ void gemm_mainloop(LA& la, LB& lb, int nkt, float* smem,
                                              typename Cfg::acc_t (&acc)[Cfg::MT][Cfg::NT]) {
  constexpr int BK = Cfg::BK, MT = Cfg::MT, NT = Cfg::NT, TS = Cfg::TS, KS = Cfg::KSTEP;
  float* sA = smem;
  float* sB = smem + 2 * LA::FLOATS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
  const int li = lane & (TS - 1), lh = lane / TS;  // row/col inside the tile, k sub-index
  const int a_off = lh * LA::LD + wm * MT * TS + li;
  const int b_off = lh * LB::LD + wn * NT * TS + li;
  if (nkt <= 0) return;
  la.load(0);
  lb.load(0);
  la.store(sA);
  lb.store(sB);
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1;
    const bool more = (kt + 1 < nkt);
    if (more) {
      la.load(kt + 1);
      lb.load(kt + 1);
    }
    const float* pa = sA + cur * LA::FLOATS + a_off;
    const float* pb = sB + cur * LB::FLOATS + b_off;
    // fragment reads run one k-step ahead of the MFMAs that consume them
    float a[2][MT], b[2][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a[0][mt] = pa[mt * TS];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) b[0][nt] = pb[nt * TS];
#pragma unroll
    for (int ks = 0; ks < BK / KS; ++ks) {
      const int cb = ks & 1, nb = cb ^ 1;
      if (ks + 1 < BK / KS) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[nb][mt] = pa[(KS * ks + KS) * LA::LD + mt * TS];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b[nb][nt] = pb[(KS * ks + KS) * LB::LD + nt * TS];
        // keep the next step's LDS reads ahead of this step's MFMAs (hipcc otherwise sinks
        // them behind the MFMAs and waits lgkmcnt(0) right before the next group)
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          if constexpr (TS == 32)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cb][mt], b[cb][nt], acc[mt][nt], 0, 0, 0);
          else
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cb][mt], b[cb][nt], acc[mt][nt], 0, 0, 0);
        }
    }
    if (more) {
      la.store(sA + (cur ^ 1) * LA::FLOATS);
      lb.store(sB + (cur ^ 1) * LB::FLOATS);
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------
// Split-precision variant of the main loop (32x32 tiles, BK = 16): the loaders split every element ONCE, while
// staging it, into three bf16 pieces x = hi + mid + lo (8 + 8 + 8 significand bits: exact) and write three
// k-contiguous LDS planes [rows][16 k] (32 bytes a row); a fragment (row = lane % 32, k = 8 * (lane / 32) .. +7) is
// one ds_read_b128 per plane, and a 16-wide k slab costs six v_mfma_f32_32x32x16_bf16 (the products of weight
// >= 2^-16; gemm_x3.h) = 192 matrix-pipe cycles per tile instead of 8 x 64 = 512 on the fp32 instructions.
// The two 16-byte halves of a row are swapped on every other group of four rows, which makes both the b128
// fragment reads (8 lanes per phase) and the 8-byte staging writes conflict-free without padding:
// LDS = 2 buffers x 3 planes x (BM + BN) x 32 B = 48 KB for a 128 x 128 tile (three workgroups per CU).
// A loader provides store3(unsigned char* tile) next to store().
// ---------------------------------------------------------------------------------------
typedef __bf16 gt_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 gt_bf16x2 __attribute__((ext_vector_type(2)));
typedef float gt_f32x2 __attribute__((ext_vector_type(2)));
typedef float gt_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int gt_u32x2 __attribute__((ext_vector_type(2)));

constexpr int kX3sRowBytes = 32;
// byte offset of k (a multiple of 4) of row r inside a plane
__device__ __forceinline__ int x3s_off(int r, int k) {
  return r * kX3sRowBytes + ((((k >> 3) ^ (r >> 2)) & 1) << 4) + ((k & 4) << 1);
}
// four consecutive k of one row -> the three planes (plane stride `plane` bytes)
__device__ __forceinline__ void x3s_store4(unsigned char* tile, int plane, int r, int k, float4 v) {
  gt_f32x2 a = {v.x, v.y}, b = {v.z, v.w};
  const gt_bf16x2 ha = __builtin_convertvector(a, gt_bf16x2), hb = __builtin_convertvector(b, gt_bf16x2);
  a -= __builtin_convertvector(ha, gt_f32x2);
  b -= __builtin_convertvector(hb, gt_f32x2);
  const gt_bf16x2 ma = __builtin_convertvector(a, gt_bf16x2), mb = __builtin_convertvector(b, gt_bf16x2);
  a -= __builtin_convertvector(ma, gt_f32x2);
  b -= __builtin_convertvector(mb, gt_f32x2);
  const gt_bf16x2 la = __builtin_convertvector(a, gt_bf16x2), lb = __builtin_convertvector(b, gt_bf16x2);
  unsigned char* d = tile + x3s_off(r, k);
  *reinterpret_cast<gt_u32x2*>(d) = gt_u32x2{__builtin_bit_cast(unsigned, ha), __builtin_bit_cast(unsigned, hb)};
  *reinterpret_cast<gt_u32x2*>(d + plane) = gt_u32x2{__builtin_bit_cast(unsigned, ma), __builtin_bit_cast(unsigned, mb)};
  *reinterpret_cast<gt_u32x2*>(d + 2 * plane) = gt_u32x2{__builtin_bit_cast(unsigned, la), __builtin_bit_cast(unsigned, lb)};
}

template <class Cfg>
struct X3sLds {
  static constexpr int PA = Cfg::BM * kX3sRowBytes, PB = Cfg::BN * kX3sRowBytes;   // plane bytes
  static constexpr int TA = 3 * PA, TB = 3 * PB;                                   // one buffer of an operand
  static constexpr int BYTES = 2 * (TA + TB);
};

template <class Cfg, class LA, class LB>
__device__ __forceinline__ void gemm_mainloop_x3s(LA& la, LB& lb, int nkt, unsigned char* smem,
                                                  typename Cfg::acc_t (&acc)[Cfg::MT][Cfg::NT]) {
  constexpr int MT = Cfg::MT, NT = Cfg::NT, TS = Cfg::TS;
  static_assert(TS == 32 && Cfg::BK == 16, "split-precision main loop: 32x32 tiles, BK = 16");
  using L = X3sLds<Cfg>;
  unsigned char* sA = smem;
  unsigned char* sB = smem + 2 * L::TA;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
  const int li = lane & 31, lh = lane >> 5;
  int a_off[MT], b_off[NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) a_off[mt] = x3s_off((wm * MT + mt) * TS + li, 8 * lh);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) b_off[nt] = x3s_off((wn * NT + nt) * TS + li, 8 * lh);
  if (nkt <= 0) return;
  la.load(0);
  lb.load(0);
  la.store3(sA);
  lb.store3(sB);
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1;
    const bool more = (kt + 1 < nkt);
    if (more) {
      la.load(kt + 1);
      lb.load(kt + 1);
    }
    const unsigned char* pa = sA + cur * L::TA;
    const unsigned char* pb = sB + cur * L::TB;
    gt_bf16x8 fa[3][MT], fb[3][NT];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) fa[p][mt] = *reinterpret_cast<const gt_bf16x8*>(pa + p * L::PA + a_off[mt]);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) fb[p][nt] = *reinterpret_cast<const gt_bf16x8*>(pb + p * L::PB + b_off[nt]);
    }
    // lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi (smallest first); term-major: consecutive MFMAs never share an accumulator
#pragma unroll
    for (int term = 0; term < 6; ++term) {
      constexpr int pa_of[6] = {2, 0, 1, 1, 0, 0}, pb_of[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[pa_of[term]][mt], fb[pb_of[term]][nt], acc[mt][nt], 0, 0, 0);
    }
    if (more) {
      la.store3(sA + (cur ^ 1) * L::TA);
      lb.store3(sB + (cur ^ 1) * L::TB);
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------
// Two scaled fp16 pieces (round 3): x * 2^s = hi + lo, 22 significand bits, THREE v_mfma_f32_32x32x16_f16 per 16-wide
// k slab (hi*hi, hi*lo, lo*hi) instead of six on three bf16 pieces -- the arithmetic of the Winograd-domain GEMMs
// (gemm_x3.h) in the implicit-GEMM main loop.  One power-of-two scale per OPERAND TENSOR, from its largest magnitude
// (an amax record, common.h): the loaders multiply while staging (`xs`), the epilogue multiplies the sums by the inverse
// product (exact).  Same LDS layout as above with two planes: 32 KB for a 128 x 128 tile.
// A loader provides store2(unsigned char* tile) and a public `float xs`.
// ---------------------------------------------------------------------------------------
typedef _Float16 gt_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 gt_f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void x2h_store4(unsigned char* tile, int plane, int r, int k, float4 v) {
  gt_f32x2 a = {v.x, v.y}, b = {v.z, v.w};
  const gt_f16x2 ha = __builtin_convertvector(a, gt_f16x2), hb = __builtin_convertvector(b, gt_f16x2);
  a -= __builtin_convertvector(ha, gt_f32x2);
  b -= __builtin_convertvector(hb, gt_f32x2);
  const gt_f16x2 la = __builtin_convertvector(a, gt_f16x2), lb = __builtin_convertvector(b, gt_f16x2);
  unsigned char* d = tile + x3s_off(r, k);
  *reinterpret_cast<gt_u32x2*>(d) = gt_u32x2{__builtin_bit_cast(unsigned, ha), __builtin_bit_cast(unsigned, hb)};
  *reinterpret_cast<gt_u32x2*>(d + plane) = gt_u32x2{__builtin_bit_cast(unsigned, la), __builtin_bit_cast(unsigned, lb)};
}
// scale of a tensor with largest magnitude amax: |x| 2^(14 - e) < 2^14 with amax = m 2^e, 0.5 <= m < 1; *e_out = e.
// NaN / infinite amax: NaN scale (the result stays loud); all-zero tensor: e = 0.
__device__ __forceinline__ float x2h_scale(float amax, int* e_out) {
  int e = 0;
  if (amax > 0.f) e = __builtin_amdgcn_frexp_expf(amax);
  *e_out = e;
  if (!(amax <= 3.0e38f)) return __builtin_nanf("");
  return __builtin_ldexpf(1.f, 14 - e);
}

template <class Cfg>
struct X2hLds {
  static constexpr int PA = Cfg::BM * kX3sRowBytes, PB = Cfg::BN * kX3sRowBytes;   // plane bytes
  static constexpr int TA = 2 * PA, TB = 2 * PB;                                   // one buffer of an operand
  static constexpr int BYTES = 2 * (TA + TB);
};

template <class Cfg, class LA, class LB>
__device__ __forceinline__ void gemm_mainloop_x2h(LA& la, LB& lb, int nkt, unsigned char* smem,
                                                  typename Cfg::acc_t (&acc)[Cfg::MT][Cfg::NT]) {
  constexpr int MT = Cfg::MT, NT = Cfg::NT, TS = Cfg::TS;
  static_assert(TS == 32 && Cfg::BK == 16, "split-precision main loop: 32x32 tiles, BK = 16");
  using L = X2hLds<Cfg>;
  unsigned char* sA = smem;
  unsigned char* sB = smem + 2 * L::TA;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
  const int li = lane & 31, lh = lane >> 5;
  int a_off[MT], b_off[NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) a_off[mt] = x3s_off((wm * MT + mt) * TS + li, 8 * lh);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) b_off[nt] = x3s_off((wn * NT + nt) * TS + li, 8 * lh);
  if (nkt <= 0) return;
  la.load(0);
  lb.load(0);
  la.store2(sA);
  lb.store2(sB);
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1;
    const bool more = (kt + 1 < nkt);
    if (more) {
      la.load(kt + 1);
      lb.load(kt + 1);
    }
    const unsigned char* pa = sA + cur * L::TA;
    const unsigned char* pb = sB + cur * L::TB;
    gt_f16x8 fa[2][MT], fb[2][NT];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) fa[p][mt] = *reinterpret_cast<const gt_f16x8*>(pa + p * L::PA + a_off[mt]);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) fb[p][nt] = *reinterpret_cast<const gt_f16x8*>(pb + p * L::PB + b_off[nt]);
    }
    // lo*hi, hi*lo, hi*hi (smallest first); term-major: consecutive MFMAs never share an accumulator
#pragma unroll
    for (int term = 0; term < 3; ++term) {
      constexpr int pa_of[3] = {1, 0, 0}, pb_of[3] = {0, 1, 0};
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[pa_of[term]][mt], fb[pb_of[term]][nt], acc[mt][nt], 0, 0, 0);
    }
    if (more) {
      la.store2(sA + (cur ^ 1) * L::TA);
      lb.store2(sB + (cur ^ 1) * L::TB);
    }
    __syncthreads();
  }
}

// The same loop with the global loads TWO k steps ahead (round 5, late).  The one-deep loop above runs one k step per
// global-load round trip: with two workgroups per compute unit a stride-2 transition of the DenseNet critic spends 1.65 us
// per k step for 0.23 us of matrix work.  Here a loader keeps its in-flight data in caller-owned STAGES (two per operand):
//     load_s(Stage&, live)          issue the loads of the next k tile (branch-free: clamped addresses; `live` false past
//                                   the end of the K range: the stage converts to zeros)
//     store2_s(tile, const Stage&)  convert and store (masked elements as zeros)
//     pin_s(Stage&)                 an empty asm that "rewrites" the stage's registers: hipcc otherwise hoists the conversion
//                                   (pure arithmetic: no chain to a scheduling barrier) to the top of the trip, where it waits
//                                   for loads issued a moment ago
// and nothing in the loop may wait on a load issued in the same half trip (a loader that prefetches an index through
// global memory -- the channel map of the list inputs -- reads it from LDS instead).
template <class Cfg, class LA, class LB>
__device__ __forceinline__ void gemm_mainloop_x2h_d2(LA& la, LB& lb, int nkt, unsigned char* smem,
                                                     typename Cfg::acc_t (&acc)[Cfg::MT][Cfg::NT]) {
  constexpr int MT = Cfg::MT, NT = Cfg::NT, TS = Cfg::TS;
  static_assert(TS == 32 && Cfg::BK == 16, "split-precision main loop: 32x32 tiles, BK = 16");
  using L = X2hLds<Cfg>;
  unsigned char* sA = smem;
  unsigned char* sB = smem + 2 * L::TA;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
  const int li = lane & 31, lh = lane >> 5;
  int a_off[MT], b_off[NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) a_off[mt] = x3s_off((wm * MT + mt) * TS + li, 8 * lh);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) b_off[nt] = x3s_off((wn * NT + nt) * TS + li, 8 * lh);
  if (nkt <= 0) return;
  auto mma = [&](int cur) {
    const unsigned char* pa = sA + cur * L::TA;
    const unsigned char* pb = sB + cur * L::TB;
    gt_f16x8 fa[2][MT], fb[2][NT];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) fa[p][mt] = *reinterpret_cast<const gt_f16x8*>(pa + p * L::PA + a_off[mt]);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) fb[p][nt] = *reinterpret_cast<const gt_f16x8*>(pb + p * L::PB + b_off[nt]);
    }
#pragma unroll
    for (int term = 0; term < 3; ++term) {
      constexpr int pa_of[3] = {1, 0, 0}, pb_of[3] = {0, 1, 0};
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[pa_of[term]][mt], fb[pb_of[term]][nt], acc[mt][nt], 0, 0, 0);
    }
  };
  typename LA::Stage a0, a1;
  typename LB::Stage b0, b1;
  la.load_s(a0, true); lb.load_s(b0, true);
  la.load_s(a1, 1 < nkt); lb.load_s(b1, 1 < nkt);
  la.store2_s(sA, a0);
  lb.store2_s(sB, b0);
  __syncthreads();
  for (int kt = 0; kt < nkt; kt += 2) {
    la.load_s(a0, kt + 2 < nkt); lb.load_s(b0, kt + 2 < nkt);
    __builtin_amdgcn_sched_barrier(0);
    mma(0);
    __builtin_amdgcn_sched_barrier(0);
    LA::pin_s(a1); LB::pin_s(b1);
    la.store2_s(sA + L::TA, a1);
    lb.store2_s(sB + L::TB, b1);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    la.load_s(a1, kt + 3 < nkt); lb.load_s(b1, kt + 3 < nkt);
    __builtin_amdgcn_sched_barrier(0);
    mma(1);
    __builtin_amdgcn_sched_barrier(0);
    LA::pin_s(a0); LB::pin_s(b0);
    la.store2_s(sA, a0);
    lb.store2_s(sB, b0);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <class Cfg>
__device__ __forceinline__ void zero_acc(typename Cfg::acc_t (&acc)[Cfg::MT][Cfg::NT]) {
#pragma unroll
  for (int mt = 0; mt < Cfg::MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < Cfg::NT; ++nt)
#pragma unroll
      for (int r = 0; r < Cfg::ACC; ++r) acc[mt][nt][r] = 0.f;
}

// Visit every accumulator element: f(row_in_block, col_in_block, mt, nt, r, value).
// For a fixed (mt, nt, r) the lanes of a (half-)wave cover consecutive columns of one row, so a
// store of `value` to out[row*ld + col] is coalesced.
template <class Cfg, class F>
__device__ __forceinline__ void foreach_acc(const typename Cfg::acc_t (&acc)[Cfg::MT][Cfg::NT], F&& f) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
#pragma unroll
  for (int mt = 0; mt < Cfg::MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < Cfg::NT; ++nt)
#pragma unroll
      for (int r = 0; r < Cfg::ACC; ++r) {
        const int row = (wm * Cfg::MT + mt) * Cfg::TS + Cfg::acc_row(lane, r);
        const int col = (wn * Cfg::NT + nt) * Cfg::TS + Cfg::acc_col(lane);
        f(row, col, mt, nt, r, acc[mt][nt][r]);
      }
}

// ---------------------------------------------------------------------------------------
// Plain-matrix loaders.
// ---------------------------------------------------------------------------------------

// Operand whose rows are K-contiguous in memory: element (r, k) at base[r*ld + k].
// Each thread fetches float4s along k (coalesced 16*BK/4 bytes per row) and scatters them
// into the [BK][LD] LDS tile.  VEC=false is the scalar fallback for unaligned / ragged K.
template <class Cfg, int BR, bool VEC>
struct MatLoaderK {
  static constexpr int BK = Cfg::BK;
  static constexpr int LD = BR + KPad<BK>::value;
  static constexpr int FLOATS = BK * LD;
  static constexpr int CPR = BK / 4;                // float4 chunks per row
  static constexpr int RPP = Cfg::THREADS / CPR;    // rows per pass
  static constexpr int PASSES = (BR + RPP - 1) / RPP;
  const float* base;  // already offset to (row0, k0)
  long ld;
  int rows;  // valid rows (relative to row0)
  int kdim;  // valid k (relative to k0)
  float scale;
  float4 reg[PASSES];

  __device__ __forceinline__ void init(const float* b, long ld_, int rows_, int kdim_, float s = 1.f) {
    base = b; ld = ld_; rows = rows_; kdim = kdim_; scale = s;
  }
  __device__ __forceinline__ void load(int kt) {
    const int c = threadIdx.x % CPR, r0 = threadIdx.x / CPR;
    const int k = kt * BK + 4 * c;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int r = r0 + p * RPP;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < BR && r < rows) {
        const float* src = base + (long)r * ld + k;
        if (VEC) {
          if (k + 3 < kdim) v = *reinterpret_cast<const float4*>(src);
          else {
            if (k + 0 < kdim) v.x = src[0];
            if (k + 1 < kdim) v.y = src[1];
            if (k + 2 < kdim) v.z = src[2];
          }
        } else {
          if (k + 0 < kdim) v.x = src[0];
          if (k + 1 < kdim) v.y = src[1];
          if (k + 2 < kdim) v.z = src[2];
          if (k + 3 < kdim) v.w = src[3];
        }
      }
      reg[p] = v;
    }
  }
  __device__ __forceinline__ void store(float* t) const {
    const int c = threadIdx.x % CPR, r0 = threadIdx.x / CPR;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int r = r0 + p * RPP;
      if (r < BR) {
        float* d = t + (4 * c) * LD + r;
        d[0] = reg[p].x * scale;
        d[LD] = reg[p].y * scale;
        d[2 * LD] = reg[p].z * scale;
        d[3 * LD] = reg[p].w * scale;
      }
    }
  }
};

// Operand whose ROW index is contiguous in memory: element (r, k) at base[k*ld + r].
// float4 along r; written to LDS with one ds_write_b128 per chunk.
template <class Cfg, int BR, bool VEC>
struct MatLoaderR {
  static constexpr int BK = Cfg::BK;
  static constexpr int LD = BR + 4;
  static constexpr int FLOATS = BK * LD;
  static constexpr int CPK = BR / 4;                 // float4 chunks per k
  static constexpr int KPP = (Cfg::THREADS / CPK) > 0 ? (Cfg::THREADS / CPK) : 1;  // k per pass
  static constexpr int PASSES = (BK + KPP - 1) / KPP;
  const float* base;  // offset to (row0, k0)
  long ld;
  int rows, kdim;
  float4 reg[PASSES];

  __device__ __forceinline__ void init(const float* b, long ld_, int rows_, int kdim_) {
    base = b; ld = ld_; rows = rows_; kdim = kdim_;
  }
  __device__ __forceinline__ void load(int kt) {
    const int c = threadIdx.x % CPK, k0 = threadIdx.x / CPK;
    const int r = 4 * c;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int kk = k0 + p * KPP;
      const int k = kt * BK + kk;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (kk < BK && k < kdim && threadIdx.x < CPK * KPP) {
        const float* src = base + (long)k * ld + r;
        if (VEC) {
          if (r + 3 < rows) v = *reinterpret_cast<const float4*>(src);
          else {
            if (r + 0 < rows) v.x = src[0];
            if (r + 1 < rows) v.y = src[1];
            if (r + 2 < rows) v.z = src[2];
          }
        } else {
          if (r + 0 < rows) v.x = src[0];
          if (r + 1 < rows) v.y = src[1];
          if (r + 2 < rows) v.z = src[2];
          if (r + 3 < rows) v.w = src[3];
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
      if (kk < BK && threadIdx.x < CPK * KPP)
        *reinterpret_cast<float4*>(t + kk * LD + 4 * c) = reg[p];
    }
  }
};
