// gemm_x3.h -- the batched NT / t-leading GEMM on the bf16 matrix pipe with split-precision operands
// (three bf16 planes per fp32 operand, six v_mfma_f32_32x32x16_bf16 per fp32-exact product).
// Shared by the Winograd-domain convolution GEMMs (winograd.hip) and the matching GEMMs -- cost Gram
// blocks and plan application (sinkhorn.hip).  Everything here has internal linkage: each translation
// unit instantiates its own copy of the kernel.
#pragma once
#include <type_traits>

#include "common.h"

namespace {

// Winograd F(m x m, 3 x 3): transformed tile edge kWA = m + 2, kWF = kWA^2 frequencies (batched GEMMs).
#ifndef OTGAN_WINO_ALPHA
#define OTGAN_WINO_ALPHA 6
#endif
constexpr int kWA = OTGAN_WINO_ALPHA;
constexpr int kWF = kWA * kWA;

// Strided 5x5 layers: is the (parity class, frequency) block structurally non-zero?  The even-parity classes have
// a zero outer tap, which empties frequency index `skip` (0: forward orientation, kWA - 1: flipped filters) in
// that dimension.
__host__ __device__ __forceinline__ bool s2_present(int cls, int f, int skip) {
  const int pi = cls >> 1, pj = cls & 1, fi = f / kWA, fj = f % kWA;
  return (pi || fi != skip) && (pj || fj != skip);
}

// ---- split-precision operands -----------------------------------------------------------------
// The NT GEMMs (forward / dgrad of both layer families) run on the bf16 matrix pipe with
// fp32-exact products: every fp32 operand is stored as three bf16 planes x = hi + mid + lo
// (8 + 8 + 8 mantissa bits) by the kernel that produces it, and the GEMM issues the six MFMAs
// hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid per k slab, accumulating in fp32.  Measured
// 2.3e-7 .. 5e-7 relative L2 against fp64 at K = 256 .. 1024 on plain GEMMs (the fp32 MFMA chain: 1.3e-6).
typedef unsigned short u16;
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// Pieces per operand element (per translation unit): 3 = bf16 hi + mid + lo, 24 significand bits, six MFMAs per
// product; 2 = fp16 hi + lo of the power-of-two SCALED value, 22 bits, three MFMAs (hi*hi, hi*lo, lo*hi).
#ifndef X3_PIECES
#define X3_PIECES 3
#endif
constexpr int X3_NP = X3_PIECES;
constexpr int X3_NTERM = X3_NP == 3 ? 6 : 3;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#if X3_PIECES == 3
typedef bf16x8 x3frag_t;
#define X3_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
constexpr int X3_PA[6] = {2, 0, 1, 1, 0, 0}, X3_PB[6] = {0, 2, 1, 0, 1, 0};   // smallest terms first
#else
typedef f16x8 x3frag_t;
#define X3_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
constexpr int X3_PA[3] = {1, 0, 0}, X3_PB[3] = {0, 1, 0};
#endif

// R fragment reads spread over the 16 MFMAs of a term group (scheduling hint; the builtin wants literals)
template <int DS, int R, int Q = 0>
__device__ __forceinline__ void x3_sched_group() {
  if constexpr (Q < R) {
    __builtin_amdgcn_sched_group_barrier(0x100, DS, 0);                                  // the DS reads of one fragment
    __builtin_amdgcn_sched_group_barrier(0x008, (Q + 1) * 16 / R - Q * 16 / R, 0);       // its share of the MFMAs
    x3_sched_group<DS, R, Q + 1>();
  }
}
// calls f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>)
template <int N, int I = 0, class F>
__device__ __forceinline__ void x3_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    x3_static_for<N, I + 1>(f);
  }
}

// (hi, mid, lo) bf16 pieces of two floats, each packed in one dword (round-to-nearest-even through
// v_cvt_pk_bf16_f32; the residuals are exact in fp32)
__device__ __forceinline__ void split2(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
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
// planes[p][idx .. idx+3] = p-th bf16 piece of v
__device__ __forceinline__ void st_split4(u16* planes, long plane_stride, long idx, f32x4 v) {
  unsigned h0, m0, l0, h1, m1, l1;
  split2(v[0], v[1], h0, m0, l0);
  split2(v[2], v[3], h1, m1, l1);
  *reinterpret_cast<u32x2*>(planes + idx) = u32x2{h0, h1};
  *reinterpret_cast<u32x2*>(planes + plane_stride + idx) = u32x2{m0, m1};
  *reinterpret_cast<u32x2*>(planes + 2 * plane_stride + idx) = u32x2{l0, l1};
}
// ---- two scaled fp16 pieces (X3_PIECES == 2) ---------------------------------------------------------------
// x * 2^s = hi + lo with hi = fp16(x 2^s), lo = fp16(x 2^s - hi): 22 significand bits, and hi*hi + hi*lo + lo*hi is
// the product to 2^-22 -- measured 7.5e-8 relative L2 on dot products before fp32 accumulation (bf16 x 3: 6e-9; the
// accumulation itself: 3e-7), heavy-tailed and outlier-ridden operands included (oracle/split_precision_np.py).
// fp16 has 5 exponent bits, so every operand carries ONE power-of-two scale per frequency, derived from the largest
// magnitude of the tensor it is a transform of (amax, a device scalar in the operand's header) and the transform's
// gain bound (product of the absolute row sums of the two 1-D transform matrices): |value| <= gain * amax < 2^e,
// scale 2^(14 - e): the largest piece stays below 2^14, typical values 4 - 8 binades lower, their lo pieces normal
// down to 2^-3.  The GEMM multiplies the sums by 2^(eA + eB - 28) on the way out (exact).  The scales are computed
// once per operand (winograd.hip: absmax_kernel's last block) into a header of X3_HDR floats in front of the planes.
constexpr int X3_HDR = 128;
__host__ __device__ __forceinline__ int x3_scale_exp(float amax, float gi, float gj) {
  const float bound = amax * (gi * gj);
  if (!(bound > 0.f)) return 0;            // all-zero tensor (NaN falls through to the arithmetic and stays loud)
  int e;
#if defined(__HIP_DEVICE_COMPILE__)
  e = __builtin_amdgcn_frexp_expf(bound);  // bound = m 2^e, 0.5 <= m < 1
#else
  frexpf(bound, &e);
#endif
  return e;
}
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
// planes[p][idx .. idx+3] = p-th fp16 piece of v (already scaled)
__device__ __forceinline__ void st_split4h(u16* planes, long plane_stride, long idx, f32x4 v) {
  f32x2_t a = {v[0], v[1]}, b = {v[2], v[3]};
  const f16x2_t ha = __builtin_convertvector(a, f16x2_t), hb = __builtin_convertvector(b, f16x2_t);
  a -= __builtin_convertvector(ha, f32x2_t);
  b -= __builtin_convertvector(hb, f32x2_t);
  const f16x2_t la = __builtin_convertvector(a, f16x2_t), lb = __builtin_convertvector(b, f16x2_t);
  *reinterpret_cast<u32x2*>(planes + idx) = u32x2{__builtin_bit_cast(unsigned, ha), __builtin_bit_cast(unsigned, hb)};
  *reinterpret_cast<u32x2*>(planes + plane_stride + idx) = u32x2{__builtin_bit_cast(unsigned, la), __builtin_bit_cast(unsigned, lb)};
}
// Layout of a split-precision operand: per (piece, frequency) the [rows][K] matrix is stored as
// [row block of 32][k block of 16] chunks of 1 KiB, and a chunk is exactly the LDS image that one
// global_load_lds instruction of the GEMM deposits (64 lanes x 16 bytes): 16-byte slot
// (row % 32) * 2 + ((k / 8) % 2 ^ (row / 8) % 2) -- the XOR keeps the GEMM's ds_read_b128 fragment
// reads conflict-free with 32-byte rows.  A row-major operand made the GEMM's global reads 32-byte
// row segments (one cache line per lane pair); blocked, every instruction reads 8 full lines
// (measured on the DCGAN shapes: 141-156 -> 205-254 TFLOP/s fp32-equivalent, tools/ablate/gemm_bf16x3_v3.hip).
// Rows are padded to a multiple of 32 (padding never written, only feeds C rows that are not stored).
__host__ __device__ inline long op_fstride(long rows, long K) { return ((rows + 31) >> 5) * (K >> 4) * 512; }
__device__ __forceinline__ long op_off(long row, int k, int kblocks) {
  const int rr = (int)(row & 31);
  return (((row >> 5) * kblocks + (k >> 4)) << 9) + ((rr * 2 + (((k >> 3) & 1) ^ ((rr >> 3) & 1))) << 3) + (k & 7);
}

// ---- the batched GEMM ---------------------------------------------------------------------
// blockIdx.z = frequency f; blockIdx.y = K split; blockIdx.x -> (tm, tn) with the 8 XCDs taking
// different row tiles, so that the column tiles that share an A row tile share an L2.
//   TN = false: C[f] = A[f] (M x K, k contiguous) . B[f]^T (N x K, k contiguous)
//   TN = true : C[f] = A[f]^T (K x M, m contiguous) . B[f] (K x N, n contiguous)
struct BgArgs {
  const float* A;
  const float* B;
  float* C;
  int M, N, K;
  long lda, ldb, ldc;
  long sA, sB, sC, sSplit;
  int tiles_m, tiles_n, kt_per_split;
  int xmap;   // 1: XCD = row-tile residue, 2: XCD = column-tile residue, 0: linear, 4: frequency-major (fmap)
  // xmap 4 (x3 kernel, one-dimensional grid in x): workgroup x runs on XCD x % 8 (round-robin placement) and takes
  // entry x / 8 of that XCD's queue = all tiles of frequency fmap[xcd][0], then of fmap[xcd][1], ...  Every tile of
  // a frequency then shares ONE L2: V[f] and U[f] are fetched from HBM once instead of once per XCD that
  // happens to hold one of the frequency's tiles (measured 2.3 - 3x the unique bytes with the tile-residue maps
  // on the F(4x4,3x3) shapes, where the GEMMs run at 2.7 - 4.2 TB/s of HBM traffic).
  signed char fmap[8][8];   // frequency per (XCD, slot), -1 = none; + 64: first half of the tiles only, + 128 (as
                            // unsigned): second half (36 frequencies do not divide by 8: the last four are shared
                            // by two XCDs each so that every XCD carries 4.5 frequencies)
  // Strided layers: the zero-padded 2-tap windows make the filter transform of an even-parity
  // class vanish at one frequency index per dimension (G row 0 picks the zero tap in the forward
  // orientation, the last row in the flipped one), so 23 of the 144 (class, frequency) blocks are
  // structurally zero and are skipped: seg_mode 1 = classes along K (forward), 2 = along N
  // (dgrad), 3 = along M (wgrad); seg_len = channels per class; seg_skip = vanishing index.
  int seg_mode, seg_len, seg_skip;
  // split-precision operands (NT only): three bf16 planes each in the blocked layout (op_off);
  // pA / pB = plane strides, sAp / sBp = frequency strides, rbA / rbB = row blocks of 32, kblocks = K / 16
  const u16* Ap;
  const u16* Bp;
  long pA, pB, sAp, sBp;
  int rbA, rbB, kblocks;
  int cbA, cbB;   // TL operands: column blocks of 16 (M / 16, N / 16)
  // Matching GEMMs (sinkhorn.hip): blockIdx.z = problem / output block with its own operand offsets (u16
  // elements from Ap / Bp), C offset (floats from C) and contraction length, instead of the uniform
  // frequency strides; m_begin = first output row of the launch (row-range plan application: tiles start
  // there, C is indexed by the absolute row); epi = 1: C = epi_scale * acc + epi_bias (+ epi_diag on
  // the diagonal m == n) -- the log-kernel -lambda * (1 - x.y) straight from the accumulators.
  int ztab, m_begin, epi;
  long zA[8], zB[8], zC[8];
  int zK[8];
  float epi_scale, epi_bias, epi_diag[8];
  // Scaled two-piece operands (X3_PIECES == 2): the operands' headers (X3_HDR floats in front of the planes:
  // [0] largest magnitude of the source tensor, [16 + f] scale 2^(14 - e_f), [64 + f] its inverse);
  // C = hdrA[64 + f] * hdrB[64 + f] * sum (null: unscaled operands)
  const float* hdrA;
  const float* hdrB;
  // Stream kernel (wino_bgemm_x3_stream_kernel): one persistent workgroup per compute unit; workgroup c of XCD x
  // owns the positions [sk_bound[x][c], sk_bound[x][c + 1]) of that XCD's queue (x3_plan_stream).
  unsigned sk_bound[8][41];
  int sk_nw;                          // workgroups per XCD (grid = 8 sk_nw)
  float* sk_partial;                  // [grid] parked 256 x 256 accumulator tiles, then [grid] 64-bit flags
  unsigned long long sk_epoch;        // flag value of this launch
  unsigned long long sk_inv_tn;       // ceil(2^32 / tiles_n): tile / tiles_n without a division on the device
  unsigned char sk_cmask[64];         // strided dgrad / wgrad: parity classes a column / row tile overlaps
#ifdef X3_TIMING
  unsigned long long* dbg;   // tools/ablate/x3_phase.hip: s_memtime stamps of workgroup x at dbg[64 x ..]
#endif
  unsigned x_total;          // 256 x 128 kernel: number of queue positions; the grid may be smaller (tile loop)
};
// 2^(eA + eB - 28) of frequency f (scaled two-piece operands): a product of two powers of two, exact
__device__ __forceinline__ float x3_out_scale(const BgArgs& a, int f) { return a.hdrA[64 + f] * a.hdrB[64 + f]; }
#ifdef X3_TIMING
// (stamps go to LDS past the stage buffers: a global store would count in the vmcnt waits of the pipeline)
#define X3_STAMP(i) do { if (threadIdx.x == 0 && (i) < 64) reinterpret_cast<unsigned long long*>(smem3 + X3_LDS)[i] = __builtin_readcyclecounter(); } while (0)
#else
#define X3_STAMP(i) do { } while (0)
#endif


// Forward pass of a strided layer (seg_mode 1): the contraction length depends on the frequency -- those with both
// indices non-zero have all four classes (full K), one zero index two classes (K / 2), frequency 0 one (K / 4).
// Workgroups are dispatched in blockIdx order, z slowest; in the natural order the LAST frequencies are
// full-length and the launch ends with a long tail.  Longest-processing-time-first order instead.
__device__ __forceinline__ int lpt_frequency(int seg_mode, int z) {
  if (seg_mode != 1) return z;
  constexpr int n = kWA - 1, full = n * n;
  if (z < full) return (1 + z / n) * kWA + 1 + z % n;          // both indices non-zero
  if (z < full + n) return 1 + (z - full);                       // row index 0
  if (z < full + 2 * n) return (1 + (z - full - n)) * kWA;       // column index 0
  return 0;
}
// ---- the NT GEMM on the bf16 pipe (split-precision operands) --------------------------------
// 256 x 256 block tile, FOUR waves (2 x 2) = one wave per SIMD with a 128 x 128 wave tile: 16 accumulator
// tiles of 32x32 (256 AGPRs) and both fragment sets of a K stage double-buffered in VGPRs, so the 24
// ds_read_b128 of stage k+1 are interleaved with the 96 MFMAs of stage k (one read per four MFMAs) and
// no MFMA waits for an LDS round trip.  K stage = 16: per (operand, piece) 256 rows x 32 bytes, three
// stages in LDS (144 KiB) filled by global_load_lds straight from the blocked operand layout (op_off:
// one instruction = one contiguous 1 KiB chunk = 32 rows; no VGPR staging, no ds_write pass), issued
// three stages ahead; one barrier per stage.  Waves 0,1 fetch A, waves 2,3 fetch B (12 chunks each).
// tools/ablate/gemm_bf16x3_v3.hip has the prototypes and the measurements behind these choices.
constexpr int X3_BM = 256, X3_BN = 256, X3_BK = 32;     // X3_BK: granularity of K (two stages)
constexpr int X3_SK = 16, X3_MT = 4, X3_NT = 4, X3_THREADS = 256, X3_NSTAGE = 3;
constexpr int X3_TA = X3_BM * X3_SK * 2, X3_TB = X3_BN * X3_SK * 2;   // bytes per (operand, piece, stage)
constexpr int X3_STAGE = X3_NP * (X3_TA + X3_TB);
constexpr size_t X3_LDS = (size_t)X3_NSTAGE * X3_STAGE;
constexpr int X3_PER_WAVE = X3_NP * (X3_BM + X3_BN) / 32 / 4;              // global_load_lds per wave per stage

struct X3Frags {
  x3frag_t a[X3_MT][X3_NP];
  x3frag_t b[X3_NT][X3_NP];
};

// s_waitcnt vmcnt(n) only (expcnt / lgkmcnt untouched): gfx9 encoding vmcnt[3:0] | expcnt 7 << 4 | lgkmcnt 15 << 8 | vmcnt[5:4] << 14
#define X3_WAIT_VM(n) __builtin_amdgcn_s_waitcnt(0x0f70 | ((n) & 15) | (((n) >> 4) << 14))

// blockIdx.y = K split (wgrad: slabs, reduced by the adjoint filter transform).
// TL ("t-leading"): both operands are stored with the CONTRACTION index as the row of the blocked layout
// (A = [K][M], B = [K][N]: the weight-gradient GEMMs contract over the tiles and read the forward / dgrad
// operands V[tile][channel] as they are).  A stage is then 16 rows of 16 + 16 column blocks; the fragments
// (eight consecutive k of one column per lane) come out of ds_read_b64_tr_b16: a 16-lane group reads a
// [4 k][16 columns] block, lane l the four columns 4(l%4).. of row l/4, and receives column l of all four rows.
template <bool PIPE, bool TL>
__global__ __launch_bounds__(X3_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1))) void wino_bgemm_x3_kernel(BgArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
  const int x = blockIdx.x;
  X3_STAMP(0);
  int tm, tn, fsel = -1;
  if (a.xmap == 4) {
    const int tiles = a.tiles_m * a.tiles_n;
    const int xcd = x & 7, idx = x >> 3;
    const int slot = idx / tiles, tile = idx - slot * tiles;
    const int code = (unsigned char)a.fmap[xcd][slot];
    if (code == 255) return;
    const int half = (tiles + 1) >> 1;
    if ((code & 64) && tile >= half) return;
    if ((code & 128) && tile < half) return;
    // (the division runs on the vector ALU: the operand addresses of the global_load_lds stream must be scalar)
    fsel = __builtin_amdgcn_readfirstlane(code & 63);
    tn = __builtin_amdgcn_readfirstlane(tile % a.tiles_n);
    tm = __builtin_amdgcn_readfirstlane(tile / a.tiles_n);
  } else if (a.xmap == 1) {
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
  const int f = fsel >= 0 ? fsel : lpt_frequency(a.seg_mode, blockIdx.z);
  const int m0 = a.m_begin + tm * X3_BM, n0 = tn * X3_BN;
  const int Kz = a.ztab ? a.zK[f] : a.K;
  if (a.seg_mode == 2 || a.seg_mode == 3) {
    // strided layers: skip tiles whose columns (dgrad) / rows (wgrad) belong to classes all absent at f
    const int lo = a.seg_mode == 2 ? n0 : m0, ext = a.seg_mode == 2 ? a.N : a.M;
    int hi = lo + (a.seg_mode == 2 ? X3_BN : X3_BM) - 1;
    if (hi >= ext) hi = ext - 1;
    bool any = false;
    for (int c = lo / a.seg_len; c <= hi / a.seg_len; ++c) any = any || s2_present(c, f, a.seg_skip);
    if (!any) return;
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, g = lane >> 5;

  // K runs: the whole K, or (forward of a strided layer) the <= 2 runs of classes present at this frequency
  int lo0 = 0, len0 = 0, lo1 = 0, len1 = 0;
  if (a.seg_mode == 1) {
    int c = 0, nrun = 0;
    while (c < 4) {
      if (!s2_present(c, f, a.seg_skip)) {
        ++c;
        continue;
      }
      int e = c + 1;
      while (e < 4 && s2_present(e, f, a.seg_skip)) ++e;
      if (nrun == 0) {
        lo0 = c * a.seg_len;
        len0 = (e - c) * a.seg_len;
      } else {
        lo1 = c * a.seg_len;
        len1 = (e - c) * a.seg_len;
      }
      ++nrun;
      c = e;
    }
  } else {
    // K split: blockIdx.y takes kt_per_split granules of X3_BK (all of K when there is one split)
    const int nkt_all = Kz / X3_BK;
    const int kt0 = blockIdx.y * a.kt_per_split;
    int nkt = nkt_all - kt0;
    if (nkt > a.kt_per_split) nkt = a.kt_per_split;
    if (nkt < 0) nkt = 0;
    lo0 = kt0 * X3_BK;
    len0 = nkt * X3_BK;
  }
  // the runs as one sequence of stages; kb_of = k block (of 16) of a stage
  const int steps0 = len0 / X3_SK;
  const int nst = steps0 + len1 / X3_SK;
  const int kb0 = lo0 / X3_SK, kb1 = lo1 / X3_SK - steps0;
  auto kb_of = [&](int st) { return st < steps0 ? kb0 + st : kb1 + st; };

  // this wave's 12 chunk streams: operand (A for waves 0,1), piece, row block (clamped at the operand's end:
  // the duplicated rows only feed C rows / columns that are not stored)
  const bool isA = wave < 2;
  const int half = wave & 1;
  const u16* opb = isA ? a.Ap + (a.ztab ? a.zA[f] : f * a.sAp) : a.Bp + (a.ztab ? a.zB[f] : f * a.sBp);
  const long plane = isA ? a.pA : a.pB;
  const int rb0 = (isA ? m0 : n0) >> 5, rbmax = (isA ? a.rbA : a.rbB) - 1;
  // TL: column blocks of 16; one instruction fetches the same 16 rows of two adjacent column blocks
  const int cb0 = (isA ? m0 : n0) >> 4, cbn = isA ? a.cbA : a.cbB;
  const unsigned voff = TL ? (unsigned)(lane >> 5) * 1024u + (unsigned)(lane & 31) * 16u : (unsigned)lane * 16u;
  const unsigned lds_base = (unsigned)(unsigned long)(__attribute__((address_space(3))) void*)smem3;
  auto issue = [&](int st, int buf, int i0 = 0, int n = X3_PER_WAVE) {
    const long kb = kb_of(st);
#pragma unroll
    for (int i = i0; i < i0 + n; ++i) {
      const int li = half * X3_PER_WAVE + i;   // 0..23 within the operand: piece = li / 8, row group = li % 8
      const int piece = li >> 3, rg = li & 7;
      const u16* src;
      if (TL) {
        int cb = cb0 + 2 * rg;
        // past the operand's last column: duplicates feed C rows / columns never stored.  An odd block count (columns a
        // multiple of 16 only) pairs the last block with the 1 KiB after it (the next row group's first block, or the
        // operand that follows in the workspace): read, multiplied into columns that are never stored
        if (cb > ((cbn - 1) & ~1)) cb = (cbn - 1) & ~1;
        src = opb + piece * plane + ((((long)(kb >> 1) * cbn + cb) << 9) + ((kb & 1) << 8));
      } else {
        int rb = rb0 + rg;
        if (rb > rbmax) rb = rbmax;
        src = opb + piece * plane + (((long)rb * a.kblocks + kb) << 9);
      }
      const unsigned dst = lds_base + buf * X3_STAGE + (isA ? 0 : X3_NP * X3_TA) + piece * X3_TA + rg * 1024;
      // scalar base + per-lane 32-bit offset (the builtin expands to 64-bit per-lane addresses inside the loop);
      // M0 = LDS address of the chunk.  Nothing else in this kernel uses M0.  The s_nop is REQUIRED: an SALU write of
      // M0 needs one wait state before an LDS-DMA instruction reads it, and nothing pads the inside of an asm statement
      // (cdna_hip_programming.md 5.7).  Without it a load occasionally went to the PREVIOUS M0 -- the chunk before it, or
      // for a wave's first load whatever M0 held at launch, which can lie outside the workgroup's LDS: round 3 saw a
      // co-resident workgroup of ANOTHER kernel (conv_rgbin_fwd) corrupted once the 256 x 128 kernel let others share a
      // compute unit (tools/debug/dist_two_rank_trace.py); the 256 x 256 kernel owns its compute unit and never showed it.
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(src), "s"(dst) : "memory");
    }
  };
  f32x16 acc[X3_MT][X3_NT];
#pragma unroll
  for (int i = 0; i < X3_MT; ++i)
#pragma unroll
    for (int j = 0; j < X3_NT; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
  // fragment of MFMA tile t, piece p: row r of the tile, k half g; the XOR matches op_off's slot swizzle.
  // TL: LDS image per (operand, piece) = [16 column blocks][16 k rows][32 bytes]; lane = (group g4 of 16, l):
  // column block 2t + g4 % 2 of the wave's eight, rows 8 (g4 / 2) + l / 4 (+ 4 for the second read), 8-byte
  // column group l % 4; the row's swizzle bit is g4 / 2.
  const int sw = (r >> 3) & 1;
  const int g4 = lane >> 4, l16 = lane & 15;
  const int ftl = (g4 & 1) * 512 + (8 * (g4 >> 1) + (l16 >> 2)) * 32 + ((((l16 >> 1) & 1) ^ (g4 >> 1)) * 16) + (l16 & 1) * 8;
  const int fa = TL ? wm * 4096 + ftl : (wm * X3_MT * 32 + r) * 32 + 16 * (g ^ sw);
  const int fb = X3_NP * X3_TA + (TL ? wn * 4096 + ftl : (wn * X3_NT * 32 + r) * 32 + 16 * (g ^ sw));
  auto read_frag = [&](const unsigned char* p) -> x3frag_t {
    if (TL) {
      typedef short s16x4 __attribute__((ext_vector_type(4)));
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 128));
      return __builtin_bit_cast(x3frag_t, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    } else {
      return *reinterpret_cast<const x3frag_t*>(p);
    }
  };
  auto load_frags = [&](X3Frags& F, int buf) {
    const unsigned char* pa = smem3 + buf * X3_STAGE + fa;
    const unsigned char* pb = smem3 + buf * X3_STAGE + fb;
#pragma unroll
    for (int t = 0; t < X3_MT; ++t)
#pragma unroll
      for (int p = 0; p < X3_NP; ++p) F.a[t][p] = read_frag(pa + p * X3_TA + t * 1024);
#pragma unroll
    for (int t = 0; t < X3_NT; ++t)
#pragma unroll
      for (int p = 0; p < X3_NP; ++p) F.b[t][p] = read_frag(pb + p * X3_TB + t * 1024);
  };
  // six products per fp32-exact product, smallest terms first; consecutive MFMAs hit different accumulators
  auto mfmas = [&](const X3Frags& F) {
#pragma unroll
    for (int t = 0; t < X3_NTERM; ++t)
#pragma unroll
      for (int i = 0; i < X3_MT; ++i)
#pragma unroll
        for (int j = 0; j < X3_NT; ++j) acc[i][j] = X3_MFMA(F.a[i][X3_PA[t]], F.b[j][X3_PB[t]], acc[i][j]);
  };
  if (PIPE) {
    // nst is even and >= 4 (host contract).  One stage: stage st+1 has landed (barrier), the buffer stage st
    // was read from is refilled with stage st+3, then the 96 MFMAs of stage st on F with the 24 fragment
    // reads of stage st+1 (into G) in between.  ISSUE: stage st+3 exists; PEND: stage st+2 is in flight;
    // LOAD: stage st+1 exists.
    auto stage = [&](int st, int bufn, const X3Frags& F, X3Frags& G, auto issue_c, auto pend_c, auto load_c) {
      constexpr bool ISSUE = decltype(issue_c)::value, PEND = decltype(pend_c)::value, LOAD = decltype(load_c)::value;
      // vmcnt: this wave's share of stage st+1 has landed; lgkmcnt(0): its fragment reads of stage st have RETURNED --
      // the buffer they came from is refilled right after the barrier (round 3: seen as a race in the 256 x 128
      // kernel, two workgroups per compute unit; the same order is kept here.  Inline asm: the compiler's waitcnt pass
      // drops an lgkmcnt(0) it finds in an s_waitcnt builtin)
      if (PEND) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(X3_PER_WAVE) : "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      // six term groups of 16 MFMAs; in front of each: two of the twelve refill loads (all twelve at once keep the
      // wave in its VMEM issue queue for several hundred cycles while the matrix pipe drains) and four of the 24
      // fragment reads of the next stage
      const int rbuf = bufn == 0 ? X3_NSTAGE - 1 : bufn - 1;
      const unsigned char* pa = smem3 + bufn * X3_STAGE + fa;
      const unsigned char* pb = smem3 + bufn * X3_STAGE + fb;
      constexpr int NR = 8 * X3_NP;   // fragment reads of a stage
      x3_static_for<X3_NTERM>([&](auto tc) {
        // term group t: its share of the refill loads and of the next stage's fragment reads in front of 16 MFMAs
        constexpr int t = decltype(tc)::value;
        constexpr int l0 = t * X3_PER_WAVE / X3_NTERM, l1 = (t + 1) * X3_PER_WAVE / X3_NTERM;
        constexpr int q0 = t * NR / X3_NTERM, q1 = (t + 1) * NR / X3_NTERM;
        if (ISSUE) issue(st + 3, rbuf, l0, l1 - l0);
        if (LOAD) {
#pragma unroll
          for (int q = q0; q < q1; ++q) {
            const int qq = q % (4 * X3_NP), tt = qq / X3_NP, p = qq % X3_NP;   // q < 4 NP: A fragments, else B
            if (q < 4 * X3_NP) G.a[tt][p] = read_frag(pa + p * X3_TA + tt * 1024);
            else G.b[tt][p] = read_frag(pb + p * X3_TB + tt * 1024);
          }
        }
#pragma unroll
        for (int i = 0; i < X3_MT; ++i)
#pragma unroll
          for (int j = 0; j < X3_NT; ++j) acc[i][j] = X3_MFMA(F.a[i][X3_PA[t]], F.b[j][X3_PB[t]], acc[i][j]);
        if (LOAD) x3_sched_group<TL ? 2 : 1, q1 - q0>();
      });
    };
    using Y = std::true_type;
    using N = std::false_type;
    X3Frags F0, F1;
    issue(0, 0);
    issue(1, 1);
    issue(2, 2);
    X3_WAIT_VM(2 * X3_PER_WAVE);
    __builtin_amdgcn_s_barrier();
    load_frags(F0, 0);
    X3_STAMP(1);
    int st = 0, bufn = 1;   // bufn = buffer of stage st + 1
    auto next = [&]() { bufn = bufn == X3_NSTAGE - 1 ? 0 : bufn + 1; };
    for (; st + 6 <= nst; st += 2) {
      stage(st, bufn, F0, F1, Y{}, Y{}, Y{});
      next();
      stage(st + 1, bufn, F1, F0, Y{}, Y{}, Y{});
      next();
      X3_STAMP(4 + (st >> 1));
    }
    stage(st, bufn, F0, F1, Y{}, Y{}, Y{});
    next();
    stage(st + 1, bufn, F1, F0, N{}, Y{}, Y{});
    next();
    stage(st + 2, bufn, F0, F1, N{}, N{}, Y{});
    next();
    stage(st + 3, bufn, F1, F0, N{}, N{}, N{});
    X3_STAMP(2);
  } else {
    // any stage count (short K runs, ragged K splits: small layers only): one stage at a time
    X3Frags F;
    for (int st = 0; st < nst; ++st) {
      issue(st, 0);
      X3_WAIT_VM(0);
      __builtin_amdgcn_s_barrier();
      load_frags(F, 0);
      mfmas(F);
      __builtin_amdgcn_s_barrier();
    }
  }
  float* C = a.C + (a.ztab ? a.zC[f] : f * a.sC) + blockIdx.y * a.sSplit;
  float es = a.epi ? a.epi_scale : 1.f;
  const float eb = a.epi ? a.epi_bias : 0.f, ed = a.epi ? a.epi_diag[f & 7] : 0.f;
  if (a.hdrA) es *= x3_out_scale(a, f);
  if (m0 + X3_BM <= a.M && n0 + X3_BN <= a.N && ed == 0.f) {
    // whole tile inside C: one running row pointer per lane, the four column tiles of a row at immediate offsets
    // (with a bounds test, an address multiply and the epilogue select per element this loop took 26.8 k cycles per
    // tile -- as long as eight K stages; tools/ablate/x3_phase.hip)
    const long ld = a.ldc;
    float* row = C + (long)(m0 + wm * X3_MT * 32 + 4 * g) * ld + (n0 + wn * X3_NT * 32 + r);
#pragma unroll
    for (int i = 0; i < X3_MT; ++i) {
#pragma unroll
      for (int qh = 0; qh < 4; ++qh) {
        float* p = row + (long)(i * 32 + 8 * qh) * ld;
#pragma unroll
        for (int ql = 0; ql < 4; ++ql) {
#pragma unroll
          for (int j = 0; j < X3_NT; ++j) {
            const float v = acc[i][j][4 * qh + ql];
            p[j * 32] = fmaf(es, v, eb);   // es = 1, eb = 0 without an epilogue: exact (nontemporal stores: no change)
          }
          p += ld;
        }
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < X3_MT; ++i)
#pragma unroll
      for (int j = 0; j < X3_NT; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int rr = (q & 3) + 8 * (q >> 2) + 4 * g;
          const int m = m0 + (wm * X3_MT + i) * 32 + rr;
          const int n = n0 + (wn * X3_NT + j) * 32 + r;
          if (m < a.M && n < a.N) {
            float v = acc[i][j][q];
            if (a.epi || a.hdrA) v = fmaf(es, v, eb) + (m == n ? ed : 0.f);
            C[(long)m * a.ldc + n] = v;
          }
        }
  }
#ifdef X3_TIMING
  X3_STAMP(3);
  __builtin_amdgcn_s_waitcnt(0);
  X3_STAMP(63);
  if (threadIdx.x < 64) a.dbg[(long)blockIdx.x * 64 + threadIdx.x] = reinterpret_cast<unsigned long long*>(smem3 + X3_LDS)[threadIdx.x];
#endif
}


#if X3_PIECES == 2
// ---- the same GEMM on a 256 x 128 tile, TWO workgroups per compute unit (round 3) ------------------------------
// Where a 256 x 256 tile's time goes with three MFMAs per product (tools/ablate/x3_phase.hip, cycles, K = 256):
// 6 k until the first fragments are in registers (every compute unit's first 96 KB arrive in one HBM burst), 32 k in
// the K loop (24.6 k of MFMA issue), 12 k writing C (256 KB per compute unit at the ~21 B/clk a compute unit gets
// out of its L2 write path -- request-bound, not store-issue bound: 16-byte stores of 32-byte row segments took
// 22 k), and the next workgroup reaches the compute unit a few thousand cycles later.  A one-wave-per-SIMD
// workgroup owns the compute unit (512 registers, 144 KB of LDS), so nothing runs under its prologue and write-out:
// 40 % of a K = 256 tile, 25 % at K = 512.  Here the tile is 256 x 128 with the same four waves (wave tile
// 128 x 64: eight accumulator tiles = 128 registers, both fragment sets double-buffered = 96, 256 in all) and three
// stages of 24 KB: two workgroups fit a compute unit and two waves share a SIMD, so one workgroup's write-out,
// prologue, barrier waits and global_load_lds issue stalls (100 - 185 cycles each next to fragment reads:
// MI355X_MICROARCH.md) are covered by the other's MFMAs.  Price: 1.5 x the operand bytes (L2 -> LDS, fragment reads)
// per product.  Same operand layout, pipeline (three stages ahead, one barrier per stage, counted vmcnt), grid maps
// and strided-layer handling as wino_bgemm_x3_kernel<true, TL>; host contract: nst even and >= 4.
constexpr int X3N_NT = 2, X3N_BN = X3N_NT * 64;
constexpr int X3N_TB = X3N_BN * X3_SK * 2;
constexpr int X3N_STAGE = X3_NP * (X3_TA + X3N_TB);
constexpr size_t X3N_LDS = (size_t)X3_NSTAGE * X3N_STAGE;
constexpr int X3N_CHA = X3_NP * X3_BM / 32, X3N_CHB = X3_NP * X3N_BN / 32;   // 1 KiB chunks per stage: A 16, B 8
constexpr int X3N_PER_WAVE = (X3N_CHA + X3N_CHB) / 4;
static_assert(X3N_PER_WAVE * 4 == X3N_CHA + X3N_CHB, "chunks divide over four waves");

struct X3NFrags {
  x3frag_t a[X3_MT][X3_NP];
  x3frag_t b[X3N_NT][X3_NP];
};
// R fragment reads spread over the NM MFMAs of a term group
template <int DS, int R, int NM, int Q = 0>
__device__ __forceinline__ void x3n_sched_group() {
  if constexpr (Q < R) {
    __builtin_amdgcn_sched_group_barrier(0x100, DS, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, (Q + 1) * NM / R - Q * NM / R, 0);
    x3n_sched_group<DS, R, NM, Q + 1>();
  }
}

#ifdef X3_TIMING
#define X3N_STAMP(i) do { if (threadIdx.x == 0 && (i) < 64) reinterpret_cast<unsigned long long*>(smem3 + X3N_LDS)[i] = __builtin_readcyclecounter(); } while (0)
#else
#define X3N_STAMP(i) do { } while (0)
#endif
#ifndef X3N_NUM_VGPR
#define X3N_NUM_VGPR 128
#endif
// Round 4: the kernel walks tiles x, x + gridDim.x, ... of the frequency-major queue (gridDim.x a multiple of 8: a
// workgroup stays on its XCD).  With a grid capped at the resident workgroups (two per compute unit) a workgroup starts
// the first three stages of its NEXT tile -- LDS-DMA loads, no registers -- before it writes the current one out: they land
// during the 256 stores per lane, and the next tile begins with its operands in LDS instead of with an HBM round trip
// (6 - 7 k of a K = 256 tile's 43 k cycles).  Uncapped grid: one tile per workgroup, exactly the round-3 kernel.
struct X3NTile {
  int f, m0, n0, nst, steps0, kb0, kb1;
  const u16* cbase[X3N_PER_WAVE];
  int ccbn[X3N_PER_WAVE];
};
template <bool TL>
__global__ __launch_bounds__(X3_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2), amdgpu_num_vgpr(X3N_NUM_VGPR))) void wino_bgemm_x3n_kernel(BgArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
  X3N_STAMP(0);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, g = lane >> 5;
  const unsigned voff = TL ? (unsigned)(lane >> 5) * 1024u + (unsigned)(lane & 31) * 16u : (unsigned)lane * 16u;
  const unsigned lds_base = (unsigned)(unsigned long)(__attribute__((address_space(3))) void*)smem3;
  // LDS offsets of this wave's six chunk streams inside a stage (A: piece x eight row groups of 32, then B: piece x four)
  unsigned cdst[X3N_PER_WAVE];
#pragma unroll
  for (int i = 0; i < X3N_PER_WAVE; ++i) {
    const int li = wave * X3N_PER_WAVE + i;
    const bool isA = li < X3N_CHA;
    const int l2 = isA ? li : li - X3N_CHA;
    const int piece = isA ? l2 >> 3 : l2 >> 2, rg = isA ? l2 & 7 : l2 & 3;
    cdst[i] = lds_base + (isA ? piece * X3_TA + rg * 1024 : X3_NP * X3_TA + piece * X3N_TB + rg * 1024);
  }
  // tile x of the queue -> its frequency, position, K stages and operand streams; false: nothing to compute there
  auto setup = [&](int x, X3NTile& T) -> bool {
    int tm, tn, fsel = -1;
    if (a.xmap == 4) {
      const int tiles = a.tiles_m * a.tiles_n;
      const int xcd = x & 7, idx = x >> 3;
      const int slot = idx / tiles, tile = idx - slot * tiles;
      const int code = (unsigned char)a.fmap[xcd][slot];
      if (code == 255) return false;
      const int half = (tiles + 1) >> 1;
      if ((code & 64) && tile >= half) return false;
      if ((code & 128) && tile < half) return false;
      fsel = __builtin_amdgcn_readfirstlane(code & 63);
      tn = __builtin_amdgcn_readfirstlane(tile % a.tiles_n);
      tm = __builtin_amdgcn_readfirstlane(tile / a.tiles_n);
    } else {
      tn = x % a.tiles_n;
      tm = x / a.tiles_n;
    }
    const int f = fsel >= 0 ? fsel : lpt_frequency(a.seg_mode, blockIdx.z);
    const int m0 = a.m_begin + tm * X3_BM, n0 = tn * X3N_BN;
    const int Kz = a.ztab ? a.zK[f] : a.K;
    if (a.seg_mode == 2 || a.seg_mode == 3) {
      const int lo = a.seg_mode == 2 ? n0 : m0, ext = a.seg_mode == 2 ? a.N : a.M;
      int hi = lo + (a.seg_mode == 2 ? X3N_BN : X3_BM) - 1;
      if (hi >= ext) hi = ext - 1;
      bool any = false;
      for (int c = lo / a.seg_len; c <= hi / a.seg_len; ++c) any = any || s2_present(c, f, a.seg_skip);
      if (!any) return false;
    }
    int lo0 = 0, len0 = 0, lo1 = 0, len1 = 0;
    if (a.seg_mode == 1) {
      int c = 0, nrun = 0;
      while (c < 4) {
        if (!s2_present(c, f, a.seg_skip)) {
          ++c;
          continue;
        }
        int e = c + 1;
        while (e < 4 && s2_present(e, f, a.seg_skip)) ++e;
        if (nrun == 0) {
          lo0 = c * a.seg_len;
          len0 = (e - c) * a.seg_len;
        } else {
          lo1 = c * a.seg_len;
          len1 = (e - c) * a.seg_len;
        }
        ++nrun;
        c = e;
      }
    } else {
      const int nkt_all = Kz / X3_BK;
      const int kt0 = blockIdx.y * a.kt_per_split;
      int nkt = nkt_all - kt0;
      if (nkt > a.kt_per_split) nkt = a.kt_per_split;
      if (nkt < 0) nkt = 0;
      lo0 = kt0 * X3_BK;
      len0 = nkt * X3_BK;
    }
    T.f = f; T.m0 = m0; T.n0 = n0;
    T.steps0 = len0 / X3_SK;
    T.nst = T.steps0 + len1 / X3_SK;
    T.kb0 = lo0 / X3_SK;
    T.kb1 = lo1 / X3_SK - T.steps0;
    const u16* opA = a.Ap + (a.ztab ? a.zA[f] : f * a.sAp);
    const u16* opB = a.Bp + (a.ztab ? a.zB[f] : f * a.sBp);
#pragma unroll
    for (int i = 0; i < X3N_PER_WAVE; ++i) {
      const int li = wave * X3N_PER_WAVE + i;
      const bool isA = li < X3N_CHA;
      const int l2 = isA ? li : li - X3N_CHA;
      const int piece = isA ? l2 >> 3 : l2 >> 2, rg = isA ? l2 & 7 : l2 & 3;
      const u16* opb = (isA ? opA : opB) + piece * (isA ? a.pA : a.pB);
      if (TL) {
        const int cbn = isA ? a.cbA : a.cbB;
        int cb = ((isA ? m0 : n0) >> 4) + 2 * rg;
        if (cb > ((cbn - 1) & ~1)) cb = (cbn - 1) & ~1;   // (see the 256 x 256 kernel: duplicates feed columns never stored)
        T.cbase[i] = opb + ((long)cb << 9);
        T.ccbn[i] = cbn;
      } else {
        int rb = ((isA ? m0 : n0) >> 5) + rg;
        const int rbmax = (isA ? a.rbA : a.rbB) - 1;
        if (rb > rbmax) rb = rbmax;
        T.cbase[i] = opb + (((long)rb * a.kblocks) << 9);
        T.ccbn[i] = 0;
      }
    }
    return true;
  };
  auto issue = [&](const X3NTile& T, int st, int buf, int i0 = 0, int n = X3N_PER_WAVE) {
    const long kb = st < T.steps0 ? T.kb0 + st : T.kb1 + st;
#pragma unroll
    for (int i = i0; i < i0 + n; ++i) {
      const u16* src = T.cbase[i] + (TL ? ((((kb >> 1) * T.ccbn[i]) << 9) + ((kb & 1) << 8)) : (kb << 9));
      const unsigned dst = cdst[i] + buf * X3N_STAGE;
#ifdef X3N_DBG_NODMA   // (tools/debug/build_corun_variants.sh: the kernel as a neighbour without one of its ingredients; results garbage)
      asm volatile("" ::"v"(voff), "s"(src), "s"(dst) : "memory");
#else
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(src), "s"(dst) : "memory");
#endif
    }
  };
  const int sw = (r >> 3) & 1;
  const int g4 = lane >> 4, l16 = lane & 15;
  const int ftl = (g4 & 1) * 512 + (8 * (g4 >> 1) + (l16 >> 2)) * 32 + ((((l16 >> 1) & 1) ^ (g4 >> 1)) * 16) + (l16 & 1) * 8;
  const int fa = TL ? wm * 4096 + ftl : (wm * X3_MT * 32 + r) * 32 + 16 * (g ^ sw);
  const int fb = X3_NP * X3_TA + (TL ? wn * 2048 + ftl : (wn * X3N_NT * 32 + r) * 32 + 16 * (g ^ sw));
  auto read_frag = [&](const unsigned char* p) -> x3frag_t {
    if (TL) {
      typedef short s16x4 __attribute__((ext_vector_type(4)));
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 128));
      return __builtin_bit_cast(x3frag_t, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    } else {
      return *reinterpret_cast<const x3frag_t*>(p);
    }
  };
  auto load_frags = [&](X3NFrags& F, int buf) {
    const unsigned char* pa = smem3 + buf * X3N_STAGE + fa;
    const unsigned char* pb = smem3 + buf * X3N_STAGE + fb;
#pragma unroll
    for (int t = 0; t < X3_MT; ++t)
#pragma unroll
      for (int p = 0; p < X3_NP; ++p) F.a[t][p] = read_frag(pa + p * X3_TA + t * 1024);
#pragma unroll
    for (int t = 0; t < X3N_NT; ++t)
#pragma unroll
      for (int p = 0; p < X3_NP; ++p) F.b[t][p] = read_frag(pb + p * X3N_TB + t * 1024);
  };
  f32x16 acc[X3_MT][X3N_NT];
  X3NTile T;
  // one stage: stage st+1 has landed (counted vmcnt + barrier), the buffer stage st was read from is refilled with
  // stage st+3, then the 24 MFMAs of stage st on F in three term groups of eight, each with two of the six refill
  // loads and four of the twelve fragment reads of stage st+1 (into G) in front
  auto stage = [&](int st, int bufn, const X3NFrags& F, X3NFrags& G, auto issue_c, auto pend_c, auto load_c) {
    constexpr bool ISSUE = decltype(issue_c)::value, PEND = decltype(pend_c)::value, LOAD = decltype(load_c)::value;
    // vmcnt: this wave's share of stage st+1 has landed; lgkmcnt(0): its fragment reads of stage st (issued during
    // stage st-1, the last ones just before this point) have RETURNED -- after the barrier the buffer they came from is
    // refilled, and with eight waves on the compute unit a read still queued in the LDS can be overtaken by the
    // first refill (seen: one wave's last B fragment of a stage stale, a few launches in ten; the MFMAs behind the
    // barrier need these fragments at once anyway)
    // (inline asm: the compiler's waitcnt pass rewrites an s_waitcnt BUILTIN and dropped its lgkmcnt(0) -- it tracks the
    // registers the reads fill, not the LDS bytes a later DMA overwrites; the kernel then raced a few launches in a
    // thousand, more with a second process on the GPU)
    static_assert(X3N_PER_WAVE == 6, "the wait below is written for six loads per wave and stage");
    if (PEND) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int rbuf = bufn == 0 ? X3_NSTAGE - 1 : bufn - 1;
    const unsigned char* pa = smem3 + bufn * X3N_STAGE + fa;
    const unsigned char* pb = smem3 + bufn * X3N_STAGE + fb;
    constexpr int NRA = X3_MT * X3_NP, NR = (X3_MT + X3N_NT) * X3_NP;   // fragment reads of a stage: A first
    x3_static_for<X3_NTERM>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      constexpr int l0 = t * X3N_PER_WAVE / X3_NTERM, l1 = (t + 1) * X3N_PER_WAVE / X3_NTERM;
      constexpr int q0 = t * NR / X3_NTERM, q1 = (t + 1) * NR / X3_NTERM;
      if (ISSUE) issue(T, st + 3, rbuf, l0, l1 - l0);
      if (LOAD) {
#pragma unroll
        for (int q = q0; q < q1; ++q) {
#ifdef X3N_DBG_NOREAD
          if (q < NRA) G.a[q / X3_NP][q % X3_NP] = F.a[q / X3_NP][q % X3_NP];
          else G.b[(q - NRA) / X3_NP][(q - NRA) % X3_NP] = F.b[(q - NRA) / X3_NP][(q - NRA) % X3_NP];
#else
          if (q < NRA) G.a[q / X3_NP][q % X3_NP] = read_frag(pa + (q % X3_NP) * X3_TA + (q / X3_NP) * 1024);
          else G.b[(q - NRA) / X3_NP][(q - NRA) % X3_NP] = read_frag(pb + ((q - NRA) % X3_NP) * X3N_TB + ((q - NRA) / X3_NP) * 1024);
#endif
        }
      }
#pragma unroll
      for (int i = 0; i < X3_MT; ++i)
#pragma unroll
        for (int j = 0; j < X3N_NT; ++j) {
#ifdef X3N_DBG_NOMFMA
          asm volatile("" : "+v"(acc[i][j]) : "v"(F.a[i][X3_PA[t]]), "v"(F.b[j][X3_PB[t]]));
#else
          acc[i][j] = X3_MFMA(F.a[i][X3_PA[t]], F.b[j][X3_PB[t]], acc[i][j]);
#endif
        }
      if (LOAD) x3n_sched_group<TL ? 2 : 1, q1 - q0, X3_MT * X3N_NT>();
    });
  };
  using Y = std::true_type;
  using N = std::false_type;
  const int xend = (int)a.x_total, xstep = (int)gridDim.x;
  int x = blockIdx.x;
  while (x < xend && !setup(x, T)) x += xstep;
  if (x >= xend) return;
  issue(T, 0, 0);
  issue(T, 1, 1);
  issue(T, 2, 2);
  for (;;) {
#pragma unroll
    for (int i = 0; i < X3_MT; ++i)
#pragma unroll
      for (int j = 0; j < X3N_NT; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    X3NFrags F0, F1;
    // (a following tile: the 18 loads were issued BEFORE the previous tile's stores -- at most 12 operations outstanding
    // means all of them have landed, completion being in order)
    X3_WAIT_VM(2 * X3N_PER_WAVE);
    __builtin_amdgcn_s_barrier();
    load_frags(F0, 0);
    X3N_STAMP(1);
    int st = 0, bufn = 1;
    const int nst = T.nst;
    auto next = [&]() { bufn = bufn == X3_NSTAGE - 1 ? 0 : bufn + 1; };
    for (; st + 6 <= nst; st += 2) {
      stage(st, bufn, F0, F1, Y{}, Y{}, Y{});
      next();
      stage(st + 1, bufn, F1, F0, Y{}, Y{}, Y{});
      next();
      X3N_STAMP(4 + (st >> 1));
    }
    stage(st, bufn, F0, F1, Y{}, Y{}, Y{});
    next();
    stage(st + 1, bufn, F1, F0, N{}, Y{}, Y{});
    next();
    stage(st + 2, bufn, F0, F1, N{}, N{}, Y{});
    next();
    stage(st + 3, bufn, F1, F0, N{}, N{}, N{});
    X3N_STAMP(2);

    // this tile's output position, then the next tile's operand streams (T is overwritten)
    const int f = T.f, m0 = T.m0, n0 = T.n0;
    int xn = x + xstep;
    bool have = false;
    while (xn < xend) {
      if (setup(xn, T)) {
        have = true;
        break;
      }
      xn += xstep;
    }
    if (have) {
      // every wave is past its last fragment read of this tile before the buffers are refilled
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      issue(T, 0, 0);
      issue(T, 1, 1);
      issue(T, 2, 2);
    }
    float* C = a.C + (a.ztab ? a.zC[f] : f * a.sC) + blockIdx.y * a.sSplit;
    float es = a.epi ? a.epi_scale : 1.f;
    const float eb = a.epi ? a.epi_bias : 0.f, ed = a.epi ? a.epi_diag[f & 7] : 0.f;
    if (a.hdrA) es *= x3_out_scale(a, f);
    if (m0 + X3_BM <= a.M && n0 + X3N_BN <= a.N && ed == 0.f) {
      const long ld = a.ldc;
      float* row = C + (long)(m0 + wm * X3_MT * 32 + 4 * g) * ld + (n0 + wn * X3N_NT * 32 + r);
#pragma unroll
      for (int i = 0; i < X3_MT; ++i) {
#pragma unroll
        for (int qh = 0; qh < 4; ++qh) {
          float* p = row + (long)(i * 32 + 8 * qh) * ld;
#pragma unroll
          for (int ql = 0; ql < 4; ++ql) {
#pragma unroll
            for (int j = 0; j < X3N_NT; ++j) p[j * 32] = fmaf(es, acc[i][j][4 * qh + ql], eb);
            p += ld;
          }
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < X3_MT; ++i)
#pragma unroll
        for (int j = 0; j < X3N_NT; ++j)
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int rr = (q & 3) + 8 * (q >> 2) + 4 * g;
            const int m = m0 + (wm * X3_MT + i) * 32 + rr;
            const int n = n0 + (wn * X3N_NT + j) * 32 + r;
            if (m < a.M && n < a.N) {
              float v = acc[i][j][q];
              if (a.epi || a.hdrA) v = fmaf(es, v, eb) + (m == n ? ed : 0.f);
              C[(long)m * a.ldc + n] = v;
            }
          }
    }
    if (!have) break;
    x = xn;
  }
#ifdef X3_TIMING
  X3N_STAMP(3);
  __builtin_amdgcn_s_waitcnt(0);
  X3N_STAMP(63);
  if (threadIdx.x < 64) a.dbg[(long)blockIdx.x * 64 + threadIdx.x] = reinterpret_cast<unsigned long long*>(smem3 + X3N_LDS)[threadIdx.x];
#endif
}
#endif   // X3_PIECES == 2

// Frequency-major grid (BgArgs::fmap): the 36 frequencies are dealt to the 8 XCDs -- equal work: four whole
// frequencies per XCD and the last four as halves on XCD pairs; strided layers: longest-processing-time-first on the
// number of parity classes present at a frequency.  Returns the grid size in x of the one-tile kernel.
inline unsigned x3_build_fmap(BgArgs& b) {
  int load[8] = {0, 0, 0, 0, 0, 0, 0, 0}, cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  memset(b.fmap, -1, sizeof(b.fmap));
  auto weight = [&](int f) {
    if (!b.seg_mode) return 4;
    int w = 0;
    for (int c = 0; c < 4; ++c) w += s2_present(c, f, b.seg_skip) ? 1 : 0;
    return w;
  };
  if (!b.seg_mode) {
    // equal work per frequency: four whole frequencies per XCD, the last four as halves (4.5 each)
    for (int f = 0; f < kWF; ++f) {
      if (f < 32) {
        b.fmap[f & 7][cnt[f & 7]++] = (signed char)f;
      } else {
        const int x0 = 2 * (f - 32);
        b.fmap[x0][cnt[x0]++] = (signed char)(f | 64);
        b.fmap[x0 + 1][cnt[x0 + 1]++] = (signed char)(f | 128);
      }
    }
    b.xmap = 4;
    return 8u * 5u * (unsigned)(b.tiles_m * b.tiles_n);
  }
  for (int w = 4; w >= 1; --w)
    for (int f = 0; f < kWF; ++f) {
      if (weight(f) != w) continue;
      int best = 0;
      for (int x = 1; x < 8; ++x)
        if (load[x] < load[best]) best = x;
      b.fmap[best][cnt[best]++] = (signed char)f;
      load[best] += w;
    }
  int slots = 0;
  for (int x = 0; x < 8; ++x) slots = cnt[x] > slots ? cnt[x] : slots;
  b.xmap = 4;
  return 8u * (unsigned)slots * (unsigned)(b.tiles_m * b.tiles_n);
}

// ---- the same GEMM as ONE persistent workgroup per compute unit ------------------------------------
// Measured on the Winograd-domain shapes (tools/ablate/x3_phase.hip, cycles per 256 x 256 tile): the K loop itself
// runs at 91 % of the matrix-pipe rate (3380 cycles per stage of 96 MFMAs), but a workgroup that owns one tile
// spends 9 k cycles waiting for its first operands, 12 k writing C in a burst that every compute unit issues at the
// same moment, and the next workgroup reaches the compute unit ~20 k cycles later -- against 54 k cycles of MFMA work
// at K = 256.  And a frequency-major grid leaves each XCD with 18 .. 288 tiles for 32 compute units (1.1 .. 9
// rounds, the last one partly empty).  Here every XCD's queue -- its frequencies' tiles, each a run of K stages --
// is cut into sk_nw equal contiguous ranges of stages, one per workgroup ("stream-K" within an XCD):
//  * the operand stream is continuous across tile boundaries: the loads of the next tile's first stages are in
//    flight while the current tile finishes and is written out;
//  * a range boundary may fall inside a tile.  A workgroup walks its range BACKWARDS, so the head part of a shared
//    tile is the first thing its workgroup computes -- parked in the workspace and flagged -- and the tail part the
//    last thing the next workgroup computes; that one adds the parked sums (fixed order: deterministic) and writes
//    C.  A workgroup only ever waits for lower-numbered workgroups, which were dispatched before it and park their
//    sums before waiting for anything themselves: no deadlock, whatever is resident;
//  * boundaries inside tiles also spread the C writes of the compute units over time.
constexpr int X3_SK_MAXW = 40;
constexpr size_t X3_SK_SLOT = (size_t)X3_BM * X3_BN;
inline size_t x3_stream_ws_floats(int grid) { return (size_t)grid * X3_SK_SLOT + 2 * (size_t)grid + 4; }
__host__ __device__ __forceinline__ unsigned x3_pos(int slot, int tile, int stage) {
  return ((unsigned)slot << 28) | ((unsigned)tile << 16) | (unsigned)stage;
}
// tile -> (row tile, column tile) with the precomputed reciprocal (exact for tile < 4096)
__host__ __device__ __forceinline__ int x3_tile_row(const BgArgs& a, int tile) {
  return (int)(((unsigned long long)(unsigned)tile * a.sk_inv_tn) >> 32);   // < 2^12 * 2^32
}
// K stages (of 16) of tile `tile` at frequency f: 0 = structurally absent (strided dgrad / wgrad)
__host__ __device__ __forceinline__ int x3_item_stages(const BgArgs& a, int f, int tile) {
  const int full = a.K / X3_SK;
  if (a.seg_mode == 0) return full;
  if (a.seg_mode == 1) {
    int w = 0;
    for (int c = 0; c < 4; ++c) w += s2_present(c, f, a.seg_skip) ? 1 : 0;
    return w * (a.seg_len / X3_SK);
  }
  const int tm = x3_tile_row(a, tile), tn = tile - tm * a.tiles_n;
  const int mask = a.sk_cmask[a.seg_mode == 2 ? tn : tm];
  bool any = false;
  for (int c = 0; c < 4; ++c) any = any || (((mask >> c) & 1) && s2_present(c, f, a.seg_skip));
  return any ? full : 0;
}
__host__ __device__ __forceinline__ void x3_slot_info(const BgArgs& a, int xcd, int slot, int& f, int& t0, int& t1) {
  const int code = (unsigned char)a.fmap[xcd][slot];
  const int tiles = a.tiles_m * a.tiles_n, half = (tiles + 1) >> 1;
  f = code & 63;
  t0 = (code & 128) ? half : 0;
  t1 = (code & 64) ? half : tiles;
}

// A workgroup's range [pstart, pstop) of its XCD's queue, walked BACKWARDS one piece at a time (device: the piece
// table of the stream kernel; host: tests/test_stream_plan_cpu.py through tools/stream_plan_check.hip).
struct X3Walk {
  unsigned end;            // everything at or after this position has been handed out
  int slot, f, t0, t1;     // the slot `end` lies in: its frequency and tile range
};
__host__ __device__ __forceinline__ void x3_walk_begin(const BgArgs& a, int xcd, unsigned pstop, X3Walk& w) {
  w.end = pstop;
  w.slot = (int)(pstop >> 28);
  x3_slot_info(a, xcd, w.slot, w.f, w.t0, w.t1);
}
// next piece (stages [s0, s1) of the S stages of (f, tile); lo = position of the tile's stage 0); false: range done
__host__ __device__ __forceinline__ bool x3_walk_next(const BgArgs& a, int xcd, unsigned pstart, X3Walk& w, int& f, int& tile,
                                                      int& s0, int& s1, int& S, unsigned& lo) {
  while (w.end > pstart) {
    tile = (int)((w.end >> 16) & 0xfffu);
    int stage = (int)(w.end & 0xffffu);
    if (stage == 0) {
      if (tile > w.t0) {
        tile -= 1;
      } else {
        --w.slot;
        x3_slot_info(a, xcd, w.slot, w.f, w.t0, w.t1);
        tile = w.t1 - 1;
      }
      w.end = x3_pos(w.slot, tile, 0);
    }
    S = x3_item_stages(a, w.f, tile);
    if (stage == 0) {
      if (S == 0) continue;
      stage = S;
    }
    lo = x3_pos(w.slot, tile, 0);
    s0 = pstart > lo ? (int)(pstart & 0xffffu) : 0;
    s1 = stage;
    f = w.f;
    w.end = lo + (unsigned)s0;
    return true;
  }
  return false;
}

// One contiguous piece of one tile's contraction as the kernel keeps it in LDS (built once per workgroup): stages
// [s0, s1) of the S stages of (f, tile), lo = queue position of the tile's stage 0, and the operand stream of the
// piece (base of the frequency's A / B planes, first row / column block of the tile, stage -> k block map).
struct X3Piece {
  const u16* opA;
  const u16* opB;
  int oA, oB;              // first row block (of 32) / column block (of 16, TL) of the tile in A / B
  int steps0, kb0, kb1;    // stage j -> k block: j < steps0 ? kb0 + j : kb1 + j (two class runs: strided forward)
  int s0, s1, S;
  int f, tile;
  unsigned lo;
  int pad;
};
static_assert(sizeof(X3Piece) == 64, "X3Piece is read as four 16-byte words");
constexpr int X3_SK_MAXPIECES = 160;
constexpr size_t X3_SK_LDS = X3_LDS + 512 + (size_t)X3_SK_MAXPIECES * sizeof(X3Piece);

template <bool TL>
__global__ __launch_bounds__(X3_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1))) void wino_bgemm_x3_stream_kernel(BgArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
  const int x = blockIdx.x, xcd = x & 7, wg = x >> 3;
  const unsigned pstart = a.sk_bound[xcd][wg], pstop = a.sk_bound[xcd][wg + 1];
  if (pstart >= pstop) return;
  X3_STAMP(0);
  int stamp_i = 2;
  (void)stamp_i;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, g = lane >> 5;

  // ---- this workgroup's pieces, last first (every thread walks the queue; thread 0 writes the table).  The first
  // piece alone is enough to start the operand stream (a piece has >= 4 stages): the rest of the table is built
  // while the first three stages are in flight.
  X3Piece* pieces = reinterpret_cast<X3Piece*>(smem3 + X3_LDS + 512);
  int npieces = 0;
  X3Walk wk;
  x3_walk_begin(a, xcd, pstop, wk);
  auto walk = [&](int limit) {
    while (npieces < limit) {
      int f, tile, s0, stage, S;
      unsigned lo;
      if (!x3_walk_next(a, xcd, pstart, wk, f, tile, s0, stage, S, lo)) break;
      X3Piece pc;
      const int tm = x3_tile_row(a, tile), tn = tile - tm * a.tiles_n;
      pc.opA = a.Ap + (long)f * a.sAp;
      pc.opB = a.Bp + (long)f * a.sBp;
      pc.oA = TL ? (tm * X3_BM) >> 4 : (tm * X3_BM) >> 5;
      pc.oB = TL ? (tn * X3_BN) >> 4 : (tn * X3_BN) >> 5;
      int lo0 = 0, len0 = a.K, lo1 = 0;
      if (a.seg_mode == 1) {
        int cc = 0, nrun = 0;
        len0 = 0;
        while (cc < 4) {
          if (!s2_present(cc, f, a.seg_skip)) {
            ++cc;
            continue;
          }
          int e = cc + 1;
          while (e < 4 && s2_present(e, f, a.seg_skip)) ++e;
          if (nrun == 0) {
            lo0 = cc * a.seg_len;
            len0 = (e - cc) * a.seg_len;
          } else {
            lo1 = cc * a.seg_len;
          }
          ++nrun;
          cc = e;
        }
      }
      pc.steps0 = len0 / X3_SK;
      pc.kb0 = lo0 / X3_SK;
      pc.kb1 = lo1 / X3_SK - pc.steps0;
      pc.s0 = s0; pc.s1 = stage; pc.S = S;
      pc.f = f; pc.tile = tile; pc.lo = lo; pc.pad = 0;
      if (tid == 0) pieces[npieces] = pc;
      ++npieces;
    }
  };
  walk(1);
  if (npieces == 0) return;   // (the planner never hands out such a range)
  __syncthreads();
  // a piece's fields as scalars (all lanes read the same words)
  auto piece_words = [&](int idx, int w) -> u32x4 {
    const u32x4 v = reinterpret_cast<const u32x4*>(pieces + idx)[w];
    u32x4 s;
#pragma unroll
    for (int e = 0; e < 4; ++e) s[e] = (unsigned)__builtin_amdgcn_readfirstlane((int)v[e]);
    return s;
  };

  // ---- operand stream (this wave: 12 chunks per stage of A (waves 0,1) or B (waves 2,3)) --------------------
  const bool isA = wave < 2;
  const int half = wave & 1;
  const long plane = isA ? a.pA : a.pB;
  const int rbmax = (isA ? a.rbA : a.rbB) - 1, cbn = isA ? a.cbA : a.cbB;
  const unsigned voff = TL ? (unsigned)(lane >> 5) * 1024u + (unsigned)(lane & 31) * 16u : (unsigned)lane * 16u;
  const unsigned lds_base = (unsigned)(unsigned long)(__attribute__((address_space(3))) void*)smem3;
  struct Cur {
    const u16* base[X3_PER_WAVE];   // per chunk stream: plane + row / column block of the tile, k block 0
    int steps0, kb0, kb1, j, s1, idx;
  };
  auto setup = [&](Cur& c, int idx) {
    const u32x4 w0 = piece_words(idx, 0), w1 = piece_words(idx, 1), w2 = piece_words(idx, 2);
    const unsigned long long pa = ((unsigned long long)w0[1] << 32) | w0[0], pb = ((unsigned long long)w0[3] << 32) | w0[2];
    const u16* opb = reinterpret_cast<const u16*>(isA ? pa : pb);
    const int o0 = (int)(isA ? w1[0] : w1[1]);
#pragma unroll
    for (int i = 0; i < X3_PER_WAVE; ++i) {
      const int li = half * X3_PER_WAVE + i;
      const int piece = li >> 3, rg = li & 7;
      if (TL) {
        int cb = o0 + 2 * rg;
        if (cb > ((cbn - 1) & ~1)) cb = (cbn - 1) & ~1;
        c.base[i] = opb + piece * plane + ((long)cb << 9);
      } else {
        int rb = o0 + rg;
        if (rb > rbmax) rb = rbmax;
        c.base[i] = opb + piece * plane + (((long)rb * a.kblocks) << 9);
      }
    }
    c.steps0 = (int)w1[2];
    c.kb0 = (int)w1[3];
    c.kb1 = (int)w2[0];
    c.j = (int)w2[1];
    c.s1 = (int)w2[2];
    c.idx = idx;
  };
  auto issue = [&](const Cur& c, int buf, int i0, int n) {
    const long kb = c.j < c.steps0 ? c.kb0 + c.j : c.kb1 + c.j;
    const long koff = TL ? (((kb >> 1) * cbn) << 9) + ((kb & 1) << 8) : kb << 9;
#pragma unroll
    for (int i = i0; i < i0 + n; ++i) {
      const int li = half * X3_PER_WAVE + i;
      const int piece = li >> 3, rg = li & 7;
      const u16* src = c.base[i] + koff;
      const unsigned dst = lds_base + buf * X3_STAGE + (isA ? 0 : X3_NP * X3_TA) + piece * X3_TA + rg * 1024;
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(src), "s"(dst) : "memory");
    }
  };
  // next stage of the stream; past the end of the range the last stage is fetched again (into a free buffer, never
  // read): the number of loads in flight stays the same for every stage
  auto advance = [&](Cur& c) {
    if (++c.j == c.s1) {
      if (c.idx + 1 < npieces) setup(c, c.idx + 1);
      else c.j = c.s1 - 1;
    }
  };

  f32x16 acc[X3_MT][X3_NT];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < X3_MT; ++i)
#pragma unroll
      for (int j = 0; j < X3_NT; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
  };
  const int sw = (r >> 3) & 1;
  const int g4 = lane >> 4, l16 = lane & 15;
  const int ftl = (g4 & 1) * 512 + (8 * (g4 >> 1) + (l16 >> 2)) * 32 + ((((l16 >> 1) & 1) ^ (g4 >> 1)) * 16) + (l16 & 1) * 8;
  const int fa = TL ? wm * 4096 + ftl : (wm * X3_MT * 32 + r) * 32 + 16 * (g ^ sw);
  const int fb = X3_NP * X3_TA + (TL ? wn * 4096 + ftl : (wn * X3_NT * 32 + r) * 32 + 16 * (g ^ sw));
  auto read_frag = [&](const unsigned char* p) -> x3frag_t {
    if (TL) {
      typedef short s16x4 __attribute__((ext_vector_type(4)));
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 128));
      return __builtin_bit_cast(x3frag_t, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    } else {
      return *reinterpret_cast<const x3frag_t*>(p);
    }
  };
  // one stage (see the one-tile kernel): the stage after this one has landed (vmcnt + barrier; `landed`: known
  // without waiting), the buffer this stage was read from is refilled three stages ahead, 96 MFMAs on F with the
  // 24 fragment reads of the next stage (into G) in between
  auto stage = [&](Cur& c, int bufn, const X3Frags& F, X3Frags& G, bool landed) {
    if (!landed) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(X3_PER_WAVE) : "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (fragment reads returned before their buffer is refilled)
    __builtin_amdgcn_s_barrier();
    const int rbuf = bufn == 0 ? X3_NSTAGE - 1 : bufn - 1;
    const unsigned char* pa = smem3 + bufn * X3_STAGE + fa;
    const unsigned char* pb = smem3 + bufn * X3_STAGE + fb;
    constexpr int NR = 8 * X3_NP;
    x3_static_for<X3_NTERM>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      constexpr int l0 = t * X3_PER_WAVE / X3_NTERM, l1 = (t + 1) * X3_PER_WAVE / X3_NTERM;
      constexpr int q0 = t * NR / X3_NTERM, q1 = (t + 1) * NR / X3_NTERM;
      issue(c, rbuf, l0, l1 - l0);
#pragma unroll
      for (int q = q0; q < q1; ++q) {
        const int qq = q % (4 * X3_NP), tt = qq / X3_NP, p = qq % X3_NP;
        if (q < 4 * X3_NP) G.a[tt][p] = read_frag(pa + p * X3_TA + tt * 1024);
        else G.b[tt][p] = read_frag(pb + p * X3_TB + tt * 1024);
      }
#pragma unroll
      for (int i = 0; i < X3_MT; ++i)
#pragma unroll
        for (int j = 0; j < X3_NT; ++j) acc[i][j] = X3_MFMA(F.a[i][X3_PA[t]], F.b[j][X3_PB[t]], acc[i][j]);
      x3_sched_group<TL ? 2 : 1, q1 - q0>();
    });
    advance(c);
  };

  // ---- what happens to a finished piece ------------------------------------------------------------------
  // Sums of a tile whose contraction is shared are chained through the workspace: a workgroup adds the parked
  // sums of the (nearest non-empty) workgroup below it to its own on the way out -- into C when the tile's
  // last stage is here, else into its own slot.  The accumulators themselves are only ever read here.
  unsigned long long* flags = reinterpret_cast<unsigned long long*>(a.sk_partial + (size_t)gridDim.x * X3_SK_SLOT);
  auto finish = [&](int idx) {
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    const u32x4 w2 = piece_words(idx, 2), w3 = piece_words(idx, 3);
    const int p_s0 = (int)w2[1], p_s1 = (int)w2[2], p_S = (int)w2[3], p_f = (int)w3[0], p_tile = (int)w3[1];
    const bool has_prev = p_s0 > 0;
    const int my_xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 15;   // HW_REG_XCC_ID
    bool poison = false;
    // Parked sums stay inside one XCD (workgroup x runs on XCD x % 8 -- tools/ablate/xcc_probe.hip -- and shares
    // tiles only with x - 8 and x + 8), whose L2 both workgroups write through and read from: no cache writeback is
    // needed (a device-scope release fence writes back the whole L2, C tiles and all: 50 - 90 k cycles per parked
    // tile; device-scope loads / stores: slower still).  Producer: plain stores, vmcnt(0) in every thread, barrier,
    // flag = epoch + its XCD.  Consumer: flag seen (and the XCD is its own, else the tile is poisoned: loud, not
    // subtly wrong), barrier, L1 invalidate, plain loads.
    const f32x4v* Pp = nullptr;
    if (has_prev) {
      int k = wg - 1;
      while (k > 0 && a.sk_bound[xcd][k] >= a.sk_bound[xcd][k + 1]) --k;
      const int xk = xcd + 8 * k;
      if (tid == 0) {
        int spins = 0;
        unsigned long long fl;
        while (((fl = __hip_atomic_load(&flags[xk], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & ~15ull) != a.sk_epoch) {
          __builtin_amdgcn_s_sleep(8);
          if (++spins > (1 << 21)) break;   // ~ 1 s: never on a healthy device; a poisoned tile instead of a hung one
        }
        __hip_atomic_store(&flags[xk], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (fl != a.sk_epoch + (unsigned long long)my_xcc) poison = true;
      }
      __syncthreads();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      Pp = reinterpret_cast<const f32x4v*>(a.sk_partial + (size_t)xk * X3_SK_SLOT) + (size_t)wave * 4096 + lane;
    }
    if (p_s1 < p_S) {
      // the tile's contraction continues in the next workgroup: park the sums (register order, 16 bytes per lane)
      f32x4v* P = reinterpret_cast<f32x4v*>(a.sk_partial + (size_t)x * X3_SK_SLOT) + (size_t)wave * 4096 + lane;
#pragma unroll
      for (int i = 0; i < X3_MT; ++i)
#pragma unroll
        for (int j = 0; j < X3_NT; ++j)
#pragma unroll
          for (int qh = 0; qh < 4; ++qh) {
            const int e = ((i * X3_NT + j) * 4 + qh) * 64;
            f32x4v v = {acc[i][j][4 * qh], acc[i][j][4 * qh + 1], acc[i][j][4 * qh + 2], acc[i][j][4 * qh + 3]};
            if (has_prev) v += Pp[e];
            if (poison) v[0] = __builtin_nanf("");
            P[e] = v;
          }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0)
        __hip_atomic_store(&flags[x], a.sk_epoch + (unsigned long long)my_xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    const int tm = x3_tile_row(a, p_tile), tn = p_tile - tm * a.tiles_n;
    const int m0 = tm * X3_BM, n0 = tn * X3_BN;
    float* C = a.C + (long)p_f * a.sC;
    const float es = a.hdrA ? x3_out_scale(a, p_f) : 1.f;   // a power of two: exact
    if (m0 + X3_BM <= a.M && n0 + X3_BN <= a.N) {
      const long ld = a.ldc;
      float* row = C + (long)(m0 + wm * X3_MT * 32 + 4 * g) * ld + (n0 + wn * X3_NT * 32 + r);
      if (!has_prev) {
        // the common case, kept as lean as the one-tile kernel's write-out: one running row pointer, immediate offsets
#pragma unroll
        for (int i = 0; i < X3_MT; ++i) {
#pragma unroll
          for (int qh = 0; qh < 4; ++qh) {
            // (a fence for the scheduler: left alone it copies the whole accumulator tile to vector registers first,
            // spills a third of the copies to scratch and then waits on the reloads with the stores in flight)
            __builtin_amdgcn_sched_barrier(0);
            float* q = row + (long)(i * 32 + 8 * qh) * ld;
#pragma unroll
            for (int ql = 0; ql < 4; ++ql) {
#pragma unroll
              for (int j = 0; j < X3_NT; ++j) q[j * 32] = acc[i][j][4 * qh + ql] * es;
              q += ld;
            }
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < X3_MT; ++i) {
#pragma unroll
          for (int qh = 0; qh < 4; ++qh) {
            __builtin_amdgcn_sched_barrier(0);
            f32x4v pv[X3_NT];
#pragma unroll
            for (int j = 0; j < X3_NT; ++j) pv[j] = Pp[((i * X3_NT + j) * 4 + qh) * 64];
            if (poison) pv[0][0] = __builtin_nanf("");
            float* q = row + (long)(i * 32 + 8 * qh) * ld;
#pragma unroll
            for (int ql = 0; ql < 4; ++ql) {
#pragma unroll
              for (int j = 0; j < X3_NT; ++j) q[j * 32] = (acc[i][j][4 * qh + ql] + pv[j][ql]) * es;
              q += ld;
            }
          }
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < X3_MT; ++i)
#pragma unroll
        for (int j = 0; j < X3_NT; ++j)
#pragma unroll
          for (int qh = 0; qh < 4; ++qh) {
            f32x4v pv = {0.f, 0.f, 0.f, 0.f};
            if (has_prev) pv = Pp[((i * X3_NT + j) * 4 + qh) * 64];
#pragma unroll
            for (int ql = 0; ql < 4; ++ql) {
              const int m = m0 + (wm * X3_MT + i) * 32 + ql + 8 * qh + 4 * g;
              const int n = n0 + (wn * X3_NT + j) * 32 + r;
              if (m < a.M && n < a.N) C[(long)m * a.ldc + n] = (acc[i][j][4 * qh + ql] + pv[ql]) * es;
            }
          }
      X3_WAIT_VM(0);   // (a wave may have issued fewer than 64 stores here)
    }
  };

  // ---- the stream -------------------------------------------------------------------------------------------
  Cur ic;
  setup(ic, 0);
  issue(ic, 0, 0, X3_PER_WAVE);
  ++ic.j;
  issue(ic, 1, 0, X3_PER_WAVE);
  ++ic.j;
  issue(ic, 2, 0, X3_PER_WAVE);
  walk(X3_SK_MAXPIECES);
  npieces = __builtin_amdgcn_readfirstlane(npieces);
  __syncthreads();
  advance(ic);
  X3_WAIT_VM(2 * X3_PER_WAVE);
  __builtin_amdgcn_s_barrier();
  X3Frags F0, F1;
  auto load_frags = [&](X3Frags& F, int buf) {
    const unsigned char* pa = smem3 + buf * X3_STAGE + fa;
    const unsigned char* pb = smem3 + buf * X3_STAGE + fb;
#pragma unroll
    for (int t = 0; t < X3_MT; ++t)
#pragma unroll
      for (int p = 0; p < X3_NP; ++p) F.a[t][p] = read_frag(pa + p * X3_TA + t * 1024);
#pragma unroll
    for (int t = 0; t < X3_NT; ++t)
#pragma unroll
      for (int p = 0; p < X3_NP; ++p) F.b[t][p] = read_frag(pb + p * X3_TB + t * 1024);
  };
  int bufn = 1;           // buffer of the stage after the one about to be computed
  bool landed = false;
  for (int ci = 0; ci < npieces; ++ci) {
    // one piece: the accumulators start from zero and the first fragments come out of LDS here (kept across the
    // write-out of the previous piece they would cost 96 registers; their buffer is refilled only after the next
    // barrier), then pairs of stages
    const u32x4 w2 = piece_words(ci, 2);
    zero_acc();
    load_frags(F0, bufn == 0 ? X3_NSTAGE - 1 : bufn - 1);
    X3_STAMP(stamp_i);
    for (int n = ((int)w2[2] - (int)w2[1]) >> 1; n > 0; --n) {
      stage(ic, bufn, F0, F1, landed);
      bufn = bufn == X3_NSTAGE - 1 ? 0 : bufn + 1;
      stage(ic, bufn, F1, F0, landed);
      bufn = bufn == X3_NSTAGE - 1 ? 0 : bufn + 1;
      landed = false;
    }
    X3_STAMP(stamp_i + 1);
    finish(ci);
    X3_STAMP(stamp_i + 2);
    // Loads and stores retire in issue order: once all but 63 of the operations issued so far have retired, the
    // two stages that were in flight before the >= 64 stores of the write-out are in LDS, and the stores drain
    // under the next two stages.
    X3_WAIT_VM(63);
    X3_STAMP(stamp_i + 3);
#ifdef X3_TIMING
    stamp_i += 4;
#endif
    landed = true;
  }
  X3_WAIT_VM(0);   // the refetched last stages are still landing in LDS
#ifdef X3_TIMING
  X3_STAMP(1);
  __syncthreads();
  if (threadIdx.x < 64) a.dbg[(long)blockIdx.x * 64 + threadIdx.x] = reinterpret_cast<unsigned long long*>(smem3 + X3_LDS)[threadIdx.x];
#endif
}

// Host: cut every XCD's queue (BgArgs::fmap, set by build_fmap) into nw ranges of equal K-stage counts.  False when
// the shape does not fit the position encoding (the caller then launches the one-tile kernel).
inline bool x3_plan_stream(BgArgs& b, int nw, int stagger = 0) {
  const int tiles = b.tiles_m * b.tiles_n;
  if (nw < 1 || nw > X3_SK_MAXW || tiles >= 4096 || b.K / X3_SK >= 65536 || b.K % X3_BK) return false;
  if (b.seg_mode == 1 && b.seg_len % X3_BK) return false;
  b.sk_inv_tn = (0x100000000ull + (unsigned)b.tiles_n - 1) / (unsigned)b.tiles_n;
  memset(b.sk_cmask, 0, sizeof(b.sk_cmask));
  if (b.seg_mode == 2 || b.seg_mode == 3) {
    const int nt = b.seg_mode == 2 ? b.tiles_n : b.tiles_m, ext = b.seg_mode == 2 ? b.N : b.M;
    if (nt > 64) return false;
    for (int t = 0; t < nt; ++t) {
      const int lo = t * 256;
      int hi = lo + 255;
      if (hi >= ext) hi = ext - 1;
      for (int c = lo / b.seg_len; c <= hi / b.seg_len && c < 4; ++c) b.sk_cmask[t] |= (unsigned char)(1 << c);
    }
  }
  struct Item {
    int slot, tile, S;
  };
  static thread_local Item items[8 * 4096];
  for (int xcd = 0; xcd < 8; ++xcd) {
    int n = 0, last_slot = 0, last_t1 = 0;
    long U = 0;
    for (int slot = 0; slot < 8 && b.fmap[xcd][slot] != -1; ++slot) {
      int f, t0, t1;
      x3_slot_info(b, xcd, slot, f, t0, t1);
      for (int t = t0; t < t1; ++t) {
        const int S = x3_item_stages(b, f, t);
        if ((S & 1) || (S > 0 && S < 4)) return false;   // (the kernel starts three stages deep inside its first piece)
        items[n++] = Item{slot, t, S};
        U += S;
      }
      last_slot = slot;
      last_t1 = t1;
    }
    unsigned* bd = b.sk_bound[xcd];
    if (n == 0 || U == 0) {
      for (int c = 0; c <= X3_SK_MAXW; ++c) bd[c] = 0;
      continue;
    }
    const unsigned stop = x3_pos(last_slot, last_t1, 0);
    bd[0] = x3_pos(items[0].slot, items[0].tile, 0);
    int idx = 0;
    long pre = 0;   // stages before items[idx]
    for (int c = 1; c < nw; ++c) {
      long target = U * c / nw;
      // stagger: when whole tiles would line up on the cuts, every compute unit writes its C tile at the same
      // moment; shifting cut c by (c mod 4) quarter tiles spreads the writes (work per workgroup: +- a quarter tile)
      if (stagger && idx < n) target += (long)(c & 3) * (items[idx].S / 4);
      if (target > U) target = U;
      target &= ~1L;
      while (idx < n && pre + items[idx].S <= target) pre += items[idx++].S;
      unsigned pos;
      if (idx >= n) {
        pos = stop;
      } else {
        long off = target - pre;
        // no slivers: a cut closer than four stages to a tile boundary moves onto it
        if (off < 4) {
          off = 0;
        } else if (items[idx].S - off < 4) {
          pre += items[idx++].S;
          off = 0;
        }
        // (a cut in front of structurally absent tiles moves behind them: a range never holds absent tiles only)
        if (off == 0)
          while (idx < n && items[idx].S == 0) ++idx;
        pos = idx >= n ? stop : x3_pos(items[idx].slot, items[idx].tile, (int)off);
      }
      // (two cuts in one tile: at least four stages apart, else the range between them is empty)
      if (pos < bd[c - 1] || ((pos >> 16) == (bd[c - 1] >> 16) && pos - bd[c - 1] < 4)) pos = bd[c - 1];
      bd[c] = pos;
    }
    for (int c = nw; c <= X3_SK_MAXW; ++c) bd[c] = stop;
    // pieces per workgroup must fit the kernel's table
    for (int c = 0; c < nw; ++c) {
      if (bd[c] >= bd[c + 1]) continue;
      X3Walk w;
      x3_walk_begin(b, xcd, bd[c + 1], w);
      int cnt = 0, f, tile, s0, s1, S;
      unsigned lo;
      while (x3_walk_next(b, xcd, bd[c], w, f, tile, s0, s1, S, lo))
        if (++cnt > X3_SK_MAXPIECES) return false;
    }
  }
  b.sk_nw = nw;
  return true;
}

}  // namespace
