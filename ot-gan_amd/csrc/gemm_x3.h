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
#ifdef X3_STREAM_TOOL
  // Stream kernel (tools/ablate/gemm_x3_stream.h; not part of the library since round 5): one persistent workgroup per
  // compute unit; workgroup c of XCD x owns the positions [sk_bound[x][c], sk_bound[x][c + 1]) of that XCD's queue.
  unsigned sk_bound[8][41];
  int sk_nw;                          // workgroups per XCD (grid = 8 sk_nw)
  float* sk_partial;                  // [grid] parked 256 x 256 accumulator tiles, then [grid] 64-bit flags
  unsigned long long sk_epoch;        // flag value of this launch
  unsigned long long sk_inv_tn;       // ceil(2^32 / tiles_n): tile / tiles_n without a division on the device
  unsigned char sk_cmask[64];         // strided dgrad / wgrad: parity classes a column / row tile overlaps
#endif
#ifdef X3_TIMING
  unsigned long long* dbg;   // tools/ablate/x3_phase.hip: s_memtime stamps of workgroup x at dbg[64 x ..]
#endif
  unsigned x_total;          // 256 x 128 kernel: number of queue positions (= gridDim.x)
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
// One tile per workgroup.  (Round 4 tried a tile loop over x, x + gridDim.x, ... with the grid capped at the resident
// workgroups and the next tile's first three stages issued before the write-out: bit-identical, 1 % slower on the DCGAN
// step -- the dispatcher's dynamic placement of short workgroups beats a static walk -- and 110 spilled SGPRs of tile-setup
// state; removed in round 5, DESIGN section 3 "Measured and dropped".)
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
  if (blockIdx.x >= a.x_total || !setup((int)blockIdx.x, T)) return;
  issue(T, 0, 0);
  issue(T, 1, 1);
  issue(T, 2, 2);
  {
#pragma unroll
    for (int i = 0; i < X3_MT; ++i)
#pragma unroll
      for (int j = 0; j < X3N_NT; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    X3NFrags F0, F1;
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

    const int f = T.f, m0 = T.m0, n0 = T.n0;
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

}  // namespace
