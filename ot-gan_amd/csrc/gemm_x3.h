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
};


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
constexpr int X3_STAGE = 3 * (X3_TA + X3_TB);
constexpr size_t X3_LDS = (size_t)X3_NSTAGE * X3_STAGE;
constexpr int X3_PER_WAVE = 3 * (X3_BM + X3_BN) / 32 / 4;              // global_load_lds per wave per stage

struct X3Frags {
  bf16x8 a[X3_MT][3];
  bf16x8 b[X3_NT][3];
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
        if (cb > cbn - 2) cb = cbn - 2;   // past the operand's last column: duplicates feed C rows / columns never stored
        src = opb + piece * plane + ((((long)(kb >> 1) * cbn + cb) << 9) + ((kb & 1) << 8));
      } else {
        int rb = rb0 + rg;
        if (rb > rbmax) rb = rbmax;
        src = opb + piece * plane + (((long)rb * a.kblocks + kb) << 9);
      }
      const unsigned dst = lds_base + buf * X3_STAGE + (isA ? 0 : 3 * X3_TA) + piece * X3_TA + rg * 1024;
      // scalar base + per-lane 32-bit offset (the builtin expands to 64-bit per-lane addresses inside the loop);
      // M0 = LDS address of the chunk.  Nothing else in this kernel uses M0.
      asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(src), "s"(dst) : "memory");
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
  const int fb = 3 * X3_TA + (TL ? wn * 4096 + ftl : (wn * X3_NT * 32 + r) * 32 + 16 * (g ^ sw));
  auto read_frag = [&](const unsigned char* p) -> bf16x8 {
    if (TL) {
      typedef short s16x4 __attribute__((ext_vector_type(4)));
      typedef short s16x8 __attribute__((ext_vector_type(8)));
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 128));
      return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    } else {
      return *reinterpret_cast<const bf16x8*>(p);
    }
  };
  auto load_frags = [&](X3Frags& F, int buf) {
    const unsigned char* pa = smem3 + buf * X3_STAGE + fa;
    const unsigned char* pb = smem3 + buf * X3_STAGE + fb;
#pragma unroll
    for (int t = 0; t < X3_MT; ++t)
#pragma unroll
      for (int p = 0; p < 3; ++p) F.a[t][p] = read_frag(pa + p * X3_TA + t * 1024);
#pragma unroll
    for (int t = 0; t < X3_NT; ++t)
#pragma unroll
      for (int p = 0; p < 3; ++p) F.b[t][p] = read_frag(pb + p * X3_TB + t * 1024);
  };
  // six products per fp32-exact product, smallest terms first; consecutive MFMAs hit different accumulators
  auto mfmas = [&](const X3Frags& F) {
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int i = 0; i < X3_MT; ++i)
#pragma unroll
        for (int j = 0; j < X3_NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.a[i][PA[t]], F.b[j][PB[t]], acc[i][j], 0, 0, 0);
  };
  if (PIPE) {
    // nst is even and >= 4 (host contract).  One stage: stage st+1 has landed (barrier), the buffer stage st
    // was read from is refilled with stage st+3, then the 96 MFMAs of stage st on F with the 24 fragment
    // reads of stage st+1 (into G) in between.  ISSUE: stage st+3 exists; PEND: stage st+2 is in flight;
    // LOAD: stage st+1 exists.
    auto stage = [&](int st, int bufn, const X3Frags& F, X3Frags& G, auto issue_c, auto pend_c, auto load_c) {
      constexpr bool ISSUE = decltype(issue_c)::value, PEND = decltype(pend_c)::value, LOAD = decltype(load_c)::value;
      if (PEND) X3_WAIT_VM(X3_PER_WAVE);
      else X3_WAIT_VM(0);
      __builtin_amdgcn_s_barrier();
      // six term groups of 16 MFMAs; in front of each: two of the twelve refill loads (all twelve at once keep the
      // wave in its VMEM issue queue for several hundred cycles while the matrix pipe drains) and four of the 24
      // fragment reads of the next stage
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
      const int rbuf = bufn == 0 ? X3_NSTAGE - 1 : bufn - 1;
      const unsigned char* pa = smem3 + bufn * X3_STAGE + fa;
      const unsigned char* pb = smem3 + bufn * X3_STAGE + fb;
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        if (ISSUE) issue(st + 3, rbuf, 2 * t, 2);
        if (LOAD) {
#pragma unroll
          for (int q = 4 * t; q < 4 * t + 4; ++q) {
            const int tt = (q % 12) / 3, p = q % 3;   // q < 12: A fragments, else B
            if (q < 12) G.a[tt][p] = read_frag(pa + p * X3_TA + tt * 1024);
            else G.b[tt][p] = read_frag(pb + p * X3_TB + tt * 1024);
          }
        }
#pragma unroll
        for (int i = 0; i < X3_MT; ++i)
#pragma unroll
          for (int j = 0; j < X3_NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.a[i][PA[t]], F.b[j][PB[t]], acc[i][j], 0, 0, 0);
        if (LOAD) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x100, TL ? 2 : 1, 0);   // the DS reads of one fragment
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);            // four MFMAs
          }
        }
      }
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
    int st = 0, bufn = 1;   // bufn = buffer of stage st + 1
    auto next = [&]() { bufn = bufn == X3_NSTAGE - 1 ? 0 : bufn + 1; };
    for (; st + 6 <= nst; st += 2) {
      stage(st, bufn, F0, F1, Y{}, Y{}, Y{});
      next();
      stage(st + 1, bufn, F1, F0, Y{}, Y{}, Y{});
      next();
    }
    stage(st, bufn, F0, F1, Y{}, Y{}, Y{});
    next();
    stage(st + 1, bufn, F1, F0, N{}, Y{}, Y{});
    next();
    stage(st + 2, bufn, F0, F1, N{}, N{}, Y{});
    next();
    stage(st + 3, bufn, F1, F0, N{}, N{}, N{});
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
  const float es = a.epi ? a.epi_scale : 1.f, eb = a.epi ? a.epi_bias : 0.f, ed = a.epi ? a.epi_diag[f & 7] : 0.f;
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
          if (a.epi) v = fmaf(es, v, eb) + (m == n ? ed : 0.f);
          C[(long)m * a.ldc + n] = v;
        }
      }
}

}  // namespace
