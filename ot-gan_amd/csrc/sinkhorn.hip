// sinkhorn.hip -- the OT-GAN matching block on gfx950:
//   cost Gram blocks (fp32 MFMA, split-K)  ->  log-domain Sinkhorn (potential form, exact
//   iteration count)  ->  plan application (fp32 MFMA)  ->  distance (fp64 accumulation).
// Replaces reference utils/matching.py:11-153 and toy_example/matching_cpu.py:4-164.
#include <stdlib.h>
#include <atomic>

#include "gemm_tile.h"
// the matching GEMMs of N >= 256 on two scaled fp16 pieces (round 4; three bf16 pieces until then): section 5
#define X3_PIECES 2
#include "gemm_x3.h"
#include "../../include/otgan.h"

namespace {

using SCfg = GemmCfg<2, 2, 2, 2, 16>;  // 128x128 block tile, 4 waves (2x2), 64x64 per wave
constexpr int kMaxProb = 6;
constexpr float kNegBig = -1.0e30f;  // masked log-kernel entry (exp -> 0, never NaN)

// ======================================================================================
// 1. cost matrices:  dot = X . Y^T  (split-K partials)  ->  K = -lambda * cost
// ======================================================================================
struct CostArgs {
  const float* X[kMaxProb];
  const float* Y[kMaxProb];
  int P, n, m, D;
  long ldf;
  int kt_per_split;
  float* ws;  // [nsplit][P][n][m]
  // FUSE (one K split): the epilogue writes K = -lambda * cost directly, no partials, no finish kernel
  float* K;   // [P][n][m]
  float lambda, inv_d;
  int cost_kind;
  const float* xsq[kMaxProb];
  const float* ysq[kMaxProb];
  float diag[kMaxProb];
};

template <bool VEC, bool FUSE>
__global__ __launch_bounds__(256) void cost_partial_kernel(CostArgs a) {
  using LA = MatLoaderK<SCfg, 128, VEC>;
  using LB = MatLoaderK<SCfg, 128, VEC>;
  __shared__ __attribute__((aligned(16))) float smem[2 * LA::FLOATS + 2 * LB::FLOATS];
  const int tn = (a.m + 127) / 128;
  const int tmi = blockIdx.x / tn, tni = blockIdx.x % tn;
  const int split = blockIdx.y, p = blockIdx.z;
  const int nkt_total = (a.D + SCfg::BK - 1) / SCfg::BK;
  const int kt0 = split * a.kt_per_split;
  int nkt = nkt_total - kt0;
  if (nkt > a.kt_per_split) nkt = a.kt_per_split;
  const int k0 = kt0 * SCfg::BK;
  LA la;
  LB lb;
  la.init(a.X[p] + (long)tmi * 128 * a.ldf + k0, a.ldf, a.n - tmi * 128, a.D - k0);
  lb.init(a.Y[p] + (long)tni * 128 * a.ldf + k0, a.ldf, a.m - tni * 128, a.D - k0);
  typename SCfg::acc_t acc[SCfg::MT][SCfg::NT];
  zero_acc<SCfg>(acc);
  gemm_mainloop<SCfg>(la, lb, nkt, smem, acc);
  const int n = a.n, m = a.m;
  if (FUSE) {
    float* out = a.K + (long)p * n * m;
    const float lam = a.lambda, inv_d = a.inv_d, dg = a.diag[p];
    const bool cosine = a.cost_kind == OTGAN_COST_COSINE;
    const float* xs = a.xsq[p];
    const float* ys = a.ysq[p];
    foreach_acc<SCfg>(acc, [&](int r, int c, int, int, int, float v) {
      const int row = tmi * 128 + r, col = tni * 128 + c;
      if (row < n && col < m) {
        float cst = cosine ? 1.f - v : xs[row] + ys[col] - v * inv_d;
        if (row == col) cst += dg;
        out[(long)row * m + col] = -lam * cst;
      }
    });
  } else {
    float* out = a.ws + ((long)split * a.P + p) * n * m;
    foreach_acc<SCfg>(acc, [&](int r, int c, int, int, int, float v) {
      const int row = tmi * 128 + r, col = tni * 128 + c;
      if (row < n && col < m) out[(long)row * m + col] = v;
    });
  }
}

// The same Gram blocks on the fp16 matrix pipe (round 5; cosine cost, the N <= 128 problems of a single-GPU step --
// VERDICT r4 item 6).  The fp32 instruction above spends 64 matrix cycles per 32 x 32 x 2 products; here every float4 is
// split ONCE while it is staged, x * 2^13 = hi + lo (two fp16 pieces, 22 significand bits: the arithmetic of the N >= 256
// matching GEMMs and of the convolutions, 5.6e-6 against 5.9e-6 for the fp32 chain on the injected gradients,
// tests/test_matching_engine_accuracy_gpu.py), and a 16-wide k slab of a 32 x 32 tile costs three v_mfma_f32_32x32x16_f16
// = 96 cycles instead of 512.  The kernel is then bound by how many feature bytes a compute unit keeps in flight, not by
// the matrix pipe: registers hold the float4s of TWO k steps ahead of the one in LDS (gemm_mainloop_x2h keeps one).
// The scale is a priori (cosine features are rows of unit length: |x| <= 1; anything below 8 is representable, beyond it
// the pieces overflow to infinity and the result is NaN -- loud; OTGAN_MATCH_FP32=1 keeps the exact-fp32 kernel).
// Requires 16-byte aligned rows and D % 4 == 0 (launch_cost checks).
constexpr float kCostH2Scale = 8192.f;              // 2^13
struct CostH2Loader {
  const float* base;   // row 0, k0
  long ld;
  int rlast, kdim, klast;
  // rows: valid rows (rows past them re-read the last one: they only feed accumulators that are never stored);
  // kdim: k (relative to k0) from which the split contributes zeros; kmax: the row's remaining floats (a multiple of 4)
  __device__ __forceinline__ void init(const float* b, long ld_, int rows, int kdim_, int kmax) {
    base = b; ld = ld_; rlast = rows - 1; kdim = kdim_; klast = kmax - 4;
  }
  __device__ __forceinline__ void load(int kt, float4 (&reg)[2]) const {
    const int c = threadIdx.x & 3, r0 = threadIdx.x >> 2;
    const int k = kt * 16 + 4 * c;
    const int kc = k < klast ? k : klast;     // branch-free: a clamped address, then a select
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      int r = r0 + 64 * p;
      r = r < rlast ? r : rlast;
      reg[p] = *reinterpret_cast<const float4*>(base + (long)r * ld + kc);
    }
  }
  // (the select lives here, not in load(): anything that touches the loaded value waits for it)
  __device__ __forceinline__ void store2(unsigned char* t, int kt, const float4 (&reg)[2]) const {
    const int c = threadIdx.x & 3, r0 = threadIdx.x >> 2;
    const float sc = (kt * 16 + 4 * c < kdim) ? kCostH2Scale : 0.f;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      float4 v = reg[p];
      v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
      x2h_store4(t, 128 * kX3sRowBytes, r0 + 64 * p, 4 * c, v);
    }
  }
};

template <bool FUSE>
__global__ __launch_bounds__(256) void cost128_h2_kernel(CostArgs a) {
  using L = X2hLds<SCfg>;
  __shared__ __attribute__((aligned(16))) unsigned char smem[L::BYTES];
  unsigned char* sA = smem;
  unsigned char* sB = smem + 2 * L::TA;
  // workgroup x runs on XCD x % 8 (round-robin placement): the P problems of one K split take adjacent slots of ONE XCD, so
  // the feature blocks they share (every block is an operand of three of the six problems) meet in that XCD's L2
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int split = (slot / a.P) * 8 + xcd, p = slot % a.P;
  const int nkt_total = (a.D + 15) / 16;
  const int kt0 = split * a.kt_per_split;
  int nkt = nkt_total - kt0;
  if (nkt <= 0) return;
  if (nkt > a.kt_per_split) nkt = a.kt_per_split;
  const int k0 = kt0 * 16;
  CostH2Loader la, lb;
  {
    const int kmax = a.D - k0, kd = nkt * 16 < kmax ? nkt * 16 : kmax;
    la.init(a.X[p] + k0, a.ldf, a.n, kd, kmax);
    lb.init(a.Y[p] + k0, a.ldf, a.m, kd, kmax);
  }
  typename SCfg::acc_t acc[2][2];
  zero_acc<SCfg>(acc);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  int a_off[2], b_off[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    a_off[t] = x3s_off((wm * 2 + t) * 32 + li, 8 * lh);
    b_off[t] = x3s_off((wn * 2 + t) * 32 + li, 8 * lh);
  }
  float4 ra0[2], ra1[2], rb0[2], rb1[2];   // two register sets: k steps kt + 1 and kt + 2 in flight while step kt is multiplied
  auto mma = [&](int cur) {
    const unsigned char* pa = sA + cur * L::TA;
    const unsigned char* pb = sB + cur * L::TB;
    gt_f16x8 fa[2][2], fb[2][2];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        fa[q][t] = *reinterpret_cast<const gt_f16x8*>(pa + q * L::PA + a_off[t]);
        fb[q][t] = *reinterpret_cast<const gt_f16x8*>(pb + q * L::PB + b_off[t]);
      }
#pragma unroll
    for (int term = 0; term < 3; ++term) {   // lo*hi, hi*lo, hi*hi (smallest first)
      constexpr int pa_of[3] = {1, 0, 0}, pb_of[3] = {0, 1, 0};
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[pa_of[term]][mt], fb[pb_of[term]][nt], acc[mt][nt], 0, 0, 0);
    }
  };
  // No conditionals in the loop (loads past the split's range are clamped and selected to zero, stores past it write a
  // buffer nobody reads): an odd step count runs one all-zero step more.
  la.load(0, ra0); lb.load(0, rb0);
  la.load(1, ra1); lb.load(1, rb1);
  la.store2(sA, 0, ra0);
  lb.store2(sB, 0, rb0);
  __syncthreads();
  // hipcc hoists the conversion of a register set (pure arithmetic: no chain to a scheduling barrier) to the top of the
  // trip, where it waits for loads issued a moment ago; an empty asm that "rewrites" the set pins its first use
#define COST_H2_PIN(R)                                                                                        \
  asm volatile("" : "+v"(*reinterpret_cast<gt_f32x4*>(&R[0])), "+v"(*reinterpret_cast<gt_f32x4*>(&R[1])))   /* 128-bit operands: the loads' register tuples stay whole */
  for (int kt = 0; kt < nkt; kt += 2) {
    la.load(kt + 2, ra0); lb.load(kt + 2, rb0);
    __builtin_amdgcn_sched_barrier(0);        // (the loads stay in front of the matrix work: hipcc sinks them behind it otherwise)
    mma(0);
    __builtin_amdgcn_sched_barrier(0);
    COST_H2_PIN(ra1); COST_H2_PIN(rb1);
    la.store2(sA + L::TA, kt + 1, ra1);
    lb.store2(sB + L::TB, kt + 1, rb1);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);        // (nor does the other set's conversion move up here: it would wait for its loads)
    la.load(kt + 3, ra1); lb.load(kt + 3, rb1);
    __builtin_amdgcn_sched_barrier(0);
    mma(1);
    __builtin_amdgcn_sched_barrier(0);
    COST_H2_PIN(ra0); COST_H2_PIN(rb0);
    la.store2(sA, kt + 2, ra0);
    lb.store2(sB, kt + 2, rb0);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
  }
#undef COST_H2_PIN
  const int n = a.n, m = a.m;
  constexpr float inv = 1.f / (kCostH2Scale * kCostH2Scale);
  if (FUSE) {
    float* out = a.K + (long)p * n * m;
    const float lam = a.lambda, dg = a.diag[p];
    foreach_acc<SCfg>(acc, [&](int r, int c, int, int, int, float v) {
      if (r < n && c < m) {
        float cst = 1.f - v * inv;
        if (r == c) cst += dg;
        out[(long)r * m + c] = -lam * cst;
      }
    });
  } else {
    float* out = a.ws + ((long)split * a.P + p) * n * m;
    foreach_acc<SCfg>(acc, [&](int r, int c, int, int, int, float v) {
      if (r < n && c < m) out[(long)r * m + c] = v * inv;
    });
  }
}

struct FinishArgs {
  const float* ws;
  int nsplit, P, n, m;
  float lambda, inv_d;
  int cost_kind;
  const float* xsq[kMaxProb];  // 0.5*mean(x^2) per row (sq-Euclid cost only)
  const float* ysq[kMaxProb];
  float diag[kMaxProb];
  float* K;  // [P][n][m]
};

// Sum of the K-split partial sums + the cost epilogue, one output per thread.  Round 4: at N = 128 (128 splits of a
// 0.39 MB result = 50 MB of partial sums) this form ran at 1.5 TB/s -- 128 four-byte loads per thread, 1.5 workgroups
// per compute unit; cost_finish4_kernel below (float4 outputs, the splits shared by four waves) took its place and
// this kernel stays for totals that are not a multiple of 4 or fewer than four splits.
__global__ void cost_finish_kernel(FinishArgs a) {
  const long per = (long)a.n * a.m;
  const long total = per * a.P;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    const int p = (int)(idx / per);
    const long rem = idx - (long)p * per;
    const int i = (int)(rem / a.m), j = (int)(rem - (long)i * a.m);
    float dot = 0.f;
    for (int s = 0; s < a.nsplit; ++s) dot += a.ws[(long)s * total + idx];
    float c;
    if (a.cost_kind == OTGAN_COST_COSINE) c = 1.f - dot;
    else c = a.xsq[p][i] + a.ysq[p][j] - dot * a.inv_d;
    if (i == j) c += a.diag[p];
    a.K[idx] = -a.lambda * c;
  }
}

// 64 consecutive float4 outputs per workgroup; its four waves take a quarter of the splits each (8 independent 16-byte
// loads in flight per lane), partial sums combined through LDS in wave order: a fixed summation order, deterministic.
// (The sum over splits is grouped differently from the scalar kernel: results differ from it in the last bits.)
__global__ __launch_bounds__(256) void cost_finish4_kernel(FinishArgs a) {
  __shared__ f32x4 part[4][64];
  const long per = (long)a.n * a.m;
  const long total4 = per * a.P / 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long i4 = (long)blockIdx.x * 64 + lane;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (i4 < total4) {
    const int s0 = wave * a.nsplit / 4, s1 = (wave + 1) * a.nsplit / 4;
    const f32x4* src = reinterpret_cast<const f32x4*>(a.ws) + i4;
    int sp = s0;
    for (; sp + 8 <= s1; sp += 8) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(src + (long)(sp + u) * total4);
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; sp < s1; ++sp) acc += __builtin_nontemporal_load(src + (long)sp * total4);
  }
  part[wave][lane] = acc;
  __syncthreads();
  if (wave != 0 || i4 >= total4) return;
  const f32x4 dot4 = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
  f32x4 out;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const long idx = 4 * i4 + e;
    const int p = (int)(idx / per);
    const long rem = idx - (long)p * per;
    const int i = (int)(rem / a.m), j = (int)(rem - (long)i * a.m);
    float c;
    if (a.cost_kind == OTGAN_COST_COSINE) c = 1.f - dot4[e];
    else c = a.xsq[p][i] + a.ysq[p][j] - dot4[e] * a.inv_d;
    if (i == j) c += a.diag[p];
    out[e] = -a.lambda * c;
  }
  *reinterpret_cast<f32x4*>(a.K + 4 * i4) = out;
}

static void launch_cost_finish(const FinishArgs& fa, hipStream_t s) {
  const long total = (long)fa.P * fa.n * fa.m;
  if (total % 4 == 0 && fa.nsplit >= 4 && (reinterpret_cast<uintptr_t>(fa.ws) & 15) == 0 && (reinterpret_cast<uintptr_t>(fa.K) & 15) == 0) {
    hipLaunchKernelGGL(cost_finish4_kernel, dim3((unsigned)ceil_div_l(total / 4, 64)), dim3(256), 0, s, fa);
    return;
  }
  const int blocks = (int)(ceil_div_l(total, 256) < 2048 ? ceil_div_l(total, 256) : 2048);
  hipLaunchKernelGGL(cost_finish_kernel, dim3(blocks), dim3(256), 0, s, fa);
}

// out[r] = 0.5 * mean_k x[r][k]^2   (toy_example/matching_cpu.py:17)
__global__ void row_halfmeansq_kernel(const float* x, long ld, int rows, int D, float* out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = blockIdx.x * (blockDim.x >> 6) + wave;
  if (r >= rows) return;
  const float* p = x + (long)r * ld;
  double s = 0.0;
  for (int k = lane; k < D; k += 64) s += (double)p[k] * (double)p[k];
  s = wave_sum_d(s);
  if (lane == 0) out[r] = (float)(0.5 * s / D);
}

// ======================================================================================
// 2. Sinkhorn.  Potential form of the reference loop (matching.py:52-54):
//      log_a == K + f_i + g_j,   rows:  f_i = -LSE_j(K_ij + g_j),   cols: g_j = -LSE_i(K_ij + f_i)
//    exactly `iters` (rows, cols) sweeps, then the row softmax of matching.py:56.
//    LSE is max-shifted like tf.reduce_logsumexp.
// ======================================================================================

// ---- 2a. n, m <= 128: one 512-thread workgroup per problem, K resident in registers ------
// Thread (idx = t >> 2, q = t & 3) holds row-role values K[idx][32q..32q+31] and column-role values K[32q..32q+31][idx];
// potentials live in LDS.  The four slices of a line sit in ADJACENT lanes: a half-sweep is 32 entries per thread, two
// quad exchanges (DPP) and ONE barrier.  (Rounds 1 - 3 had the slices 128 threads apart and combined them through LDS
// with two barriers: round 2 measured this quad form at 1.12 us per half-sweep against 1.07 us, because the half-sweep was
// bound by the transcendental rate -- 16384 v_exp_f32 per problem at 16 per clock = 0.5 us -- and the LDS form stored the
// plan fully coalesced.  With the exponentials gone from most sweeps (below) the barriers and LDS round trips were what
// was left: 0.77 us per linear half-sweep in the LDS form.)
//
// Round 4 -- sweeps without exponentials.  With E = exp(K + f0 + g0) for potentials (f0, g0) of some earlier sweep and
// u = exp(f - f0), v = exp(g - g0), the same two updates read u_i = 1 / sum_j E_ij v_j and v_j = 1 / sum_i E_ij u_i:
// 32 multiply-adds per thread instead of 32 v_exp_f32 plus the max pass.  The iterates are the reference's
// (matching.py:52-54) up to rounding; what the linear form cannot do is start: exp(K) underflows whole rows at
// lambda = 500.  So the kernel runs the log-domain sweep until no potential moves by more than kLogSettle in a sweep (two
// or three sweeps), materialises E once (as many exponentials as one sweep), and continues in the linear form; should a
// scaling factor leave [e^-20, e^20] -- entries flushed to zero when E was made could begin to matter, or a sum overflow --
// it folds u, v into the potentials and goes back to the log-domain form (a NaN takes the same exit and stays loud).  The
// last half-step (the row softmax of matching.py:56) and the plan are log-domain as before.
constexpr float kLogSettle = 5.f;                       // nats per sweep
struct LinCtl {
  int enabled;
  float settle, lo, hi;    // kLogSettle; the scaling factors stay inside (lo, hi) = (e^-20, e^20)
  // sweep statistics (otgan_sinkhorn_counters; null = off): [0] problems solved, [1] log-domain sweeps, [2] linear sweeps,
  // [3] entries into the linear form, [4] fold-backs to the log-domain form, [5] sum over problems of the sweep at which the
  // linear form was first entered, [6] problems that never entered it
  unsigned long long* counters;
};
struct SweepCount {
  unsigned logs = 0, lins = 0, entries = 0, folds = 0, first = 0;
  __device__ __forceinline__ void commit(unsigned long long* c) const {
    if (!c) return;
    atomicAdd(c + 0, 1ull);
    atomicAdd(c + 1, (unsigned long long)logs);
    atomicAdd(c + 2, (unsigned long long)lins);
    atomicAdd(c + 3, (unsigned long long)entries);
    atomicAdd(c + 4, (unsigned long long)folds);
    atomicAdd(c + 5, (unsigned long long)first);
    if (!entries) atomicAdd(c + 6, 1ull);
  }
};
// value of lane (l ^ 1) / (l ^ 2) of the quad (DPP quad_perm [1,0,3,2] / [2,3,0,1])
__device__ __forceinline__ float quad_xor1(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float quad_xor2(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
}
// potentials of line i live at s[pot_slot(i)]: 36 floats per 32 lines, so that the four slices a quad reads at once (one
// float4 each, 32 lines apart) fall into different LDS banks
__device__ __forceinline__ int pot_slot(int i) { return i + 4 * (i >> 5); }
constexpr int kPotFloats = 4 * 36;

// "does any thread of the workgroup say yes", behind ONE barrier (the library's __syncthreads_or costs three): the yes
// goes to a flag word in LDS, four words used in turn -- the word of this call was cleared two calls ago, and thread 0
// clears the one two calls ahead behind the barrier (nobody reads or writes that one now).
__device__ __forceinline__ bool block_any(bool yes, unsigned* s_flag, unsigned& turn) {
  const unsigned slot = turn & 3u;
  ++turn;
  if (yes) s_flag[slot] = 1u;
  __syncthreads();
  const bool any = s_flag[slot] != 0u;
  if (threadIdx.x == 0) s_flag[(slot + 2u) & 3u] = 0u;
  return any;
}

// out[idx] = 1 / sum_e ev[e] * in[32 q + e]; true when some factor left (lo, hi) (or is not a number)
__device__ __forceinline__ bool small_lin_step(const float (&ev)[32], const float* s_in, float* s_out, int idx, int q,
                                               int limit, const LinCtl& lc, unsigned* s_flag, unsigned& turn) {
  const float4* in4 = reinterpret_cast<const float4*>(s_in + 36 * q);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const float4 g = in4[c];
    s0 = fmaf(ev[4 * c + 0], g.x, s0);
    s1 = fmaf(ev[4 * c + 1], g.y, s1);
    s2 = fmaf(ev[4 * c + 2], g.z, s2);
    s3 = fmaf(ev[4 * c + 3], g.w, s3);
  }
  float S = (s0 + s1) + (s2 + s3);
  S += quad_xor1(S);
  S += quad_xor2(S);                 // (s_q0 + s_q1) + (s_q2 + s_q3) in every lane of the quad
  bool far = false;
  if (q == 0) {
    const float u = idx < limit ? 1.f / S : 0.f;
    s_out[pot_slot(idx)] = u;
    far = idx < limit && !(u > lc.lo && u < lc.hi);
  }
  return block_any(far, s_flag, turn);
}

// log-domain half-step; returns true when some potential moved by more than `settle` (or is not a number)
__device__ __forceinline__ bool small_half_step(const float (&kv)[32], const float* s_in, float* s_out, int idx, int q,
                                                int limit, float settle, unsigned* s_flag, unsigned& turn) {
  const float4* in4 = reinterpret_cast<const float4*>(s_in + 36 * q);
  float v[32];
  float mx = -3.0e38f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const float4 g = in4[c];
    v[4 * c + 0] = kv[4 * c + 0] + g.x;
    v[4 * c + 1] = kv[4 * c + 1] + g.y;
    v[4 * c + 2] = kv[4 * c + 2] + g.z;
    v[4 * c + 3] = kv[4 * c + 3] + g.w;
    mx = fmaxf(mx, fmaxf(fmaxf(v[4 * c], v[4 * c + 1]), fmaxf(v[4 * c + 2], v[4 * c + 3])));
  }
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 32; ++e) s += exp_neg(v[e] - mx);
  float M = fmaxf(mx, quad_xor1(mx));
  M = fmaxf(M, quad_xor2(M));
  float S = s * exp_neg(mx - M);
  S += quad_xor1(S);
  S += quad_xor2(S);
  bool moved = false;
  if (q == 0) {
    const float fresh = (idx < limit) ? -(M + logf(S)) : 0.f;
    moved = !(fabsf(fresh - s_out[pot_slot(idx)]) < settle);
    s_out[pot_slot(idx)] = fresh;
  }
  return block_any(moved, s_flag, turn);
}

__global__ __launch_bounds__(512) void sinkhorn_small_kernel(const float* __restrict__ Kmat,
                                                             int n, int m, int iters,
                                                             float inv_lambda,
                                                             float* __restrict__ plan,
                                                             float* __restrict__ planT,
                                                             double* __restrict__ stats, LinCtl lc) {
  const int p = blockIdx.x;
  const float* K = Kmat + (long)p * n * m;
  const int t = threadIdx.x, idx = t >> 2, q = t & 3;
  __shared__ __attribute__((aligned(16))) float s_f[kPotFloats];
  __shared__ __attribute__((aligned(16))) float s_g[kPotFloats];
  __shared__ __attribute__((aligned(16))) float s_u[kPotFloats];
  __shared__ __attribute__((aligned(16))) float s_v[kPotFloats];
  __shared__ double s_red[3][8];
  __shared__ unsigned s_flag[4];
  unsigned turn = 0;
  if (t < 4) s_flag[t] = 0u;

  float kr[32], kc[32];
#pragma unroll
  for (int e = 0; e < 32; ++e) {
    const int j = 32 * q + e;
    kr[e] = (idx < n && j < m) ? K[(long)idx * m + j] : kNegBig;
  }
#pragma unroll
  for (int e = 0; e < 32; ++e) {
    const int i = 32 * q + e;
    kc[e] = (i < n && idx < m) ? K[(long)i * m + idx] : kNegBig;
  }
  if (t < kPotFloats) {
    s_f[t] = 0.f;
    s_g[t] = 0.f;
    s_u[t] = 1.f;
    s_v[t] = 1.f;
  }
  __syncthreads();

  float er[32], ec[32];
  bool linear = false;
  SweepCount sc;
  auto absorb = [&]() {   // back to potentials: f += log u, g += log v
    if (t < n) s_f[pot_slot(t)] += logf(s_u[pot_slot(t)]);
    if (t < m) s_g[pot_slot(t)] += logf(s_v[pot_slot(t)]);
    __syncthreads();
  };
  for (int it = 0; it < iters; ++it) {
    if (linear) ++sc.lins; else ++sc.logs;
    if (!linear) {
      const bool mf = small_half_step(kr, s_g, s_f, idx, q, n, lc.settle, s_flag, turn);  // rows:    f from g
      const bool mg = small_half_step(kc, s_f, s_g, idx, q, m, lc.settle, s_flag, turn);  // columns: g from f
      if (!mf && !mg && it + 1 < iters && lc.enabled) {
        // settled: E = exp(K + f + g) in both roles (masked entries: exp(-1e30) = 0), u = v = 1
        const float fi = s_f[pot_slot(idx)], gj = s_g[pot_slot(idx)];
#pragma unroll
        for (int e = 0; e < 32; ++e) {
          er[e] = expf(kr[e] + fi + s_g[36 * q + e]);
          ec[e] = expf(kc[e] + s_f[36 * q + e] + gj);
        }
        if (t < kPotFloats) {
          s_u[t] = 1.f;
          s_v[t] = 1.f;
        }
        __syncthreads();
        linear = true;
        if (!sc.entries++) sc.first = it + 1;
      }
    } else {
      const bool fu = small_lin_step(er, s_v, s_u, idx, q, n, lc, s_flag, turn);  // rows:    u from v
      const bool fv = small_lin_step(ec, s_u, s_v, idx, q, m, lc, s_flag, turn);  // columns: v from u
      if (fu || fv) {
        absorb();
        linear = false;
        ++sc.folds;
      }
    }
  }
  if (t == 0) sc.commit(lc.counters);
  if (linear) absorb();
  small_half_step(kr, s_g, s_f, idx, q, n, lc.settle, s_flag, turn);    // final row softmax (matching.py:56)

  // plan M_ij = exp(K_ij + f_i + g_j): the column role writes M, the row role M^T and the statistics (a wave stores
  // four 64-byte row segments per instruction: 0.8 MB per launch, once)
  {
    const float gj = s_g[pot_slot(idx)];
#pragma unroll
    for (int e = 0; e < 32; ++e) {
      const int i = 32 * q + e;
      if (i < n && idx < m) plan[(long)p * n * m + (long)i * m + idx] = expf(kc[e] + s_f[36 * q + e] + gj);
    }
  }
  float h = 0.f, w = 0.f, sm = 0.f;
  {
    const float fi = s_f[pot_slot(idx)];
#pragma unroll
    for (int e = 0; e < 32; ++e) {
      const int j = 32 * q + e;
      if (idx < n && j < m) {
        const float lm = kr[e] + fi + s_g[36 * q + e];
        const float mij = expf(lm);
        planT[(long)p * n * m + (long)j * n + idx] = mij;
        h -= mij * lm;
        w -= mij * kr[e];
        sm += mij;
      }
    }
  }
  double dh = wave_sum_d((double)h), dw = wave_sum_d((double)w), ds = wave_sum_d((double)sm);
  const int wave = t >> 6, lane = t & 63;
  if (lane == 0) {
    s_red[0][wave] = dh;
    s_red[1][wave] = dw;
    s_red[2][wave] = ds;
  }
  __syncthreads();
  if (t == 0) {
    double a = 0, b = 0, c = 0;
    for (int k = 0; k < 8; ++k) {
      a += s_red[0][k];
      b += s_red[1][k];
      c += s_red[2][k];
    }
    stats[p * 4 + 0] = a;                        // sum_i H(M_i.)
    stats[p * 4 + 1] = b * (double)inv_lambda;   // <M, C>,  C = -K/lambda
    stats[p * 4 + 2] = c;                        // sum(M)
    stats[p * 4 + 3] = 0.0;
  }
}

// ---- 2a'. 128 < N <= 1024 (square): persistent multi-workgroup kernel ----------------------
// Problem p is split over R = ceil(N / RPW) workgroups of 1024 threads.  Workgroup r keeps
// on chip both a row panel K[r*RPW .. +RPW][all columns] (row role, REGISTERS: 32 values per
// thread) and a column panel K[all rows][r*RPW .. +RPW] (column role, LDS: <= 128 KiB).
// A half-sweep is a local reduction over the panel; the only traffic between workgroups is
// the potential vector (N floats per problem) and one counter barrier per half-sweep
// (agent-scope release/acquire, placement independent; spins are bounded so a mis-launch
// cannot hang the GPU).  Thread t: line = t % RPW (its row / column inside the panel),
// q = t / RPW (which 32-wide slice of the other dimension it reduces).
constexpr int kPanelThreads = 1024;

struct PanelArgs {
  const float* K;      // [P][N][N]
  int N, iters, R;
  float inv_lambda;
  unsigned long long* f;   // [P][N] exchange buffers: (sequence number << 32) | float bits; zeroed before the launch
  unsigned long long* g;   // [P][N]
  unsigned* fail;          // [1] set when a spin gave up
  int P;                   // problems
  int xcd_local;           // 1: problem p runs on XCD p (workgroup x -> XCD x % 8, r = x / 8): its exchanges stay in one L2
  float* plan;
  float* planT;
  double* stats;       // zeroed before the launch
};

// Exchange protocol: every potential travels as ONE naturally aligned 8-byte word
// (half-sweep sequence number << 32 | float bits), published with a relaxed agent-scope atomic
// store (write-through, sc1) and polled by the thread that needs it with relaxed agent-scope
// atomic loads (L1 bypass) until the sequence number matches.  An 8-byte atomic is single-copy
// atomic, so value and tag arrive together: no counter, no flag line, no fence, no store drain --
// one store and one load round trip per half-sweep instead of store drain + counter atomic +
// counter poll + data load (measured: N = 1024, 100 sweeps 2.25 ms -> see profiles/README.md).
// Reuse of the two buffers is safe: a workgroup can publish f of sweep k+1 only after it consumed
// every g of sweep k, which every workgroup publishes only after consuming every f of sweep k.
// Round 5, XCD-local problems (PanelArgs::xcd_local): an agent-scope store (sc1) writes through AND drops the line from the
// writer's L2, so every poll -- also one from the same XCD -- is served at the cross-XCD rate (MI355X_MICROARCH.md,
// "stores of each flavour").  When all workgroups of a problem sit on ONE XCD the word can stay in that XCD's L2: a PLAIN
// 8-byte store (still single-copy atomic, still written through by the L1) and the same L1-bypassing polls, which then hit
// the L2.  Only valid while no other XCD reads the word -- the launcher checks the workgroup -> XCD mapping once per process
// and every workgroup checks its own XCC_ID.
__device__ __forceinline__ void publish_tagged(unsigned long long* p, float v, unsigned seq, bool local = false) {
  const unsigned long long w = ((unsigned long long)seq << 32) | (unsigned long long)__float_as_uint(v);
  if (local) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(w) : "memory");
  else __hip_atomic_store(p, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float consume_tagged(const unsigned long long* p, unsigned seq, unsigned* fail,
                                                bool& ok) {
  unsigned spins = 0;
  for (;;) {
    const unsigned long long w = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((unsigned)(w >> 32) == seq) return __uint_as_float((unsigned)w);
    if (++spins > (1u << 22)) {  // ~1 s: give up instead of hanging the device
      __hip_atomic_store(fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ok = false;
      return 0.f;
    }
    __builtin_amdgcn_s_sleep(1);
  }
}
__device__ __forceinline__ float value_of(const unsigned long long* p) {   // a completed exchange slot
  return __uint_as_float((unsigned)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// combine the TPR partial (max, sum) pairs of one line and publish -LSE to global memory
template <int TPR, int RPW>
__device__ __forceinline__ void panel_combine(float mx, float s, float* s_pm, float* s_ps, int line,
                                              int q, int gline, int N, unsigned long long* out_global,
                                              unsigned seq, bool local) {
  s_pm[q * RPW + line] = mx;
  s_ps[q * RPW + line] = s;
  __syncthreads();
  if (q == 0) {
    float M = s_pm[line];
#pragma unroll 4
    for (int k = 1; k < TPR; ++k) M = fmaxf(M, s_pm[k * RPW + line]);
    float S = 0.f;
#pragma unroll 4
    for (int k = 0; k < TPR; ++k) S += s_ps[k * RPW + line] * exp_neg(s_pm[k * RPW + line] - M);
    int gl = gline;
    asm volatile("" : "+v"(gl));      // (address formed here, not carried around the loop in a register pair)
    if (gline < N) publish_tagged(out_global + gl, -(M + logf(S)), seq, local);
  }
}

// exp through the native 2^x unit for arguments of either sign (entries of E are at most e^kLogSettle)
__device__ __forceinline__ float exp_native(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }

// The linear form's combine: the TPR partial sums of one line -> 1 / sum, published.  Round 4 (tools/debug/
// build_panel_timing.sh): the first version had the RPW threads of slice 0 add the TPR partial sums of their line one
// dependent 4-byte LDS read after the other -- 2.3 us of a 3.8 us half-sweep at N = 1024.  Spreading the lines over all
// sixteen waves (one partial sum per thread, exchanges, every wave publishing its two lines) was slower still (0.77 ->
// 0.95 ms per 100 sweeps): 32 separate 8-byte write-through stores instead of one 256-byte one.  So: ONE publishing
// stripe of RPW x H threads (H = 64 / RPW, at least 1: a whole wave at RPW = 32), each summing TPR / H partial sums
// fetched as independent 16-byte reads (rows of TPR + 4 floats: aligned, conflict-free), halves joined by one exchange,
// and the RPW results published by consecutive lanes.
template <int TPR, int RPW>
__device__ __forceinline__ void panel_combine_lin(float sum, float* s_part, int line, int q, int r, int N,
                                                  unsigned long long* out_global, unsigned seq, bool local) {
  constexpr int LD = TPR + 4;
  constexpr int H = RPW >= 64 ? 1 : 64 / RPW;       // threads per line in the publishing stripe
  constexpr int PER = TPR / H;                      // partial sums per thread: 16 (RPW 32), 16 (RPW 64), 8 (RPW 128)
  s_part[line * LD + q] = sum;
  __syncthreads();
  const int t = threadIdx.x;
  if (t < RPW * H) {
    const int ln = t % RPW, h = t / RPW;
    const f32x4* src = reinterpret_cast<const f32x4*>(s_part + ln * LD + h * PER);
    f32x4 acc = src[0];
#pragma unroll
    for (int c = 1; c < PER / 4; ++c) acc += src[c];
    float S = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    if (H == 2) S += __shfl_xor(S, 32, 64);
    int gl = r * RPW + ln;
    asm volatile("" : "+v"(gl));
    if (h == 0 && gl < N) publish_tagged(out_global + gl, 1.f / S, seq, local);
  }
}

// Sweeps without exponentials as in sinkhorn_small_kernel (see there): xr / the column panel hold the log-kernel K while the
// sweeps are log-domain and E = exp(K + f + g) while they are linear; s_f / s_g hold the potentials (of the last
// log-domain sweep), s_u / s_v the scaling factors exchanged since.  Every workgroup of a problem consumes the same N
// values per half-sweep and derives the mode from them alone (a flag word in LDS, four slots used in turn): the
// workgroups of a problem switch together without talking about it.
#ifdef PANEL_TIMING   // dev (tools/debug/build_panel_timing.sh): where a linear half-sweep's time goes, thread 0 of workgroup 0
#define PT_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) { const unsigned long long now_ = __builtin_readcyclecounter(); if (i) pt_acc[i] += now_ - pt_last; pt_last = now_; } } while (0)
#else
#define PT_STAMP(i) do { } while (0)
#endif
template <int TPR>  // slices per line: 8 (RPW 128, N <= 256), 16 (RPW 64, N <= 512), 32 (RPW 32)
__global__ __launch_bounds__(kPanelThreads) void sinkhorn_panel_kernel(PanelArgs a, LinCtl lc) {
#ifdef PANEL_TIMING
  unsigned long long pt_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pt_last = 0;
#endif
  constexpr int RPW = kPanelThreads / TPR;
  extern __shared__ __attribute__((aligned(16))) float psm[];
  float* s_kc = psm;                         // [TPR*32][RPW] column panel: K, or E in the linear sweeps
  float* s_f = s_kc + TPR * 32 * RPW;        // [1024] row potentials (all rows of the problem)
  float* s_g = s_f + 1024;                   // [1024] column potentials
  float* s_u = s_g + 1024;                   // [1024] row scaling factors of the linear sweeps
  float* s_v = s_u + 1024;                   // [1024]
  float* s_pm = s_v + 1024;                  // [TPR][RPW]
  float* s_ps = s_pm + TPR * RPW;            // [TPR][RPW]
  unsigned* s_flag = reinterpret_cast<unsigned*>(s_ps + TPR * RPW);   // [4] mode requests in turn, [4] a spin gave up
  const bool local = a.xcd_local != 0;
  const int p = local ? (int)(blockIdx.x & 7) : (int)(blockIdx.x / a.R), r = local ? (int)(blockIdx.x >> 3) : (int)(blockIdx.x % a.R);
  if (p >= a.P) return;                 // (XCD-local grid: 8 R workgroups, XCDs P .. 7 have no problem)
  if (local && (int)(__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 15) != p) {   // HW_REG_XCC_ID
    // not where the mapping says: plain stores would not reach the other workgroups of the problem -- fail loudly
    // (the others' spins are bounded; entropy and distance come out NaN)
    if (threadIdx.x == 0) __hip_atomic_store(a.fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const int N = a.N;
  const float* K = a.K + (long)p * N * N;
  const int t = threadIdx.x, line = t % RPW, q = t / RPW;
  const int gline = r * RPW + line;  // global row (row role) / column (column role)
  float xr[32];
  // index bases re-formed where they are used: left to itself the compiler keeps one loop-invariant LDS address per panel
  // entry (32 registers, most of them spilled) instead of one base and immediate offsets
  auto fresh = [](int v) {
    asm volatile("" : "+v"(v));
    return v;
  };
  auto load_k = [&]() {
    int gl = gline, q32 = q * 32;
    asm volatile("" : "+v"(gl), "+v"(q32));   // (the addresses are formed here, not hoisted out of the sweep loop: 64 registers)
#pragma unroll
    for (int e = 0; e < 32; ++e) {
      const int j = q32 + e;
      xr[e] = (j < N && gl < N) ? K[(long)gl * N + j] : kNegBig;
    }
    const int kb = fresh(q * 32 * RPW + line);
#pragma unroll 4
    for (int e = 0; e < 32; ++e) {
      const int i = q32 + e;
      s_kc[kb + e * RPW] = (i < N && gl < N) ? K[(long)i * N + gl] : kNegBig;
    }
  };
  unsigned long long* f = a.f + (long)p * N;
  unsigned long long* g = a.g + (long)p * N;
  unsigned phase = 0;
  s_f[t] = 0.f;
  s_g[t] = 0.f;  // g = 0 (1024 threads cover the 1024 slots)
  if (t < 8) s_flag[t] = 0u;
  __syncthreads();
  bool ok = true;
  // one exchange: every thread takes slot t of `slots` into dst[t]; `mark` = this value asks for a mode change.
  // Returns that request (the same in every workgroup of the problem); ok is cleared when a spin gave up.
  auto consume_all = [&](const unsigned long long* slots, float* dst, auto&& mark) -> bool {
    bool okl = true;
    int tt = t;
    asm volatile("" : "+v"(tt));      // (keeps the 64-bit slot addresses out of the loop-carried registers: they spilled)
    // a poll is a round trip through the fabric: one sent right behind the publish finds nothing and the next one costs
    // a second round trip; wait for about the propagation time first
    const float val = t < N ? consume_tagged(slots + tt, phase, a.fail, okl) : 0.f;
    PT_STAMP(4);
    const unsigned slot = phase & 3u;
    if (t < N && mark(val, dst[t])) s_flag[slot] = 1u;
    if (!okl) s_flag[4] = 1u;                      // a spin gave up: sticky
    dst[t] = val;
    __syncthreads();                               // (ONE barrier: the library's __syncthreads_and costs three)
    ok = s_flag[4] == 0u;
    const bool req = s_flag[slot] != 0u;
    if (t == 0) s_flag[(slot + 2u) & 3u] = 0u;     // (last read two barriers ago, next written two half-sweeps from now)
    return req;
  };
  auto log_rows = [&]() {   // f from g, published
    // Two passes over the thread's 32 entries (max, then the shifted sum) WITHOUT keeping the 32 sums K + g in
    // registers (round 3: 23 - 27 spilled registers inside the latency-critical loop).  The potentials are LDS
    // broadcasts (every lane of a slice reads the same word): reading them twice costs less than the spills did.
    const int q32 = fresh(q * 32);
    float mx = -3.0e38f;
#pragma unroll
    for (int e = 0; e < 32; ++e) mx = fmaxf(mx, xr[e] + s_g[q32 + e]);
    float sm = 0.f;
#pragma unroll
    for (int e = 0; e < 32; ++e) sm += exp_neg(xr[e] + s_g[q32 + e] - mx);
    panel_combine<TPR, RPW>(mx, sm, s_pm, s_ps, line, q, gline, N, f, ++phase, local);
  };
  auto moved = [&](float fresh, float old) { return !(fabsf(fresh - old) < lc.settle); };
  auto far = [&](float fresh, float) { return !(fresh > lc.lo && fresh < lc.hi); };
  bool linear = false, need_k = true;
  SweepCount sc;
  auto absorb = [&]() {   // linear -> log: f += log u, g += log v; K comes back at the top of the loop (its one load site)
    if (t < N) {
      s_f[t] += logf(s_u[t]);
      s_g[t] += logf(s_v[t]);
    }
    linear = false;
    need_k = true;
  };
  for (int it = 0;; ++it) {
    if (it == a.iters && linear) absorb();
    if (need_k) {
      load_k();
      __syncthreads();
      need_k = false;
    }
    if (it >= a.iters || !ok) break;
    if (linear) ++sc.lins; else ++sc.logs;
    if (!linear) {
      log_rows();
      const bool mf = consume_all(f, s_f, moved);
      if (!ok) break;
      {  // columns: g from f: the column panel lives in LDS; four chunks of eight keep the sums K + f of a chunk in
         // registers between its max and its shifted sum (one LDS read per entry), merged online
        float mx = -3.0e38f, sm = 0.f;
        const int q32 = fresh(q * 32), kb = fresh(q * 32 * RPW + line);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float v[8];
          float m = -3.0e38f;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            v[e] = s_kc[kb + (8 * c + e) * RPW] + s_f[q32 + 8 * c + e];
            m = fmaxf(m, v[e]);
          }
          float sc = 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) sc += exp_neg(v[e] - m);
          const float mn = fmaxf(mx, m);
          sm = sm * exp_neg(mx - mn) + sc * exp_neg(m - mn);
          mx = mn;
        }
        panel_combine<TPR, RPW>(mx, sm, s_pm, s_ps, line, q, gline, N, g, ++phase, local);
      }
      const bool mg = consume_all(g, s_g, moved);
      if (!ok) break;
      if (lc.enabled && !mf && !mg && it + 1 < a.iters) {
        // settled: E = exp(K + f + g) in both roles (masked entries: exp(-1e30) = 0), u = v = 1
        const float fl = s_f[gline < N ? gline : 0], gl = s_g[gline < N ? gline : 0];
        const int q32 = fresh(q * 32), kb = fresh(q * 32 * RPW + line);
#pragma unroll
        for (int e = 0; e < 32; ++e) xr[e] = exp_native(xr[e] + fl + s_g[q32 + e]);
#pragma unroll 4
        for (int e = 0; e < 32; ++e) s_kc[kb + e * RPW] = exp_native(s_kc[kb + e * RPW] + s_f[q32 + e] + gl);
        s_u[t] = 1.f;
        s_v[t] = 1.f;
        __syncthreads();
        linear = true;
        if (!sc.entries++) sc.first = it + 1;
      }
    } else {
      const int q32 = fresh(q * 32), kb = fresh(q * 32 * RPW + line);
      float sm = 0.f;
      PT_STAMP(0);
#pragma unroll
      for (int c = 0; c < 8; ++c) {        // (the factors as 16-byte broadcasts: 8 instead of 32 LDS instructions)
        const f32x4 vv = *reinterpret_cast<const f32x4*>(s_v + q32 + 4 * c);
        sm = fmaf(xr[4 * c + 0], vv[0], sm);
        sm = fmaf(xr[4 * c + 1], vv[1], sm);
        sm = fmaf(xr[4 * c + 2], vv[2], sm);
        sm = fmaf(xr[4 * c + 3], vv[3], sm);
      }
      PT_STAMP(1);
      panel_combine_lin<TPR, RPW>(sm, s_pm, line, q, r, N, f, ++phase, local);
      PT_STAMP(2);
      const bool fu = consume_all(f, s_u, far);
      PT_STAMP(3);
      if (!ok) break;
      sm = 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const f32x4 uu = *reinterpret_cast<const f32x4*>(s_u + q32 + 4 * c);
        sm = fmaf(s_kc[kb + (4 * c + 0) * RPW], uu[0], sm);
        sm = fmaf(s_kc[kb + (4 * c + 1) * RPW], uu[1], sm);
        sm = fmaf(s_kc[kb + (4 * c + 2) * RPW], uu[2], sm);
        sm = fmaf(s_kc[kb + (4 * c + 3) * RPW], uu[3], sm);
      }
      panel_combine_lin<TPR, RPW>(sm, s_pm, line, q, r, N, g, ++phase, local);
      const bool fv = consume_all(g, s_v, far);
      if (!ok) break;
      if (fu || fv) {
        absorb();
        ++sc.folds;
      }
    }
  }
  if (t == 0 && r == 0) sc.commit(lc.counters);     // (every workgroup of a problem takes the same decisions: one reports)
  if (ok) {   // the final row softmax (matching.py:56)
    log_rows();
    consume_all(f, s_f, moved);
  }
  // here s_f = final f (all rows), s_g = g of the last column step (all columns), xr / the panel = K
  // column role: plan[i][gline] = exp(K[i][gline] + f_i + g_gline), coalesced along the line
  const float gl = gline < N ? s_g[gline] : 0.f;
#pragma unroll 4
  for (int e = 0; e < 32; ++e) {
    const int i = q * 32 + e;
    if (i < N && gline < N)
      a.plan[(long)p * N * N + (long)i * N + gline] = expf(s_kc[i * RPW + line] + s_f[i] + gl);
  }
  // row role: transposed plan and statistics
  const float fl = gline < N ? s_f[gline] : 0.f;
  float h = 0.f, w = 0.f, sm = 0.f;
#pragma unroll
  for (int e = 0; e < 32; ++e) {
    const int j = q * 32 + e;
    if (j < N && gline < N) {
      const float lm = xr[e] + fl + s_g[j];
      const float mij = expf(lm);
      a.planT[(long)p * N * N + (long)j * N + gline] = mij;
      h -= mij * lm;
      w -= mij * xr[e];
      sm += mij;
    }
  }
  const double dh = wave_sum_d((double)h), dw = wave_sum_d((double)w), ds = wave_sum_d((double)sm);
  if ((t & 63) == 0) {
    atomicAdd(&a.stats[p * 4 + 0], dh);
    atomicAdd(&a.stats[p * 4 + 1], dw * (double)a.inv_lambda);
    atomicAdd(&a.stats[p * 4 + 2], ds);
  }
  if (!ok && t == 0) a.stats[p * 4 + 3] = __builtin_nan("");
#ifdef PANEL_TIMING
  if (threadIdx.x == 0 && blockIdx.x == 0)
    printf("panel timing (counter units, thread 0 of workgroup 0, linear row half-sweeps): fma %llu  combine+publish %llu  poll %llu  barrier after poll %llu\n",
           pt_acc[1], pt_acc[2], pt_acc[4], pt_acc[3]);
#endif
}

// OTGAN_SINKHORN_LINEAR=0: every sweep in the log domain (the kernels of rounds 1 - 3).  Test knobs:
// OTGAN_SINKHORN_LIN_RANGE=<nats> (default 20) narrows the band of the scaling factors, OTGAN_SINKHORN_SETTLE=<nats>
// (default 5) the entry condition -- tiny values force the fold-back path on every sweep.
unsigned long long* g_sweep_counters = nullptr;      // otgan_sinkhorn_counters(1): device buffer of eight counters
inline LinCtl lin_ctl_static() {
  static const LinCtl c = [] {
    LinCtl v;
    v.counters = nullptr;
    const char* e = getenv("OTGAN_SINKHORN_LINEAR");
    v.enabled = e && e[0] == '0' ? 0 : 1;
    const char* r = getenv("OTGAN_SINKHORN_LIN_RANGE");
    const float range = r && atof(r) > 0 ? (float)atof(r) : 20.f;
    const char* st = getenv("OTGAN_SINKHORN_SETTLE");
    v.settle = st && atof(st) > 0 ? (float)atof(st) : kLogSettle;
    v.lo = expf(-range);
    v.hi = expf(range);
    return v;
  }();
  return c;
}
inline LinCtl lin_ctl() {
  LinCtl c = lin_ctl_static();
  c.counters = g_sweep_counters;
  return c;
}

// Does this device place workgroup x of a one-dimensional grid on XCD x % 8?  (MI355X in SPX mode does; a partitioned
// device has fewer XCDs.)  Probed once per process with a 64-workgroup launch that records HW_REG_XCC_ID; OTGAN_PANEL_XCD=0
// keeps the problem-major grid and agent-scope publishes of round 4.
__global__ void xcc_id_probe_kernel(int* out) {
  if (threadIdx.x == 0) out[blockIdx.x] = (int)(__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 15);
}
// (cached PER DEVICE; never probed while `s` is capturing -- the probe allocates, launches on the NULL stream and
// copies back synchronously, any of which would invalidate a capture: an unprobed device then takes the placement-independent
// agent-scope protocol for that launch and is probed by the next launch outside a capture.)
inline bool xcd_round_robin(hipStream_t s) {
  static const bool enabled = [] {
    const char* e = getenv("OTGAN_PANEL_XCD");
    return !(e && e[0] == '0');
  }();
  if (!enabled) return false;
  constexpr int kMaxDev = 64;
  static std::atomic<int> state[kMaxDev];      // 0 unknown, 1 round-robin, 2 not
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) { (void)hipGetLastError(); return false; }
  const int st = state[dev].load(std::memory_order_acquire);
  if (st) return st == 1;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cap) != hipSuccess) { (void)hipGetLastError(); return false; }
  if (cap != hipStreamCaptureStatusNone) return false;
  int* d = nullptr;
  int h[64];
  bool ok = false;
  if (hipMalloc((void**)&d, sizeof(h)) == hipSuccess) {
    hipLaunchKernelGGL(xcc_id_probe_kernel, dim3(64), dim3(64), 0, 0, d);
    const bool copied = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess;
    (void)hipFree(d);
    if (!copied) (void)hipGetLastError();
    ok = copied;
    for (int i = 0; ok && i < 64; ++i) ok = h[i] == (i & 7);
  } else {
    (void)hipGetLastError();
  }
  state[dev].store(ok ? 1 : 2, std::memory_order_release);
  return ok;
}

// The panel kernel spin-waits across its P*R workgroups, so ALL of them must be resident at once.
// co_resident_capacity = (workgroups of this kernel one CU can host) x (CUs of the current device),
// from the occupancy calculator -- on a partitioned / CU-masked device it is smaller than on the
// full MI355X and the caller falls back to the multi-launch path.  (A device shared with another
// process can still starve the grid: the spins are bounded and the failure is reported through
// stats[p][3] = NaN, which poisons entropy and distance -- see entropy_finalize_kernel.)
template <int TPR>
bool launch_panel(const PanelArgs& a, int P, hipStream_t s) {
  constexpr int RPW = kPanelThreads / TPR;
  const size_t lds = sizeof(float) * ((size_t)TPR * 32 * RPW + 4 * 1024 + 2 * TPR * RPW + 8);
  static thread_local int cap_dev = -1, capacity = 0;
  int dev = 0;
  hipGetDevice(&dev);
  if (dev != cap_dev) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(sinkhorn_panel_kernel<TPR>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, sinkhorn_panel_kernel<TPR>, kPanelThreads, lds) !=
            hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
      (void)hipGetLastError();
      per_cu = cus = 0;
    }
    capacity = per_cu * cus;
    cap_dev = dev;
  }
  int cap = capacity;
  const char* lim = getenv("OTGAN_PANEL_MAX_WG");   // tests: pretend a smaller / partitioned device
  if (lim && atoi(lim) >= 0 && atoi(lim) < cap) cap = atoi(lim);
  if (P * a.R > cap) return false;
  PanelArgs b = a;
  b.P = P;
  // one problem per XCD when the device dispatches workgroup x to XCD x % 8 (probed once) and every XCD can host a problem's
  // R workgroups: the grid is then 8 R workgroups of which those on XCDs P .. 7 return at once
  b.xcd_local = (P <= 8 && 8 * a.R <= capacity && !(lim && atoi(lim) >= 0) && xcd_round_robin(s)) ? 1 : 0;
  const unsigned grid = b.xcd_local ? 8u * (unsigned)a.R : (unsigned)(P * a.R);
  hipLaunchKernelGGL(sinkhorn_panel_kernel<TPR>, dim3(grid), dim3(kPanelThreads), lds, s, b, lin_ctl());
  return true;
}

// ---- 2b. general sizes: K stays in HBM/L2, two kernels per sweep -------------------------
// rows: one wave per row.  out[p][i] = -LSE_j(K[p][i][j] + in[p][j])
__global__ __launch_bounds__(256) void sinkhorn_row_kernel(const float* __restrict__ Kmat,
                                                           int n, int m,
                                                           const float* __restrict__ g,
                                                           float* __restrict__ f) {
  const int p = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + wave;
  if (i >= n) return;
  const float* row = Kmat + ((long)p * n + i) * m;
  const float* gp = g + (long)p * m;
  float mx = -3.0e38f;
  for (int j = lane; j < m; j += 64) mx = fmaxf(mx, row[j] + gp[j]);
  mx = wave_max(mx);
  float s = 0.f;
  for (int j = lane; j < m; j += 64) s += exp_neg(row[j] + gp[j] - mx);
  s = wave_sum(s);
  if (lane == 0) f[(long)p * n + i] = -(mx + logf(s));
}

// columns: a 256-thread block owns 64 columns (lane = column); its 4 waves split the rows.
__global__ __launch_bounds__(256) void sinkhorn_col_kernel(const float* __restrict__ Kmat,
                                                           int n, int m,
                                                           const float* __restrict__ f,
                                                           float* __restrict__ g) {
  const int p = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j = blockIdx.x * 64 + lane;
  const float* Kp = Kmat + (long)p * n * m;
  const float* fp = f + (long)p * n;
  __shared__ float s_m[4][64], s_s[4][64];
  float mx = -3.0e38f, s = 0.f;
  if (j < m) {
    for (int i = wave; i < n; i += 4) mx = fmaxf(mx, Kp[(long)i * m + j] + fp[i]);
    for (int i = wave; i < n; i += 4) s += exp_neg(Kp[(long)i * m + j] + fp[i] - mx);
  }
  s_m[wave][lane] = mx;
  s_s[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && j < m) {
    const float M = fmaxf(fmaxf(s_m[0][lane], s_m[1][lane]), fmaxf(s_m[2][lane], s_m[3][lane]));
    float S = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) S += s_s[w][lane] * exp_neg(s_m[w][lane] - M);
    g[(long)p * m + j] = -(M + logf(S));
  }
}

// final: row softmax, plan, transposed plan, statistics (atomics into zeroed stats).
__global__ __launch_bounds__(256) void sinkhorn_final_kernel(const float* __restrict__ Kmat,
                                                             int n, int m,
                                                             const float* __restrict__ g,
                                                             float inv_lambda,
                                                             float* __restrict__ plan,
                                                             float* __restrict__ planT,
                                                             double* __restrict__ stats) {
  const int p = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + wave;
  if (i >= n) return;
  const float* row = Kmat + ((long)p * n + i) * m;
  const float* gp = g + (long)p * m;
  float mx = -3.0e38f;
  for (int j = lane; j < m; j += 64) mx = fmaxf(mx, row[j] + gp[j]);
  mx = wave_max(mx);
  float s = 0.f;
  for (int j = lane; j < m; j += 64) s += exp_neg(row[j] + gp[j] - mx);
  s = wave_sum(s);
  const float fi = -(mx + logf(s));
  float h = 0.f, w = 0.f, sm = 0.f;
  float* prow = plan + ((long)p * n + i) * m;
  float* pT = planT + (long)p * n * m;
  for (int j = lane; j < m; j += 64) {
    const float kij = row[j];
    const float lm = kij + fi + gp[j];
    const float mij = expf(lm);
    prow[j] = mij;
    pT[(long)j * n + i] = mij;
    h -= mij * lm;
    w -= mij * kij;
    sm += mij;
  }
  const double dh = wave_sum_d((double)h), dw = wave_sum_d((double)w), ds = wave_sum_d((double)sm);
  if (lane == 0) {
    atomicAdd(&stats[p * 4 + 0], dh);
    atomicAdd(&stats[p * 4 + 1], dw * (double)inv_lambda);
    atomicAdd(&stats[p * 4 + 2], ds);
  }
}

// ======================================================================================
// 3. plan application: out = alpha * sum_t plan_t[rows, kdim_t] . feat_t[kdim_t, D]
// ======================================================================================
struct ApplyTerm {
  const float* plan;
  const float* feat;
  long ldp;
  int kdim;
};
struct ApplyBlock {
  float* out;
  int rows;
  int nterms;
  float alpha;
  ApplyTerm t[3];
  // the accumulators are multiplied by rescale[t] BEFORE term t is added (1 or 0: untouched).  A power of two is
  // exact, so "(-1/2)(M2 F1 + M3 F2) + M0 F0" -- an injected gradient f_aa - f_ab (train.py:111) -- is one block:
  // the two half-weight terms first, rescale[2] = -0.5, then the unit term
  float rescale[3];
};
struct ApplyArgs {
  ApplyBlock b[8];
  int D;
  long ldf, ldo;
  int tiles_d;
};

template <bool VEC>
__global__ __launch_bounds__(256) void plan_apply_kernel(ApplyArgs a) {
  using LA = MatLoaderK<SCfg, 128, VEC>;
  using LB = MatLoaderR<SCfg, 128, VEC>;
  __shared__ __attribute__((aligned(16))) float smem[2 * LA::FLOATS + 2 * LB::FLOATS];
  const ApplyBlock& blk = a.b[blockIdx.y];
  const int tmi = blockIdx.x / a.tiles_d, tdi = blockIdx.x % a.tiles_d;
  const int row0 = tmi * 128, d0 = tdi * 128;
  if (row0 >= blk.rows) return;
  typename SCfg::acc_t acc[SCfg::MT][SCfg::NT];
  zero_acc<SCfg>(acc);
  for (int t = 0; t < blk.nterms; ++t) {
    const float rs = blk.rescale[t];
    if (rs != 0.f && rs != 1.f) {
#pragma unroll
      for (int i = 0; i < SCfg::MT; ++i)
#pragma unroll
        for (int j = 0; j < SCfg::NT; ++j) acc[i][j] *= rs;
    }
    LA la;
    LB lb;
    la.init(blk.t[t].plan + (long)row0 * blk.t[t].ldp, blk.t[t].ldp, blk.rows - row0, blk.t[t].kdim);
    lb.init(blk.t[t].feat + d0, a.ldf, a.D - d0, blk.t[t].kdim);
    gemm_mainloop<SCfg>(la, lb, (blk.t[t].kdim + SCfg::BK - 1) / SCfg::BK, smem, acc);
  }
  float* out = blk.out;
  const int rows = blk.rows, D = a.D;
  const long ldo = a.ldo;
  const float alpha = blk.alpha;
  foreach_acc<SCfg>(acc, [&](int r, int c, int, int, int, float v) {
    const int row = row0 + r, col = d0 + c;
    if (row < rows && col < D) out[(long)row * ldo + col] = alpha * v;
  });
}

// The plan application of the one-tile problems (N <= 128) on the fp16 matrix pipe (round 5), next to cost128_h2_kernel:
// out[128 x 128 d] = sum_t coef_t * plan_t[128 x kdim] . feat_t[kdim x 128 d], both operands split while they are staged
// (plan entries are at most 1 -- rows and columns sum to 1 or less -- and features are rows of unit length: scale 2^13 for both).
// The plan rows are k-contiguous (the cost kernel's LDS planes); the features are d-contiguous, so their tile goes to LDS
// as it is read -- eight blocks of [16 k][16 d] fp16 (32 bytes a row, the two 16-byte halves swapped on k rows 8..15, 32
// bytes of padding per block so that a 16-lane write group hits 32 distinct banks) -- and the MFMA fragments (eight
// consecutive k of one column per lane) come out of ds_read_b64_tr_b16, as in the t-leading mode of gemm_x3.h.
// The accumulator rescaling of plan_apply_kernel ("(-1/2)(M2 F1 + M3 F2) + M0 F0") is folded into the plan operand's
// scale per term (powers of two: exact).  Workgroup x -> XCD x % 8: the blocks of one d tile (they read the same
// feature columns) take adjacent slots of one XCD.
struct ApplyH2Term {
  const float* plan;
  const float* feat;
  float coef;      // pscale * (product of the rescale factors applied after this term) : the plan operand's scale
};
struct ApplyH2Block {
  float* out;
  float out_scale;   // alpha / (pscale * kCostH2Scale)
  ApplyH2Term t[3];
};
struct ApplyH2Args {
  ApplyH2Block b[4];
  int nblk, rows, kdim, nterms, D, tiles_d;
  long ldp, ldf, ldo;
};
constexpr int kApplyCbStride = 544;                       // bytes per [16 k][16 d] block (512 + padding)
constexpr int kApplyPB = 8 * kApplyCbStride;              // one feature plane of a stage
constexpr int kApplyPA = 128 * kX3sRowBytes;              // one plan plane of a stage

__global__ __launch_bounds__(256) void plan_apply128_h2_kernel(ApplyH2Args a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 2 * kApplyPA + 2 * 2 * kApplyPB];
  unsigned char* sA = smem;
  unsigned char* sB = smem + 4 * kApplyPA;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int dtile = (slot / a.nblk) * 8 + xcd, z = slot % a.nblk;
  if (dtile >= a.tiles_d) return;
  const int d0 = dtile * 128;
  const ApplyH2Block& blk = a.b[z];
  const int spt = a.kdim >> 4, nst = a.nterms * spt;     // k steps per term, in all
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // ---- loaders.  Plan: thread -> (row r0 / r0 + 64, float4 c of the 16 k); features: thread -> (k row kk / kk + 8, float4 d4)
  const int pc = tid & 3, pr0 = tid >> 2;
  const int fd4 = tid & 31, fk0 = tid >> 5;
  const int rlast = a.rows - 1;
  int dcol = d0 + 4 * fd4;
  const bool dok = dcol < a.D;
  if (!dok) dcol = a.D - 4;
  const long prow0 = (long)(pr0 < rlast ? pr0 : rlast) * a.ldp, prow1 = (long)(pr0 + 64 < rlast ? pr0 + 64 : rlast) * a.ldp;
  auto load = [&](int st, float4 (&rp)[2], float4 (&rf)[2]) {
    int s2 = st < nst ? st : nst - 1;                    // (steps past the end re-read the last one; stored with scale 0)
    const int term = s2 / spt, kt = s2 - term * spt;
    const float* pp = blk.t[term].plan + kt * 16 + 4 * pc;
    const float* fp = blk.t[term].feat + (long)(kt * 16 + fk0) * a.ldf + dcol;
    rp[0] = *reinterpret_cast<const float4*>(pp + prow0);
    rp[1] = *reinterpret_cast<const float4*>(pp + prow1);
    rf[0] = *reinterpret_cast<const float4*>(fp);
    rf[1] = *reinterpret_cast<const float4*>(fp + 8 * a.ldf);
  };
  const int fgrp = fd4 & 3, fcb = fd4 >> 2;
  auto store = [&](int st, int buf, const float4 (&rp)[2], const float4 (&rf)[2]) {
    const bool live = st < nst;
    const int term = live ? st / spt : 0;
    const float ps = live ? blk.t[term].coef : 0.f;
    const float fs = (live && dok) ? kCostH2Scale : 0.f;
    unsigned char* ta = sA + buf * 2 * kApplyPA;
    unsigned char* tb = sB + buf * 2 * kApplyPB;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      float4 v = rp[p];
      v.x *= ps; v.y *= ps; v.z *= ps; v.w *= ps;
      x2h_store4(ta, kApplyPA, pr0 + 64 * p, 4 * pc, v);
      float4 w = rf[p];
      w.x *= fs; w.y *= fs; w.z *= fs; w.w *= fs;
      const int kk = fk0 + 8 * p;
      unsigned char* d = tb + fcb * kApplyCbStride + kk * 32 + (((fgrp >> 1) ^ p) << 4) + ((fgrp & 1) << 3);   // (kk >> 3 == p)
      gt_f32x2 x = {w.x, w.y}, y = {w.z, w.w};
      const gt_f16x2 hx = __builtin_convertvector(x, gt_f16x2), hy = __builtin_convertvector(y, gt_f16x2);
      x -= __builtin_convertvector(hx, gt_f32x2);
      y -= __builtin_convertvector(hy, gt_f32x2);
      const gt_f16x2 lx = __builtin_convertvector(x, gt_f16x2), ly = __builtin_convertvector(y, gt_f16x2);
      *reinterpret_cast<gt_u32x2*>(d) = gt_u32x2{__builtin_bit_cast(unsigned, hx), __builtin_bit_cast(unsigned, hy)};
      *reinterpret_cast<gt_u32x2*>(d + kApplyPB) = gt_u32x2{__builtin_bit_cast(unsigned, lx), __builtin_bit_cast(unsigned, ly)};
    }
  };
  // ---- fragments
  const int li = lane & 31, lh = lane >> 5;
  const int g4 = lane >> 4, l16 = lane & 15;
  int a_off[2], b_off[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    a_off[t] = x3s_off((wm * 2 + t) * 32 + li, 8 * lh);
    b_off[t] = ((wn * 2 + t) * 2 + (g4 & 1)) * kApplyCbStride + (8 * (g4 >> 1) + (l16 >> 2)) * 32 +
               ((((l16 >> 1) & 1) ^ (g4 >> 1)) << 4) + ((l16 & 1) << 3);
  }
  typename SCfg::acc_t acc[2][2];
  zero_acc<SCfg>(acc);
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  auto mma = [&](int buf) {
    const unsigned char* pa = sA + buf * 2 * kApplyPA;
    const unsigned char* pb = sB + buf * 2 * kApplyPB;
    gt_f16x8 fa[2][2], fb[2][2];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        fa[q][t] = *reinterpret_cast<const gt_f16x8*>(pa + q * kApplyPA + a_off[t]);
        const unsigned char* p = pb + q * kApplyPB + b_off[t];
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 128));
        fb[q][t] = __builtin_bit_cast(gt_f16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
      }
#pragma unroll
    for (int term = 0; term < 3; ++term) {   // lo*hi, hi*lo, hi*hi (smallest first)
      constexpr int pa_of[3] = {1, 0, 0}, pb_of[3] = {0, 1, 0};
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[pa_of[term]][mt], fb[pb_of[term]][nt], acc[mt][nt], 0, 0, 0);
    }
  };
  // ---- the cost kernel's loop: two register sets, k steps st + 1 and st + 2 in flight while step st is multiplied
  float4 p0[2], p1[2], f0[2], f1[2];
#define APPLY_H2_PIN(R)                                                                                       \
  asm volatile("" : "+v"(*reinterpret_cast<gt_f32x4*>(&R[0])), "+v"(*reinterpret_cast<gt_f32x4*>(&R[1])))
  load(0, p0, f0);
  load(1, p1, f1);
  store(0, 0, p0, f0);
  __syncthreads();
  for (int st = 0; st < nst; st += 2) {
    load(st + 2, p0, f0);
    __builtin_amdgcn_sched_barrier(0);
    mma(0);
    __builtin_amdgcn_sched_barrier(0);
    APPLY_H2_PIN(p1); APPLY_H2_PIN(f1);
    store(st + 1, 1, p1, f1);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    load(st + 3, p1, f1);
    __builtin_amdgcn_sched_barrier(0);
    mma(1);
    __builtin_amdgcn_sched_barrier(0);
    APPLY_H2_PIN(p0); APPLY_H2_PIN(f0);
    store(st + 2, 0, p0, f0);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
  }
#undef APPLY_H2_PIN
  float* out = blk.out;
  const int rows = a.rows, D = a.D;
  const long ldo = a.ldo;
  const float os = blk.out_scale;
  foreach_acc<SCfg>(acc, [&](int r, int c, int, int, int, float v) {
    const int col = d0 + c;
    if (r < rows && col < D) out[(long)r * ldo + col] = os * v;
  });
}

// ======================================================================================
// 4. distance (matching.py:139-153) with fp64 accumulation
// ======================================================================================
__global__ __launch_bounds__(256) void dot3_kernel(const float* __restrict__ a,
                                                   const float* __restrict__ b,
                                                   const float* __restrict__ aa,
                                                   const float* __restrict__ bb,
                                                   const float* __restrict__ ab, long total,
                                                   double* __restrict__ out3) {
  double s_aa = 0, s_bb = 0, s_ab = 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const double av = a[i], bv = b[i];
    s_aa += av * (double)aa[i];
    s_bb += bv * (double)bb[i];
    s_ab += av * (double)ab[i];
  }
  s_aa = wave_sum_d(s_aa);
  s_bb = wave_sum_d(s_bb);
  s_ab = wave_sum_d(s_ab);
  __shared__ double red[3][4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) {
    red[0][wave] = s_aa;
    red[1][wave] = s_bb;
    red[2][wave] = s_ab;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&out3[0], red[0][0] + red[0][1] + red[0][2] + red[0][3]);
    atomicAdd(&out3[1], red[1][0] + red[1][1] + red[1][2] + red[1][3]);
    atomicAdd(&out3[2], red[2][0] + red[2][1] + red[2][2] + red[2][3]);
  }
}

// stats[p][3] is 0 for a solved problem and NaN when the persistent Sinkhorn kernel gave up (its
// workgroups were not co-resident): adding it in makes entropy AND distance NaN, so a failed
// matching can never pass for a valid one (the plans / matched features of that call are garbage).
__device__ __forceinline__ double failure_poison(const double* stats, int P) {
  double z = 0.0;
  if (stats)
    for (int p = 0; p < P; ++p) z += stats[p * 4 + 3];
  return z;
}

__global__ void distance_finalize_kernel(const double* s3, double denom, double* dist, const double* stats,
                                         int P) {
  // (nd_bb + nd_aa - 2 nd_ab) / denom          (matching.py:150,152)
  dist[0] = (s3[1] + s3[0] - 2.0 * s3[2]) / denom + failure_poison(stats, P);
}

__global__ void entropy_finalize_kernel(const double* stats, int P, int n, float* entropy) {
  double e = failure_poison(stats, P);
  for (int p = 0; p < P; ++p) e += stats[p * 4 + 0] / (double)n;  // mean row entropy (matching.py:57)
  entropy[0] = (float)(e / P);                                     // mean over problems (:61)
}

// two-batch distance from the per-problem statistics (cancellation-free closed form of
// SURVEY.md section 3.4): nd = sum(M) - <M,C> per problem (cosine cost),
// dist = [2 T(a1a2) + 2 T(b2b1) - T(a1b1) - T(a1b2) - T(a2b1) - T(a2b2)] / (4N)
__global__ void closed_form_distance_kernel(const double* stats, int N, double* dist) {
  double T[6];
  for (int p = 0; p < 6; ++p) T[p] = stats[p * 4 + 2] - stats[p * 4 + 1];
  dist[0] = (2.0 * T[0] + 2.0 * T[1] - (T[2] + T[3] + T[4] + T[5])) / (4.0 * N) + failure_poison(stats, 6);
}

// single-batch distance the same way (matching.py:147-152 with the three plans of :131-134): the reference's nd_xy is a dot
// product with the REAL features, i.e. nd_xy = sum(M_xy) - <M_xy, C_xy> with the cosine cost WITHOUT the 999 the a-a / b-b
// problems carry on their diagonals (matching.py:109-110), while the kernel statistics hold <M, C + 999 I>: 999 trace(M) is
// added back (exactly 0 from lambda * 999 of about 90 upwards, 999 sum(M_ii) / (2 n) of bias for small lambda),
// dist = (nd_bb + nd_aa - 2 nd_ab) / (2 n).  One workgroup of 256 threads; plan = the three [n][n] plans.
__global__ void closed_form_single_distance_kernel(const double* stats, const float* plan, int n, double diag, double* dist) {
  __shared__ double tr[2][256];
  for (int p = 0; p < 2; ++p) {
    double t = 0.0;
    for (long i = threadIdx.x; i < n; i += 256) t += (double)plan[(long)p * n * n + i * (n + 1L)];
    tr[p][threadIdx.x] = t;
  }
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {          // fixed-order tree: deterministic
    if ((int)threadIdx.x < w) {
      tr[0][threadIdx.x] += tr[0][threadIdx.x + w];
      tr[1][threadIdx.x] += tr[1][threadIdx.x + w];
    }
    __syncthreads();
  }
  if (threadIdx.x) return;
  double T[3];
  for (int p = 0; p < 3; ++p) T[p] = stats[p * 4 + 2] - stats[p * 4 + 1];
  T[0] += diag * tr[0][0];
  T[1] += diag * tr[1][0];
  dist[0] = (T[1] + T[0] - 2.0 * T[2]) / (2.0 * n) + failure_poison(stats, 3);
}

// ======================================================================================
// 5. The matching GEMMs on the fp16 matrix pipe with split-precision operands (gemm_x3.h)
// ======================================================================================
// For N >= 256 (the 64x64 configuration and the multi-GPU problems) both GEMM families of the block run on
// the 256 x 256 split-precision engine that carries the convolutions.  Round 4: every fp32 operand is two
// scaled fp16 planes (x 2^s = hi + lo: 22 significand bits, three MFMAs per product, fp32 accumulate) -- rounds
// 2 - 3 kept three bf16 planes here (24 bits, six MFMAs) because the log-kernel amplifies the dot product's
// error by lambda; measured, the error of a D = 32768 dot product is the fp32 ACCUMULATION's (max |K - fp64|
// 1.6e-4 at lambda = 500 with either operand form, 1.7e-3 with two K splits instead of 21: longer fp32 chains),
// not the operands'.  Plain fp16 / bf16 inputs still fail the 1e-4 loss bound.
//   * the stacked features [a1; a2; b1; b2] are split ONCE into the blocked operand layout (op_off): the cost
//     GEMMs read it as the row-major NT operand (rows = samples, k = D), the plan application reads the same
//     buffer as its t-leading B operand (rows = contraction index = samples, columns = D);
//   * cost: C_p = X_p . Y_p^T, six problems in one launch (blockIdx.z, operand offsets per problem); one K split
//     -> the epilogue writes K = -lambda (1 - x.y) itself, more -> partial sums + cost_finish_kernel;
//   * plan application: out = A^T . F with A = the plan (or its transpose) as a t-leading operand; the two-term
//     blocks 0.5 (M_1 F_1 + M_2 F_2) are ONE GEMM over the stacked contraction index [F_1; F_2], alpha folded
//     into the plan planes; eight output blocks (four for a rank's row range) in one launch.
__device__ __forceinline__ f32x4 x3_ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

struct SplitSrc {
  const float* src[12];
  long ld[12];
  long row0[12];     // destination row inside the stacked operand
  float scale[12];
  int n;
};
// src[i]: [rows x K] fp32 row-major  ->  dst: stacked operand (kblocks = K / 16), two fp16 planes of scale[i] * src[i] * 2^(14 - e).
// A wave covers 16 rows x 16 k: every store instruction writes 512 contiguous bytes of one chunk.
//
// The power-of-two scale of an operand needs the largest magnitude in it, and a reduction pass over the features of a
// rank's problem would be one more read of 0.4 - 0.8 GB.  Instead: pass 0 splits with the exponent e0 the caller expects
// (features of a cosine cost are rows of unit length: |x| < 2; plan entries are at most 1) and max-accumulates the
// magnitudes it sees into the operand's record; pass 1 -- same grid, normally a few hundred idle workgroups -- reads the
// record, writes the header the GEMM scales its sums by, and only when the expectation was wrong (|x| >= 2^(e0 + 1):
// the hi piece could overflow; or < 2^(e0 - 12): lo pieces would go subnormal) splits everything again with the
// exponent of the data.  Operand header: X3_HDR floats (gemm_x3.h) + one amax record (common.h) in front of the planes.
constexpr int kX3HdrFloats = X3_HDR + kAmaxSub * kAmaxSubStride;
constexpr size_t kX3HdrBytes = sizeof(float) * kX3HdrFloats;
__host__ __device__ inline float* x3_hdr(const u16* planes) {
  return reinterpret_cast<float*>(const_cast<u16*>(planes)) - kX3HdrFloats;
}
__global__ __launch_bounds__(256) void x3_split_rows_kernel(SplitSrc a, int rows, int K, u16* dst, long plane_stride, float* hdr,
                                                            int e0, int pass, long nblocks) {
  int e = e0;
  if (pass == 1) {
    const float amax = amax_record_value(hdr + X3_HDR);
    const int ea = x3_scale_exp(amax, 1.f, 1.f);          // amax < 2^ea  (0 for an all-zero or NaN operand)
    const bool keep = !(amax > 0.f) || (ea <= e0 + 1 && ea >= e0 - 12);
    if (!keep) e = ea;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 40) {
      if (threadIdx.x == 0) hdr[0] = amax;
      hdr[16 + threadIdx.x] = ldexpf(1.f, 14 - e);
      hdr[64 + threadIdx.x] = ldexpf(1.f, e - 14);
    }
    if (keep) return;
  }
  const int nk4 = K >> 2, kgroups = (nk4 + 15) >> 4;
  const int i = blockIdx.y;
  const float up = ldexpf(1.f, 14 - e);
  unsigned mb = 0u;
  // (pass 0: one position per workgroup; pass 1: a small grid that walks all of them in the rare case it has work)
  for (long b = blockIdx.x; b < nblocks; b += gridDim.x) {
    const int k4 = (int)(b % kgroups) * 16 + (threadIdx.x >> 6) * 4 + (threadIdx.x & 3);
    const long row = (b / kgroups) * 16 + ((threadIdx.x >> 2) & 15);
    if (row >= rows || k4 >= nk4) continue;
    f32x4 v = x3_ld4(a.src[i] + row * a.ld[i] + 4 * k4);
    v *= a.scale[i];
    mb = amax_bits4(v, mb);
    v *= up;
    st_split4h(dst, plane_stride, op_off(a.row0[i] + row, 4 * k4, K >> 4), v);
  }
  if (pass == 0) amax_commit(hdr + X3_HDR, mb);
}

template <auto Kern>
inline void x3_ensure_lds() {
  static const bool done = [] {
    hipFuncSetAttribute(reinterpret_cast<const void*>(Kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)X3_LDS);
    return true;
  }();
  (void)done;
}

inline bool match_x3_enabled() {
  static const bool on = [] {
    const char* e = getenv("OTGAN_MATCH_FP32");     // 1: keep the matching GEMMs on the exact-fp32 MFMA engine
    return !(e && e[0] == '1');
  }();
  return on;
}
// operand elements (u16 per plane) of a stacked [rows x K] operand
inline size_t x3_plane_elems(size_t rows, size_t K) { return ((rows + 31) / 32) * 32 * K; }
inline long x3_row_off(long row, long K) { return (row >> 5) * (K >> 4) * 512; }   // element offset of row block row/32

// one or more launches into operands that share a header (and with it one scale): all first passes, then all second ones
struct X3SplitJob {
  SplitSrc ss;
  int rows, K;
  u16* dst;
  long plane_stride;
};
constexpr int kX3FeatureExp = 1;      // cosine-cost features: |x| < 2 expected (rows of unit length)
void x3_split_group(const X3SplitJob* jobs, int njobs, float* hdr, int e0, hipStream_t s) {
  hipMemsetAsync(hdr + X3_HDR, 0, sizeof(float) * kAmaxSub * kAmaxSubStride, s);
  for (int pass = 0; pass < 2; ++pass)
    for (int j = 0; j < njobs; ++j) {
      const X3SplitJob& q = jobs[j];
      if (!q.ss.n) continue;
      const int nk4 = q.K / 4, kgroups = (nk4 + 15) / 16;
      const long nblocks = ((long)(q.rows + 15) / 16) * kgroups;
      const dim3 grid((unsigned)(pass == 0 || nblocks < 1024 ? nblocks : 1024), q.ss.n);
      hipLaunchKernelGGL(x3_split_rows_kernel, grid, dim3(256), 0, s, q.ss, q.rows, q.K, q.dst, q.plane_stride, hdr, e0, pass,
                         nblocks);
    }
}
void x3_split(const SplitSrc& ss, int rows, int K, u16* dst, long plane_stride, int e0, hipStream_t s) {
  X3SplitJob q{ss, rows, K, dst, plane_stride};
  x3_split_group(&q, 1, x3_hdr(dst), e0, s);
}
// expected exponent of a plan operand: entries of a plan are at most 1 (its rows and columns sum to 1 or less)
inline int x3_plan_exp(const SplitSrc& ss) {
  float m = 0.f;
  for (int i = 0; i < ss.n; ++i) m = fmaxf(m, fabsf(ss.scale[i]));
  return x3_scale_exp(m, 1.f, 1.f);
}

// the conditions under which the split-precision engine takes a matching problem of n x m blocks, feature width D
inline bool x3_shape_ok(int n, int m, int D) {
  return match_x3_enabled() && n >= 256 && m >= 256 && n % 32 == 0 && m % 32 == 0 && D % 32 == 0 && D >= 64;
}

// K splits of the cost GEMM on the 256 x 256 engine: one workgroup per CU (a workgroup owns a CU's LDS), none
// when there are enough tiles
struct X3CostPlan {
  int tiles, nsplit, kt_per_split;
};
// (The 256 x 128 two-workgroups-per-CU kernel of the convolutions was measured 5 % slower on these contractions -- thousands of
// stages long, no prologue or write-out for a second workgroup to hide: round 4, 156 / 250 us against 145 / 242 us at
// N = 1024, D = 32768 -- and is not offered here.)
inline X3CostPlan x3_plan_cost(int P, int n, int m, int D) {
  X3CostPlan c;
  const int resident = 256;
  c.tiles = ceil_div(n, X3_BM) * ceil_div(m, X3_BN);
  const int nkt = D / X3_BK;
  // The split count that minimises (rounds of resident workgroups) x (granules per split + a workgroup's fixed cost: about four
  // granules of prologue and write-out, ten with a partial tile that is written and read again).  Round 4: until then ceil(256 / tiles) splits -- 12 tiles (a rank's three row
  // slices at N = 1024) became 264 workgroups, i.e. a second round for eight of them and twice the time (376 us, MFMA busy
  // 0.41 in profiles/r04_pmc_kernels_matching_N1024_D32768_rows256_rank.txt).
  const int max_split = nkt / 4 > 0 ? nkt / 4 : 1;            // >= 4 granules (8 stages) per split
  long best_t = -1;
  int best = 1;
  for (int ns = 1; ns <= max_split && ns <= 256; ++ns) {
    const int kt = ceil_div(nkt, ns), real = ceil_div(nkt, kt);
    if (nkt - (real - 1) * kt < 2) continue;                   // (the pipelined kernels need four stages in every split)
    const long t = (long)ceil_div(c.tiles * P * real, resident) * (kt + (real > 1 ? 10 : 4));   // (+ the partial tile's write-out and re-read)
    if (best_t < 0 || t < best_t) { best_t = t; best = real; }
  }
  c.kt_per_split = ceil_div(nkt, best);
  c.nsplit = ceil_div(nkt, c.kt_per_split);
  return c;
}

// cost GEMMs: problem p multiplies rows [xrow[p], +n) by rows [yrow[p], +m) of the stacked operand FP
// (kblocks = D / 16).  nsplit == 1: K written directly; else partial sums (then cost_finish_kernel).
int launch_cost_x3(const u16* FP, long plane, long rows_total, const long* xrow, const long* yrow, const float* diag,
                   int P, int n, int m, int D, float lambda, float* partial_ws, float* K, hipStream_t s) {
  const X3CostPlan cp = x3_plan_cost(P, n, m, D);
  BgArgs b;
  memset(&b, 0, sizeof(b));
  b.Ap = FP; b.Bp = FP; b.pA = plane; b.pB = plane;
  b.hdrA = b.hdrB = x3_hdr(FP);
  b.M = n; b.N = m; b.K = D;
  b.ldc = m;
  b.tiles_m = ceil_div(n, X3_BM); b.tiles_n = ceil_div(m, X3_BN);
  b.kt_per_split = cp.kt_per_split;
  b.rbA = n / 32; b.rbB = m / 32; b.kblocks = D / 16;
  b.ztab = 1;
  for (int p = 0; p < P; ++p) {
    b.zA[p] = x3_row_off(xrow[p], D);
    b.zB[p] = x3_row_off(yrow[p], D);
    b.zC[p] = (long)p * n * m;
    b.zK[p] = D;
    b.epi_diag[p] = diag ? -lambda * diag[p] : 0.f;
  }
  const bool fuse = cp.nsplit == 1;
  b.C = fuse ? K : partial_ws;
  b.sSplit = (long)P * n * m;
  if (fuse) {   // K = -lambda (1 - dot) = lambda * dot - lambda
    b.epi = 1; b.epi_scale = lambda; b.epi_bias = -lambda;
  }
  x3_ensure_lds<wino_bgemm_x3_kernel<true, false>>();
  // (Round 5: the tiles of one K split on one XCD -- a one-dimensional grid, workgroup x -> XCD x % 8 -- were measured on a
  // rank's 12 tiles: each workgroup ran 30 % faster, but 12 tiles x a multiple of 8 splits fill 24 of an XCD's 32 compute
  // units: 153 - 158 us on 192 workgroups against 158 - 160 us on 252 in the linear order, N = 512 6 % slower;
  // profiles/r05_cost_xcd_ab.txt.  Not kept.)
  const dim3 grid(b.tiles_m * b.tiles_n, cp.nsplit, P);
  hipLaunchKernelGGL((wino_bgemm_x3_kernel<true, false>), grid, dim3(X3_THREADS), X3_LDS, s, b);
  OTGAN_CHECK_LAUNCH("cost GEMM (split precision)");
  if (fuse) return OTGAN_OK;
  FinishArgs fa;
  memset(&fa, 0, sizeof(fa));
  fa.ws = partial_ws; fa.nsplit = cp.nsplit; fa.P = P; fa.n = n; fa.m = m;
  fa.lambda = lambda; fa.inv_d = 1.f / (float)D; fa.cost_kind = OTGAN_COST_COSINE; fa.K = K;
  for (int p = 0; p < P; ++p) fa.diag[p] = diag ? diag[p] : 0.f;
  launch_cost_finish(fa, s);
  OTGAN_CHECK_LAUNCH("cost_finish_kernel");
  return OTGAN_OK;
}

// plan application on the t-leading engine.  Block z: out_z[m_begin .. m_end) x D = A_z^T . F_z with
// A_z = rows [arow, +K_z) of the plan operand PA (columns = output rows, cbA = ncolsA / 16) and
// F_z = rows [frow, +K_z) of the stacked feature operand FP.
struct X3ApplyBlock {
  const u16* A;      // PT or PM base
  long arow, frow;
  int K;
  float* out;        // row m_begin of this block's output
};
int launch_apply_x3(const X3ApplyBlock* blk, int nblk, const u16* PA_base, const float* hdrA, long planeA, int ncolsA, const u16* FP,
                    long planeF, int m_begin, int m_count, int D, long ldo, hipStream_t s) {
  BgArgs b;
  memset(&b, 0, sizeof(b));
  b.Ap = PA_base; b.Bp = FP; b.pA = planeA; b.pB = planeF;
  b.hdrA = hdrA; b.hdrB = x3_hdr(FP);
  b.M = m_begin + m_count; b.N = D;
  b.ldc = ldo;
  b.tiles_m = ceil_div(m_count, X3_BM); b.tiles_n = ceil_div(D, X3_BN);
  b.cbA = ncolsA / 16; b.cbB = D / 16;
  b.ztab = 1; b.m_begin = m_begin;
  double flops = 0;
  float* base = blk[0].out;
  for (int z = 0; z < nblk; ++z) {
    b.zA[z] = (blk[z].A - PA_base) + x3_row_off(blk[z].arow, ncolsA);
    b.zB[z] = x3_row_off(blk[z].frow, D);
    b.zC[z] = (blk[z].out - base) - (long)m_begin * ldo;
    b.zK[z] = blk[z].K;
    flops += 2.0 * m_count * (double)blk[z].K * D;
  }
  b.C = base;
  b.kt_per_split = 1 << 28;    // one split: each block contracts over its whole zK
  ProfScope ps(OTGAN_PROF_PLAN_APPLY, flops, 0.0, s);
  x3_ensure_lds<wino_bgemm_x3_kernel<true, true>>();
  const dim3 grid(b.tiles_m * b.tiles_n, 1, nblk);
  hipLaunchKernelGGL((wino_bgemm_x3_kernel<true, true>), grid, dim3(X3_THREADS), X3_LDS, s, b);
  OTGAN_CHECK_LAUNCH("plan application (split precision)");
  return OTGAN_OK;
}

// ---------------------------------------------------------------------------------------
// host-side planning helpers
// ---------------------------------------------------------------------------------------
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

struct CostPlan {
  int tiles, nsplit, kt_per_split;
};
// the one-tile cosine problems (N <= 128) on two fp16 pieces: cost128_h2_kernel
inline bool cost_h2_ok(int n, int m, int D, long ldf, int cost_kind, bool vec) {
  return match_x3_enabled() && cost_kind == OTGAN_COST_COSINE && vec && n <= 128 && m <= 128 && D % 4 == 0 && ldf % 4 == 0;
}
inline CostPlan plan_cost(int P, int n, int m, int D, bool h2 = false) {
  CostPlan c;
  c.tiles = ceil_div(n, 128) * ceil_div(m, 128);
  const int nkt = ceil_div(D, SCfg::BK);
  // K splits.  With at least one tile per CU (256) no split is needed: the GEMM epilogue writes the
  // log-kernel itself (no partial sums in memory at all).  Small problems (N = 128: 6 tiles) need the
  // parallelism: ~3 workgroups per CU (768; 512 = 2 per CU measured 92 vs 67 us at
  // N = 128, D = 32768: the engine hides its staging latency with co-resident workgroups), reduced by cost_finish_kernel.
  // cost128_h2_kernel (h2): 384 -- 1.5 workgroups per CU, two k steps in flight each: 35.8 us at N = 128, D = 32768 against
  // 45 / 35 / 52 at 512 / 768 / 1024 (profiles/r05_cost128_ab.txt); fewer splits = fewer partial sums for cost_finish4_kernel
  const int target = h2 ? 384 : 768;
  int want = c.tiles * P >= 256 ? 1 : ceil_div(target, c.tiles * P);
  if (want < 1) want = 1;
  if (want > nkt) want = nkt;
  c.kt_per_split = ceil_div(nkt, want);
  c.nsplit = ceil_div(nkt, c.kt_per_split);
  return c;
}

struct Carver {
  char* base;
  size_t off, cap;
  Carver(void* b, size_t c) : base((char*)b), off(0), cap(c) {}
  void* take(size_t bytes) {
    void* p = base ? base + off : nullptr;
    off += align_up(bytes, 256);
    return p;
  }
};

int launch_cost(const float* const* X, const float* const* Y, const float* const* xsq,
                const float* const* ysq, const float* diag, int P, int n, int m, int D, long ldf,
                float lambda, int cost_kind, float* partial_ws, float* K, hipStream_t s) {
  CostArgs ca;
  memset(&ca, 0, sizeof(ca));
  bool vec = (ldf % 4 == 0);
  for (int p = 0; p < P; ++p) {
    ca.X[p] = X[p];
    ca.Y[p] = Y[p];
    vec = vec && aligned16(X[p]) && aligned16(Y[p]);
  }
  // (the workspace queries size the partial sums with the fp32 kernel's split count, which is never smaller)
  const CostPlan cp = plan_cost(P, n, m, D, cost_h2_ok(n, m, D, ldf, cost_kind, vec));
  ca.P = P; ca.n = n; ca.m = m; ca.D = D; ca.ldf = ldf;
  ca.kt_per_split = cp.kt_per_split;
  ca.ws = partial_ws;
  ca.K = K; ca.lambda = lambda; ca.inv_d = 1.f / (float)D; ca.cost_kind = cost_kind;
  for (int p = 0; p < P; ++p) {
    ca.xsq[p] = xsq ? xsq[p] : nullptr;
    ca.ysq[p] = ysq ? ysq[p] : nullptr;
    ca.diag[p] = diag ? diag[p] : 0.f;
  }
  const bool fuse = cp.nsplit == 1;
  dim3 grid(cp.tiles, cp.nsplit, P);
  {
    ProfScope ps(OTGAN_PROF_COST_GEMM, 2.0 * P * n * (double)m * D,
                 4.0 * P * ((double)n + m) * D, s);
    if (cost_h2_ok(n, m, D, ldf, cost_kind, vec)) {
      const dim3 g2(8 * ceil_div(cp.nsplit, 8) * P);
      if (fuse) hipLaunchKernelGGL(cost128_h2_kernel<true>, g2, dim3(256), 0, s, ca);
      else hipLaunchKernelGGL(cost128_h2_kernel<false>, g2, dim3(256), 0, s, ca);
    } else if (fuse) {
      if (vec) hipLaunchKernelGGL((cost_partial_kernel<true, true>), grid, dim3(256), 0, s, ca);
      else hipLaunchKernelGGL((cost_partial_kernel<false, true>), grid, dim3(256), 0, s, ca);
    } else {
      if (vec) hipLaunchKernelGGL((cost_partial_kernel<true, false>), grid, dim3(256), 0, s, ca);
      else hipLaunchKernelGGL((cost_partial_kernel<false, false>), grid, dim3(256), 0, s, ca);
    }
  }
  OTGAN_CHECK_LAUNCH("cost_partial_kernel");
  if (fuse) return OTGAN_OK;
  FinishArgs fa;
  memset(&fa, 0, sizeof(fa));
  fa.ws = partial_ws; fa.nsplit = cp.nsplit; fa.P = P; fa.n = n; fa.m = m;
  fa.lambda = lambda; fa.inv_d = 1.f / (float)D; fa.cost_kind = cost_kind; fa.K = K;
  for (int p = 0; p < P; ++p) {
    fa.xsq[p] = xsq ? xsq[p] : nullptr;
    fa.ysq[p] = ysq ? ysq[p] : nullptr;
    fa.diag[p] = diag ? diag[p] : 0.f;
  }
  launch_cost_finish(fa, s);
  OTGAN_CHECK_LAUNCH("cost_finish_kernel");
  return OTGAN_OK;
}

int launch_sinkhorn(const float* K, int P, int n, int m, int iters, float lambda, float* plan,
                    float* planT, double* stats, float* fg_ws, hipStream_t s) {
  ProfScope ps(OTGAN_PROF_SINKHORN, 0.0, 0.0, s);
  if (n <= 128 && m <= 128) {
    hipLaunchKernelGGL(sinkhorn_small_kernel, dim3(P), dim3(512), 0, s, K, n, m, iters,
                       1.f / lambda, plan, planT, stats, lin_ctl());
    OTGAN_CHECK_LAUNCH("sinkhorn_small_kernel");
    return OTGAN_OK;
  }
  float* f = fg_ws;
  float* g = fg_ws + (size_t)P * n;
  hipMemsetAsync(stats, 0, sizeof(double) * 4 * P, s);
  if (n == m && n <= 1024) {
    // persistent panel kernel: all P*R workgroups must be co-resident (<= 192 of 256 CUs; checked)
    const int rpw = n <= 256 ? 128 : (n <= 512 ? 64 : 32);
    PanelArgs a;
    memset(&a, 0, sizeof(a));
    a.K = K; a.N = n; a.iters = iters; a.R = ceil_div(n, rpw);
    a.inv_lambda = 1.f / lambda;
    a.f = reinterpret_cast<unsigned long long*>(fg_ws);
    a.g = a.f + (size_t)P * n;
    a.fail = reinterpret_cast<unsigned*>(a.g + (size_t)P * n);
    a.plan = plan; a.planT = planT; a.stats = stats;
    hipMemsetAsync(fg_ws, 0, sizeof(unsigned long long) * 2 * (size_t)P * n + sizeof(unsigned), s);
    const bool launched = rpw == 128 ? launch_panel<8>(a, P, s)
                          : rpw == 64 ? launch_panel<16>(a, P, s) : launch_panel<32>(a, P, s);
    if (launched) {
      OTGAN_CHECK_LAUNCH("sinkhorn_panel_kernel");
      return OTGAN_OK;
    }
    // not enough CUs to keep the whole grid resident: multi-launch path below
  }
  hipMemsetAsync(g, 0, sizeof(float) * (size_t)P * m, s);
  const dim3 grow(ceil_div(n, 4), P), gcol(ceil_div(m, 64), P);
  for (int it = 0; it < iters; ++it) {
    hipLaunchKernelGGL(sinkhorn_row_kernel, grow, dim3(256), 0, s, K, n, m, g, f);
    hipLaunchKernelGGL(sinkhorn_col_kernel, gcol, dim3(256), 0, s, K, n, m, f, g);
  }
  hipLaunchKernelGGL(sinkhorn_final_kernel, grow, dim3(256), 0, s, K, n, m, g, 1.f / lambda,
                     plan, planT, stats);
  OTGAN_CHECK_LAUNCH("sinkhorn general kernels");
  return OTGAN_OK;
}

// bounded_plans: the plan operands are Sinkhorn plans (entries <= 1: what plan_apply128_h2_kernel's scale assumes)
int launch_apply(const ApplyBlock* blocks, int nblocks, int max_rows, int D, long ldf, long ldo,
                 hipStream_t s, bool bounded_plans = true) {
  ApplyArgs aa;
  memset(&aa, 0, sizeof(aa));
  bool vec = (ldf % 4 == 0);
  double flops = 0;
  for (int b = 0; b < nblocks; ++b) {
    aa.b[b] = blocks[b];
    for (int t = 0; t < blocks[b].nterms; ++t) {
      vec = vec && aligned16(blocks[b].t[t].plan) && aligned16(blocks[b].t[t].feat) &&
            (blocks[b].t[t].ldp % 4 == 0);
      flops += 2.0 * blocks[b].rows * (double)blocks[b].t[t].kdim * D;
    }
  }
  // one-tile problems on the fp16 matrix pipe: every block max_rows <= 128 rows, one contraction length (a multiple of 16)
  // and one plan stride for all terms
  bool h2 = bounded_plans && match_x3_enabled() && vec && max_rows <= 128 && D % 4 == 0 && D >= 4 && nblocks > 0;
  const int kdim = blocks[0].t[0].kdim;
  const long ldp = blocks[0].t[0].ldp;
  for (int b = 0; b < nblocks && h2; ++b) {
    h2 = blocks[b].rows == max_rows && blocks[b].nterms >= 1 && blocks[b].nterms <= 3;
    for (int t = 0; t < blocks[b].nterms && h2; ++t) h2 = blocks[b].t[t].kdim == kdim && blocks[b].t[t].ldp == ldp;
  }
  if (h2 && kdim % 16 == 0 && kdim >= 16) {
    const float pscale = kCostH2Scale;                    // (a plan's rows and columns sum to 1 or less: entries <= 1)
    ProfScope ps(OTGAN_PROF_PLAN_APPLY, flops, 0.0, s);
    for (int nt = 1; nt <= 3; ++nt) {
      ApplyH2Args ha;
      memset(&ha, 0, sizeof(ha));
      ha.rows = max_rows; ha.kdim = kdim; ha.nterms = nt; ha.D = D; ha.tiles_d = ceil_div(D, 128);
      ha.ldp = ldp; ha.ldf = ldf; ha.ldo = ldo;
      auto flush = [&]() {
        if (!ha.nblk) return;
        hipLaunchKernelGGL(plan_apply128_h2_kernel, dim3(8 * ceil_div(ha.tiles_d, 8) * ha.nblk), dim3(256), 0, s, ha);
        ha.nblk = 0;
      };
      for (int b = 0; b < nblocks; ++b) {
        if (blocks[b].nterms != nt) continue;
        ApplyH2Block& hb = ha.b[ha.nblk++];
        hb.out = blocks[b].out;
        hb.out_scale = blocks[b].alpha / (pscale * kCostH2Scale);
        float c = 1.f;                                    // product of the rescale factors applied after term t
        for (int t = nt - 1; t >= 0; --t) {
          hb.t[t] = ApplyH2Term{blocks[b].t[t].plan, blocks[b].t[t].feat, pscale * c};
          const float rs = blocks[b].rescale[t];
          if (rs != 0.f && rs != 1.f) c *= rs;
        }
        if (ha.nblk == 4) flush();
      }
      flush();
    }
    OTGAN_CHECK_LAUNCH("plan_apply128_h2_kernel");
    return OTGAN_OK;
  }
  aa.D = D; aa.ldf = ldf; aa.ldo = ldo;
  aa.tiles_d = ceil_div(D, 128);
  dim3 grid(aa.tiles_d * ceil_div(max_rows, 128), nblocks);
  {
    ProfScope ps(OTGAN_PROF_PLAN_APPLY, flops, 0.0, s);
    if (vec) hipLaunchKernelGGL(plan_apply_kernel<true>, grid, dim3(256), 0, s, aa);
    else hipLaunchKernelGGL(plan_apply_kernel<false>, grid, dim3(256), 0, s, aa);
  }
  OTGAN_CHECK_LAUNCH("plan_apply_kernel");
  return OTGAN_OK;
}

int launch_distance(const float* a, const float* b, const float* aa, const float* bb,
                    const float* ab, long total, double denom, double* dist, double* scratch3,
                    hipStream_t s, const double* stats = nullptr, int P = 0) {
  hipMemsetAsync(scratch3, 0, 3 * sizeof(double), s);
  long blocks = ceil_div_l(total, 256 * 8);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(dot3_kernel, dim3((int)blocks), dim3(256), 0, s, a, b, aa, bb, ab, total,
                     scratch3);
  hipLaunchKernelGGL(distance_finalize_kernel, dim3(1), dim3(1), 0, s, scratch3, denom, dist, stats, P);
  OTGAN_CHECK_LAUNCH("distance kernels");
  return OTGAN_OK;
}

// workspace layout shared by the size query (base == nullptr) and the launchers
struct MatchWs {
  float* sq_a;     // [rows_a] 0.5*mean(x^2) (toy cost)
  float* sq_b;
  float* partial;  // [nsplit][P][n][m]
  float* K;        // [P][n][m]
  float* plan;     // [P][n][m]
  float* planT;    // [P][m][n]
  float* fg;       // [P][n+m]
  double* stats;   // [P][4]
  double* dot3;    // [3]
  // split-precision path (x3_shape_ok): stacked feature operand [fa; fb] and the plans as t-leading operands
  bool x3;
  u16* FP;         // X3_NP planes x [2 * feat_rows][D], each operand behind its header + amax record (x3_hdr)
  u16* PT;         // X3_NP planes x [P * n][n]   (transposed plans: A operand of M . F)
  u16* PM;         // X3_NP planes x [P * n][n]   (plans: A operand of M^T . F)
  long planeF, planeP;
  size_t bytes;
};
MatchWs carve_match(void* base, size_t cap, int P, int n, int D, int feat_rows, bool grad = false) {
  Carver c(base, cap);
  MatchWs w;
  const CostPlan cp = plan_cost(P, n, n, D);
  const size_t pnm = (size_t)P * n * n;
  w.x3 = x3_shape_ok(n, n, D);
  int nsplit = cp.nsplit;
  if (w.x3 && x3_plan_cost(P, n, n, D).nsplit > nsplit) nsplit = x3_plan_cost(P, n, n, D).nsplit;
  w.sq_a = (float*)c.take(sizeof(float) * feat_rows);
  w.sq_b = (float*)c.take(sizeof(float) * feat_rows);
  // grad variant (otgan_matching_two_batch_grad_f32): feature stack [a1 b1 b2 a2 a1 b1] (3 n-row blocks more than the
  // four halves) and ONE plan operand of twelve n-row blocks (PT; PM unused)
  w.planeF = (long)x3_plane_elems((grad ? 3 : 2) * (size_t)feat_rows, D);
  w.planeP = (long)x3_plane_elems((grad ? 2 : 1) * (size_t)P * n, n);
  auto operand = [&](bool on, long plane) -> u16* {     // header + record, then the planes
    char* p = (char*)c.take(on ? kX3HdrBytes + sizeof(u16) * X3_NP * (size_t)plane : 0);
    return (u16*)(p ? p + kX3HdrBytes : (char*)nullptr);
  };
  w.FP = operand(w.x3, w.planeF);
  w.PT = operand(w.x3, w.planeP);
  w.PM = operand(w.x3 && !grad, w.planeP);
  w.partial = (float*)c.take(sizeof(float) * pnm * nsplit);
  w.K = (float*)c.take(sizeof(float) * pnm);
  w.plan = (float*)c.take(sizeof(float) * pnm);
  w.planT = (float*)c.take(sizeof(float) * pnm);
  w.fg = (float*)c.take(sizeof(unsigned long long) * (size_t)P * 2 * n + 64);  // tagged potentials + fail word
  w.stats = (double*)c.take(sizeof(double) * 4 * P);
  w.dot3 = (double*)c.take(sizeof(double) * 4);
  w.bytes = c.off;
  return w;
}

// stacked feature operand [fa; fb] (each `rows` rows) for the split-precision GEMMs
void x3_split_features(const MatchWs& w, const float* fa, const float* fb, int rows, int D, long ldf, hipStream_t s) {
  SplitSrc ss;
  memset(&ss, 0, sizeof(ss));
  ss.n = 2;
  ss.src[0] = fa; ss.ld[0] = ldf; ss.row0[0] = 0; ss.scale[0] = 1.f;
  ss.src[1] = fb; ss.ld[1] = ldf; ss.row0[1] = rows; ss.scale[1] = 1.f;
  x3_split(ss, rows, D, w.FP, w.planeF, kX3FeatureExp, s);
}

// plans / transposed plans of P problems -> t-leading operands; order[i] = problem stored at rows [i n, (i+1) n)
void x3_split_plans(const MatchWs& w, const float* plan, const float* planT, int P, int n, const int* orderM,
                    const float* alpha, hipStream_t s) {
  SplitSrc st, sm;
  memset(&st, 0, sizeof(st));
  memset(&sm, 0, sizeof(sm));
  st.n = sm.n = P;
  for (int i = 0; i < P; ++i) {
    st.src[i] = planT + (size_t)i * n * n; st.ld[i] = n; st.row0[i] = (long)i * n; st.scale[i] = alpha[i];
    const int p = orderM[i];
    sm.src[i] = plan + (size_t)p * n * n; sm.ld[i] = n; sm.row0[i] = (long)i * n; sm.scale[i] = alpha[p];
  }
  // both operands behind ONE header (w.PT's): a launch of the plan application reads blocks of either
  X3SplitJob jobs[2] = {{st, n, n, w.PT, w.planeP}, {sm, n, n, w.PM, w.planeP}};
  const int e0 = x3_plan_exp(st) > x3_plan_exp(sm) ? x3_plan_exp(st) : x3_plan_exp(sm);
  x3_split_group(jobs, 2, x3_hdr(w.PT), e0, s);
}

}  // namespace

// =======================================================================================
// C ABI
// =======================================================================================
extern "C" {

size_t otgan_matching_workspace_bytes(int mode, int rows, int D) {
  if (rows <= 0 || D <= 0) return 0;
  if (mode == OTGAN_MATCH_TWO_BATCH) return carve_match(nullptr, 0, 6, rows, D, 2 * rows).bytes;
  return carve_match(nullptr, 0, 3, rows, D, rows).bytes;
}

int otgan_matching_two_batch_f32(const float* fa, const float* fb, int N, int D, long ldf,
                                 float lambda, int iters, int cost_kind, float* f_aa,
                                 float* f_bb, float* f_ab, float* f_ba, long ldo,
                                 float* entropy, double* dist, double* stats, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  OTGAN_CHECK_ARG(fa && fb && f_aa && f_bb && f_ab && f_ba && entropy && dist, "null pointer");
  OTGAN_CHECK_ARG(N > 0 && D > 0 && ldf >= D && ldo >= D && iters >= 0, "bad sizes N=%d D=%d", N, D);
  OTGAN_CHECK_ARG(cost_kind == OTGAN_COST_COSINE || cost_kind == OTGAN_COST_SQEUCLID_MEAN,
                  "unknown cost kind %d", cost_kind);
  // the distance reduction walks the [2N, D] arrays linearly
  OTGAN_CHECK_ARG(ldf == D && ldo == D, "feature arrays must be contiguous (ld == D)");
  hipStream_t s = (hipStream_t)stream;
  MatchWs w = carve_match(workspace, workspace_bytes, 6, N, D, 2 * N);
  if (!workspace || workspace_bytes < w.bytes) {
    otgan_set_error("workspace too small: need %zu bytes, got %zu", w.bytes, workspace_bytes);
    return OTGAN_ERR_WORKSPACE;
  }
  const float *fa1 = fa, *fa2 = fa + (long)N * ldf, *fb1 = fb, *fb2 = fb + (long)N * ldf;
  // problem order of the reference (matching.py:41-43): a1a2, b2b1, a1b1, a1b2, a2b1, a2b2
  const float* X[6] = {fa1, fb2, fa1, fa1, fa2, fa2};
  const float* Y[6] = {fa2, fb1, fb1, fb2, fb1, fb2};
  const float *xsq[6], *ysq[6];
  if (cost_kind == OTGAN_COST_SQEUCLID_MEAN) {
    hipLaunchKernelGGL(row_halfmeansq_kernel, dim3(ceil_div(2 * N, 4)), dim3(256), 0, s, fa, ldf,
                       2 * N, D, w.sq_a);
    hipLaunchKernelGGL(row_halfmeansq_kernel, dim3(ceil_div(2 * N, 4)), dim3(256), 0, s, fb, ldf,
                       2 * N, D, w.sq_b);
    const float *sa1 = w.sq_a, *sa2 = w.sq_a + N, *sb1 = w.sq_b, *sb2 = w.sq_b + N;
    const float* xs[6] = {sa1, sb2, sa1, sa1, sa2, sa2};
    const float* ys[6] = {sa2, sb1, sb1, sb2, sb1, sb2};
    memcpy(xsq, xs, sizeof(xs));
    memcpy(ysq, ys, sizeof(ys));
  }
  const bool x3 = w.x3 && cost_kind == OTGAN_COST_COSINE && ldf % 4 == 0 && aligned16(fa) && aligned16(fb);
  int rc;
  if (x3) {
    // stacked rows: a1 [0,N) a2 [N,2N) b1 [2N,3N) b2 [3N,4N); problems a1a2, b2b1, a1b1, a1b2, a2b1, a2b2
    const long xrow[6] = {0, 3L * N, 0, 0, N, N};
    const long yrow[6] = {N, 2L * N, 2L * N, 3L * N, 2L * N, 3L * N};
    ProfScope ps(OTGAN_PROF_COST_GEMM, 12.0 * N * (double)N * D, 16.0 * N * (double)D, s);
    x3_split_features(w, fa, fb, 2 * N, D, ldf, s);
    rc = launch_cost_x3(w.FP, w.planeF, 4L * N, xrow, yrow, nullptr, 6, N, N, D, lambda, w.partial, w.K, s);
  } else {
    rc = launch_cost(X, Y, cost_kind == OTGAN_COST_SQEUCLID_MEAN ? xsq : nullptr,
                     cost_kind == OTGAN_COST_SQEUCLID_MEAN ? ysq : nullptr, nullptr, 6, N, N, D,
                     ldf, lambda, cost_kind, w.partial, w.K, s);
  }
  if (rc) return rc;
  rc = launch_sinkhorn(w.K, 6, N, N, iters, lambda, w.plan, w.planT, w.stats, w.fg, s);
  if (rc) return rc;
  const size_t nn = (size_t)N * N;
  const float *M0 = w.plan, *M1 = w.plan + nn, *M2 = w.plan + 2 * nn, *M3 = w.plan + 3 * nn,
              *M4 = w.plan + 4 * nn, *M5 = w.plan + 5 * nn;
  const float *T0 = w.planT, *T1 = w.planT + nn, *T2 = w.planT + 2 * nn, *T3 = w.planT + 3 * nn,
              *T4 = w.planT + 4 * nn, *T5 = w.planT + 5 * nn;
  // matching.py:64-83.  Eight [N,D] output blocks, each one or two plan.feature products.
  ApplyBlock blk[8];
  memset(blk, 0, sizeof(blk));
  auto set1 = [&](int i, float* out, const float* P0, const float* F0, float alpha) {
    blk[i].out = out; blk[i].rows = N; blk[i].nterms = 1; blk[i].alpha = alpha;
    blk[i].t[0] = ApplyTerm{P0, F0, (long)N, N};
  };
  auto set2 = [&](int i, float* out, const float* P0, const float* F0, const float* P1,
                  const float* F1, float alpha) {
    blk[i].out = out; blk[i].rows = N; blk[i].nterms = 2; blk[i].alpha = alpha;
    blk[i].t[0] = ApplyTerm{P0, F0, (long)N, N};
    blk[i].t[1] = ApplyTerm{P1, F1, (long)N, N};
  };
  const long half = (long)N * ldo;
  set1(0, f_aa, M0, fa2, 1.f);                  // a1 <- M_a1a2 . a2          (:64)
  set1(1, f_aa + half, T0, fa1, 1.f);           // a2 <- M_a1a2^T . a1        (:70)
  set1(2, f_bb, T1, fb2, 1.f);                  // b1 <- M_b2b1^T . b2        (:65)
  set1(3, f_bb + half, M1, fb1, 1.f);           // b2 <- M_b2b1 . b1          (:71)
  set2(4, f_ab, M2, fb1, M3, fb2, 0.5f);        // a1 <- (M_a1b1.b1 + M_a1b2.b2)/2   (:66,67,80)
  set2(5, f_ab + half, M4, fb1, M5, fb2, 0.5f); // a2 <- (M_a2b1.b1 + M_a2b2.b2)/2   (:68,69,80)
  set2(6, f_ba, T2, fa1, T4, fa2, 0.5f);        // b1 <- (M_a1b1^T.a1 + M_a2b1^T.a2)/2 (:72,74,82)
  set2(7, f_ba + half, T3, fa1, T5, fa2, 0.5f); // b2 <- (M_a1b2^T.a1 + M_a2b2^T.a2)/2 (:73,75,82)
  if (x3 && ldo % 4 == 0) {
    // PT rows: T0 T1 T2 T3 T4 T5;  PM rows: M0 M1 M2 M4 M3 M5 (the pairs contracted together are adjacent)
    const int orderM[6] = {0, 1, 2, 4, 3, 5};
    const float alpha[6] = {1.f, 1.f, 0.5f, 0.5f, 0.5f, 0.5f};
    x3_split_plans(w, w.plan, w.planT, 6, N, orderM, alpha, s);
    const long n1 = N;
    const X3ApplyBlock xb[8] = {
        {w.PT, 0 * n1, 1 * n1, N, f_aa},            // a1 <- M0 . a2
        {w.PM, 0 * n1, 0 * n1, N, f_aa + half},     // a2 <- M0^T . a1
        {w.PM, 1 * n1, 3 * n1, N, f_bb},            // b1 <- M1^T . b2
        {w.PT, 1 * n1, 2 * n1, N, f_bb + half},     // b2 <- M1 . b1
        {w.PT, 2 * n1, 2 * n1, 2 * N, f_ab},        // a1 <- (M2 . b1 + M3 . b2) / 2
        {w.PT, 4 * n1, 2 * n1, 2 * N, f_ab + half}, // a2 <- (M4 . b1 + M5 . b2) / 2
        {w.PM, 2 * n1, 0 * n1, 2 * N, f_ba},        // b1 <- (M2^T . a1 + M4^T . a2) / 2
        {w.PM, 4 * n1, 0 * n1, 2 * N, f_ba + half}, // b2 <- (M3^T . a1 + M5^T . a2) / 2
    };
    rc = launch_apply_x3(xb, 8, w.PT < w.PM ? w.PT : w.PM, x3_hdr(w.PT), w.planeP, N, w.FP, w.planeF, 0, N, D, ldo, s);
  } else {
    rc = launch_apply(blk, 8, N, D, ldf, ldo, s);
  }
  if (rc) return rc;
  hipLaunchKernelGGL(entropy_finalize_kernel, dim3(1), dim3(1), 0, s, w.stats, 6, N, entropy);
  const double denom = (cost_kind == OTGAN_COST_COSINE) ? 2.0 * (2.0 * N)          // matching.py:152
                                                        : 2.0 * (2.0 * N) * (double)D;  // matching_cpu.py:158-163
  rc = launch_distance(fa, fb, f_aa, f_bb, f_ab, (long)2 * N * D, denom, dist, w.dot3, s, w.stats, 6);
  if (rc) return rc;
  if (stats) hipMemcpyAsync(stats, w.stats, sizeof(double) * 24, hipMemcpyDeviceToDevice, s);
  return OTGAN_OK;
}

int otgan_matching_two_batch_rows_f32(const float* fa, const float* fb, int N, int D, long ldf,
                                      float lambda, int iters, int row_begin, int row_count,
                                      const float* K_pre, float* f_aa, float* f_bb, float* f_ab,
                                      float* f_ba, long ldo, float* entropy, double* dist,
                                      double* stats, void* workspace, size_t workspace_bytes,
                                      void* stream) {
  OTGAN_CHECK_ARG(fa && fb && f_aa && f_bb && f_ab && f_ba && entropy && dist, "null pointer");
  OTGAN_CHECK_ARG(N > 0 && D > 0 && ldf >= D && ldo >= D && iters >= 0, "bad sizes N=%d D=%d", N, D);
  OTGAN_CHECK_ARG(row_begin >= 0 && row_count > 0 && row_begin + row_count <= 2 * N,
                  "row range [%d, %d) outside [0, %d)", row_begin, row_begin + row_count, 2 * N);
  const int half = row_begin / N;
  OTGAN_CHECK_ARG((row_begin + row_count - 1) / N == half, "row range must not straddle the two mini-batches");
  hipStream_t s = (hipStream_t)stream;
  MatchWs w = carve_match(workspace, workspace_bytes, 6, N, D, 2 * N);
  if (!workspace || workspace_bytes < w.bytes) {
    otgan_set_error("workspace too small: need %zu bytes, got %zu", w.bytes, workspace_bytes);
    return OTGAN_ERR_WORKSPACE;
  }
  const float *fa1 = fa, *fa2 = fa + (long)N * ldf, *fb1 = fb, *fb2 = fb + (long)N * ldf;
  const float* X[6] = {fa1, fb2, fa1, fa1, fa2, fa2};
  const float* Y[6] = {fa2, fb1, fb1, fb2, fb1, fb2};
  int rc = OTGAN_OK;
  const float* Kuse = K_pre;
  const bool x3 = w.x3 && ldf % 4 == 0 && ldo % 4 == 0 && aligned16(fa) && aligned16(fb) && row_begin % 16 == 0;
  if (x3) x3_split_features(w, fa, fb, 2 * N, D, ldf, s);     // cost operand and B operand of the plan application
  if (!K_pre) {
    if (x3) {
      const long xrow[6] = {0, 3L * N, 0, 0, N, N};
      const long yrow[6] = {N, 2L * N, 2L * N, 3L * N, 2L * N, 3L * N};
      ProfScope ps(OTGAN_PROF_COST_GEMM, 12.0 * N * (double)N * D, 16.0 * N * (double)D, s);
      rc = launch_cost_x3(w.FP, w.planeF, 4L * N, xrow, yrow, nullptr, 6, N, N, D, lambda, w.partial, w.K, s);
    } else {
      rc = launch_cost(X, Y, nullptr, nullptr, nullptr, 6, N, N, D, ldf, lambda, OTGAN_COST_COSINE,
                       w.partial, w.K, s);
    }
    if (rc) return rc;
    Kuse = w.K;
  }
  rc = launch_sinkhorn(Kuse, 6, N, N, iters, lambda, w.plan, w.planT, w.stats, w.fg, s);
  if (rc) return rc;
  const size_t nn = (size_t)N * N;
  const long ro = (long)(row_begin - half * N) * N;  // first plan row of the range
  const float* M[6];
  const float* T[6];
  for (int p = 0; p < 6; ++p) {
    M[p] = w.plan + p * nn + ro;
    T[p] = w.planT + p * nn + ro;
  }
  ApplyBlock blk[4];
  memset(blk, 0, sizeof(blk));
  auto set = [&](int i, float* out, const float* P0, const float* F0, const float* P1, const float* F1,
                 float alpha) {
    blk[i].out = out; blk[i].rows = row_count; blk[i].nterms = P1 ? 2 : 1; blk[i].alpha = alpha;
    blk[i].t[0] = ApplyTerm{P0, F0, (long)N, N};
    if (P1) blk[i].t[1] = ApplyTerm{P1, F1, (long)N, N};
  };
  if (half == 0) {  // rows of a1 / b1  (matching.py:64-67,72,74)
    set(0, f_aa, M[0], fa2, nullptr, nullptr, 1.f);
    set(1, f_bb, T[1], fb2, nullptr, nullptr, 1.f);
    set(2, f_ab, M[2], fb1, M[3], fb2, 0.5f);
    set(3, f_ba, T[2], fa1, T[4], fa2, 0.5f);
  } else {          // rows of a2 / b2  (matching.py:68-71,73,75)
    set(0, f_aa, T[0], fa1, nullptr, nullptr, 1.f);
    set(1, f_bb, M[1], fb1, nullptr, nullptr, 1.f);
    set(2, f_ab, M[4], fb1, M[5], fb2, 0.5f);
    set(3, f_ba, T[3], fa1, T[5], fa2, 0.5f);
  }
  if (x3) {
    const int orderM[6] = {0, 1, 2, 4, 3, 5};
    const float alpha[6] = {1.f, 1.f, 0.5f, 0.5f, 0.5f, 0.5f};
    x3_split_plans(w, w.plan, w.planT, 6, N, orderM, alpha, s);
    const long n1 = N;
    const int r0 = row_begin - half * N;
    X3ApplyBlock xb[4];
    if (half == 0) {  // rows of a1 / b1
      xb[0] = {w.PT, 0 * n1, 1 * n1, N, f_aa};         // M0 . a2
      xb[1] = {w.PM, 1 * n1, 3 * n1, N, f_bb};         // M1^T . b2
      xb[2] = {w.PT, 2 * n1, 2 * n1, 2 * N, f_ab};     // (M2 . b1 + M3 . b2) / 2
      xb[3] = {w.PM, 2 * n1, 0 * n1, 2 * N, f_ba};     // (M2^T . a1 + M4^T . a2) / 2
    } else {          // rows of a2 / b2
      xb[0] = {w.PM, 0 * n1, 0 * n1, N, f_aa};         // M0^T . a1
      xb[1] = {w.PT, 1 * n1, 2 * n1, N, f_bb};         // M1 . b1
      xb[2] = {w.PT, 4 * n1, 2 * n1, 2 * N, f_ab};     // (M4 . b1 + M5 . b2) / 2
      xb[3] = {w.PM, 4 * n1, 0 * n1, 2 * N, f_ba};     // (M3^T . a1 + M5^T . a2) / 2
    }
    rc = launch_apply_x3(xb, 4, w.PT < w.PM ? w.PT : w.PM, x3_hdr(w.PT), w.planeP, N, w.FP, w.planeF, r0, row_count, D, ldo, s);
  } else {
    rc = launch_apply(blk, 4, row_count, D, ldf, ldo, s);
  }
  if (rc) return rc;
  hipLaunchKernelGGL(entropy_finalize_kernel, dim3(1), dim3(1), 0, s, w.stats, 6, N, entropy);
  hipLaunchKernelGGL(closed_form_distance_kernel, dim3(1), dim3(1), 0, s, w.stats, N, dist);
  OTGAN_CHECK_LAUNCH("matching finalize");
  if (stats) hipMemcpyAsync(stats, w.stats, sizeof(double) * 24, hipMemcpyDeviceToDevice, s);
  return OTGAN_OK;
}

// ---- training-mode matching: the injected gradients directly (round 3) ---------------------------------------
// The reference only ever consumes the DIFFERENCES of the matched features (train.py:111,125-126: grad_ys =
// features_a_a - features_a_b for the generated shards, features_b_b - features_b_a for the data shards) and the
// distance.  Per half-batch a difference is one three-term plan application,
//   g(a1) = M0 a2 - (M2 b1 + M3 b2)/2      g(a2) = M0^T a1 - (M4 b1 + M5 b2)/2
//   g(b1) = M1^T b2 - (M2^T a1 + M4^T a2)/2  g(b2) = M1 b1 - (M3^T a1 + M5^T a2)/2
// (problems a1a2, b2b1, a1b1, a1b2, a2b1, a2b2 = M0 .. M5), so four output blocks (two on the five generator steps out
// of six, which need no data-side gradient) replace eight, the four [2N, D] matched arrays are never written, the two
// subtractions disappear, and the distance comes from the Sinkhorn kernel's statistics (closed form) instead of a pass
// over five [2N, D] arrays.
static int matching_grad_impl(const float* fa, const float* fb, int N, int D, long ldf, float lambda, int iters,
                              int row_begin, int row_count, const float* K_pre, float* grad_a, float* grad_b, long ldo,
                              float* entropy, double* dist, double* stats, void* workspace, size_t workspace_bytes,
                              void* stream, const void* stack = nullptr) {
  OTGAN_CHECK_ARG((stack || (fa && fb)) && grad_a && entropy && dist, "null pointer");
  OTGAN_CHECK_ARG(N > 0 && D > 0 && ldf >= D && ldo >= D && iters >= 0, "bad sizes N=%d D=%d", N, D);
  OTGAN_CHECK_ARG(row_begin >= 0 && row_count > 0 && row_begin + row_count <= 2 * N,
                  "row range [%d, %d) outside [0, %d)", row_begin, row_begin + row_count, 2 * N);
  const bool full = row_begin == 0 && row_count == 2 * N;
  const int half = row_begin / N;
  OTGAN_CHECK_ARG(full || (row_begin + row_count - 1) / N == half, "row range must not straddle the two mini-batches");
  hipStream_t s = (hipStream_t)stream;
  MatchWs w = carve_match(workspace, workspace_bytes, 6, N, D, 2 * N, true);
  if (!workspace || workspace_bytes < w.bytes) {
    otgan_set_error("workspace too small: need %zu bytes, got %zu", w.bytes, workspace_bytes);
    return OTGAN_ERR_WORKSPACE;
  }
  const float *fa1 = fa, *fa2 = fa + (long)N * ldf, *fb1 = fb, *fb2 = fb + (long)N * ldf;
  const float* X[6] = {fa1, fb2, fa1, fa1, fa2, fa2};
  const float* Y[6] = {fa2, fb1, fb1, fb2, fb1, fb2};
  int rc = OTGAN_OK;
  const bool x3 = w.x3 && ldf % 4 == 0 && ldo % 4 == 0 && aligned16(fa) && aligned16(fb) && aligned16(grad_a) &&
                  aligned16(grad_b) && row_begin % 16 == 0;
  const int nstack = grad_b ? 6 : 4;
  if (stack) {
    // the caller split the blocks this call reads already (otgan_matching_stack_split_f32): same layout, its own buffer
    OTGAN_CHECK_ARG(x3 && K_pre, "a feature stack needs the split-precision engine's shapes / alignment and precomputed log-kernels");
    w.FP = (u16*)((char*)const_cast<void*>(stack) + kX3HdrBytes);
  } else if (x3) {
    // stacked feature operand [a1 b1 b2 a2 (a1 b1)]: every difference contracts over three ADJACENT blocks
    SplitSrc ss;
    memset(&ss, 0, sizeof(ss));
    const float* blocks[6] = {fa1, fb1, fb2, fa2, fa1, fb1};
    ss.n = nstack;
    for (int i = 0; i < nstack; ++i) { ss.src[i] = blocks[i]; ss.ld[i] = ldf; ss.row0[i] = (long)i * N; ss.scale[i] = 1.f; }
    x3_split(ss, N, D, w.FP, w.planeF, kX3FeatureExp, s);
  }
  const float* Kuse = K_pre;
  if (!K_pre) {
    if (x3) {
      // rows of the stack: a1 0, b1 N, b2 2N, a2 3N
      const long xrow[6] = {0, 2L * N, 0, 0, 3L * N, 3L * N};
      const long yrow[6] = {3L * N, N, N, 2L * N, N, 2L * N};
      ProfScope ps(OTGAN_PROF_COST_GEMM, 12.0 * N * (double)N * D, 16.0 * N * (double)D, s);
      rc = launch_cost_x3(w.FP, w.planeF, (long)nstack * N, xrow, yrow, nullptr, 6, N, N, D, lambda, w.partial, w.K, s);
    } else {
      rc = launch_cost(X, Y, nullptr, nullptr, nullptr, 6, N, N, D, ldf, lambda, OTGAN_COST_COSINE, w.partial, w.K, s);
    }
    if (rc) return rc;
    Kuse = w.K;
  }
  rc = launch_sinkhorn(Kuse, 6, N, N, iters, lambda, w.plan, w.planT, w.stats, w.fg, s);
  if (rc) return rc;
  const size_t nn = (size_t)N * N;
  const float *M[6], *T[6];
  for (int p = 0; p < 6; ++p) { M[p] = w.plan + p * nn; T[p] = w.planT + p * nn; }
  // the four differences: {unit-weight plan, its features, the two half-weight plans and their features}
  struct Diff {
    const float *P0, *F0, *P1, *F1, *P2, *F2;
  };
  const Diff diff[4] = {
      {M[0], fa2, M[2], fb1, M[3], fb2},   // g(a1)   (matching.py:64,66,67,80)
      {T[0], fa1, M[4], fb1, M[5], fb2},   // g(a2)   (:70,68,69,80)
      {T[1], fb2, T[2], fa1, T[4], fa2},   // g(b1)   (:65,72,74,82)
      {M[1], fb1, T[3], fa1, T[5], fa2},   // g(b2)   (:71,73,75,82)
  };
  // output blocks of this call: which difference, where it goes, which plan rows
  int which[4], nblk = 0;
  float* outp[4];
  const int r0 = full ? 0 : row_begin - half * N, cnt = full ? N : row_count;
  if (full) {
    which[nblk] = 0; outp[nblk++] = grad_a;
    which[nblk] = 1; outp[nblk++] = grad_a + (long)N * ldo;
    if (grad_b) {
      which[nblk] = 2; outp[nblk++] = grad_b;
      which[nblk] = 3; outp[nblk++] = grad_b + (long)N * ldo;
    }
  } else {
    which[nblk] = half; outp[nblk++] = grad_a;
    if (grad_b) { which[nblk] = 2 + half; outp[nblk++] = grad_b; }
  }
  if (x3) {
    // plan operand: per difference three n-row blocks in the order of its features in the stack
    //   g(a2): stack rows [0, 3N)  = a1 b1 b2 -> M0   -T4/2 -T5/2      (A = the TRANSPOSE of the plan applied: out = A^T F)
    //   g(a1): stack rows [N, 4N)  = b1 b2 a2 -> -T2/2 -T3/2  T0
    //   g(b1): stack rows [2N, 5N) = b2 a2 a1 -> M1   -M4/2 -M2/2
    //   g(b2): stack rows [3N, 6N) = a2 a1 b1 -> -M5/2 -M3/2  T1
    SplitSrc sp;
    memset(&sp, 0, sizeof(sp));
    const float* src[4][3] = {{T[2], T[3], T[0]}, {M[0], T[4], T[5]}, {M[1], M[4], M[2]}, {M[5], M[3], T[1]}};
    const float scl[4][3] = {{-0.5f, -0.5f, 1.f}, {1.f, -0.5f, -0.5f}, {1.f, -0.5f, -0.5f}, {-0.5f, -0.5f, 1.f}};
    const long frow[4] = {N, 0, 2L * N, 3L * N};
    sp.n = 3 * nblk;
    for (int z = 0; z < nblk; ++z)
      for (int t = 0; t < 3; ++t) {
        const int i = 3 * z + t;
        sp.src[i] = src[which[z]][t]; sp.ld[i] = N; sp.row0[i] = (long)i * N; sp.scale[i] = scl[which[z]][t];
      }
    x3_split(sp, N, N, w.PT, w.planeP, x3_plan_exp(sp), s);
    X3ApplyBlock xb[4];
    for (int z = 0; z < nblk; ++z) xb[z] = X3ApplyBlock{w.PT, 3L * z * N, frow[which[z]], 3 * N, outp[z]};
    rc = launch_apply_x3(xb, nblk, w.PT, x3_hdr(w.PT), w.planeP, N, w.FP, w.planeF, r0, cnt, D, ldo, s);
  } else {
    ApplyBlock blk[4];
    memset(blk, 0, sizeof(blk));
    const long ro = (long)r0 * N;
    for (int z = 0; z < nblk; ++z) {
      const Diff& d = diff[which[z]];
      blk[z].out = outp[z]; blk[z].rows = cnt; blk[z].nterms = 3; blk[z].alpha = 1.f;
      blk[z].t[0] = ApplyTerm{d.P1 + ro, d.F1, (long)N, N};
      blk[z].t[1] = ApplyTerm{d.P2 + ro, d.F2, (long)N, N};
      blk[z].t[2] = ApplyTerm{d.P0 + ro, d.F0, (long)N, N};
      blk[z].rescale[2] = -0.5f;
    }
    rc = launch_apply(blk, nblk, cnt, D, ldf, ldo, s);
  }
  if (rc) return rc;
  hipLaunchKernelGGL(entropy_finalize_kernel, dim3(1), dim3(1), 0, s, w.stats, 6, N, entropy);
  hipLaunchKernelGGL(closed_form_distance_kernel, dim3(1), dim3(1), 0, s, w.stats, N, dist);
  OTGAN_CHECK_LAUNCH("matching finalize");
  if (stats) hipMemcpyAsync(stats, w.stats, sizeof(double) * 24, hipMemcpyDeviceToDevice, s);
  return OTGAN_OK;
}

// ---- one split of the features per step for a data-parallel rank (round 5) ------------------------------------------
// A rank of the global matching scope multiplies the gathered features twice per step -- its three cost row slices
// (matching.py:29-39) and the plans applied to its own rows (:64-83) -- and until round 5 each library call split the
// blocks it read into the engine's two-fp16-plane operand itself: 3 328 + 4 096 rows of D floats at N = 1024, 0.76 ms of a
// 1.44 ms rank call.  The STACK is that operand as an object of the caller: six N-row blocks [a1 b1 b2 a2 a1 b1] (the
// layout of otgan_matching_two_batch_grad_f32: every difference contracts over three adjacent blocks), of which a rank
// fills only the row ranges its calls read -- a first-half rank in a generator step rows [N, 4N) = b1 b2 a2 (the Y blocks
// of its cost slices AND the contraction blocks of g(a1)) plus its own rows of a1 -- once, and hands it to both calls.
size_t otgan_matching_stack_bytes(int N, int D) {
  if (N <= 0 || D <= 0 || !x3_shape_ok(N, N, D)) return 0;
  return kX3HdrBytes + sizeof(u16) * X3_NP * x3_plane_elems(6 * (size_t)N, D);
}
int otgan_matching_stack_split_f32(const float* fa, const float* fb, int N, int D, long ldf, int nranges,
                                   const int* range_begin, const int* range_rows, void* stack, void* stream) {
  OTGAN_CHECK_ARG(fa && fb && stack && range_begin && range_rows, "null pointer");
  OTGAN_CHECK_ARG(x3_shape_ok(N, N, D) && ldf >= D && ldf % 4 == 0 && aligned16(fa) && aligned16(fb),
                  "shape N=%d D=%d not taken by the split-precision engine (otgan_matching_stack_bytes == 0) or unaligned", N, D);
  OTGAN_CHECK_ARG(nranges > 0 && nranges <= 6, "1 .. 6 row ranges");
  hipStream_t s = (hipStream_t)stream;
  const float *fa1 = fa, *fa2 = fa + (long)N * ldf, *fb1 = fb, *fb2 = fb + (long)N * ldf;
  const float* blocks[6] = {fa1, fb1, fb2, fa2, fa1, fb1};
  u16* FP = (u16*)((char*)stack + kX3HdrBytes);
  const long plane = (long)x3_plane_elems(6 * (size_t)N, D);
  // one job per (range, block it touches): a job's sources share a row count
  X3SplitJob jobs[12];
  int nj = 0;
  for (int i = 0; i < nranges; ++i) {
    int b = range_begin[i], e = range_begin[i] + range_rows[i];
    OTGAN_CHECK_ARG(b >= 0 && e <= 6 * N && b < e && b % 32 == 0 && e % 32 == 0, "range %d: [%d, %d) outside the stack or not on 32-row blocks", i, b, e);
    while (b < e) {
      const int blk = b / N, stop = (blk + 1) * N < e ? (blk + 1) * N : e;
      // pieces with the same row count share a job = ONE launch per pass (blockIdx.y = piece): a rank's three or four whole
      // blocks were four launches of 45 - 50 us plus four empty second-pass launches (round 5, late: 205 -> ~185 us per step)
      int j = 0;
      while (j < nj && !(jobs[j].rows == stop - b && jobs[j].ss.n < 12)) ++j;
      if (j == nj) {
        OTGAN_CHECK_ARG(nj < 12, "too many block pieces");
        memset(&jobs[nj], 0, sizeof(jobs[nj]));
        jobs[nj].rows = stop - b; jobs[nj].K = D; jobs[nj].dst = FP; jobs[nj].plane_stride = plane;
        ++nj;
      }
      X3SplitJob& q = jobs[j];
      const int k = q.ss.n++;
      q.ss.src[k] = blocks[blk] + (long)(b - blk * N) * ldf; q.ss.ld[k] = ldf; q.ss.row0[k] = b; q.ss.scale[k] = 1.f;
      b = stop;
    }
  }
  x3_split_group(jobs, nj, x3_hdr(FP), kX3FeatureExp, s);
  OTGAN_CHECK_LAUNCH("feature stack split");
  return OTGAN_OK;
}
size_t otgan_cost_slices_stack_workspace_bytes(int P, int nrows, int N, int D) {
  if (P <= 0 || P > kMaxProb || nrows <= 0 || N <= 0 || D <= 0) return 0;
  return align_up(sizeof(float) * (size_t)P * nrows * N * x3_plan_cost(P, nrows, N, D).nsplit, 256);
}
int otgan_cost_slices_stack_f32(const void* stack, int N, int D, int P, const long* xrow, const long* yrow, int nrows,
                                float lambda, float* K, void* workspace, size_t workspace_bytes, void* stream) {
  OTGAN_CHECK_ARG(stack && xrow && yrow && K, "null pointer");
  OTGAN_CHECK_ARG(P > 0 && P <= kMaxProb && x3_shape_ok(nrows, N, D) && x3_shape_ok(N, N, D), "shape not taken by the split-precision engine");
  for (int p = 0; p < P; ++p)
    OTGAN_CHECK_ARG(xrow[p] >= 0 && xrow[p] % 32 == 0 && xrow[p] + nrows <= 6L * N && yrow[p] >= 0 && yrow[p] % N == 0 && yrow[p] < 6L * N,
                    "problem %d: rows outside the stack", p);
  const size_t need = otgan_cost_slices_stack_workspace_bytes(P, nrows, N, D);
  if (!workspace || workspace_bytes < need) {
    otgan_set_error("workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
    return OTGAN_ERR_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  const u16* FP = (const u16*)((const char*)stack + kX3HdrBytes);
  const long plane = (long)x3_plane_elems(6 * (size_t)N, D);
  ProfScope ps(OTGAN_PROF_COST_GEMM, 2.0 * P * nrows * (double)N * D, 4.0 * P * ((double)nrows + N) * D, s);
  return launch_cost_x3(FP, plane, 6L * N, xrow, yrow, nullptr, P, nrows, N, D, lambda, (float*)workspace, K, s);
}
int otgan_matching_two_batch_rows_grad_stack_f32(const void* stack, int N, int D, float lambda, int iters, int row_begin,
                                                 int row_count, const float* K_pre, float* grad_a, float* grad_b, long ldo,
                                                 float* entropy, double* dist, double* stats, void* workspace,
                                                 size_t workspace_bytes, void* stream) {
  OTGAN_CHECK_ARG(stack, "null stack");
  OTGAN_CHECK_ARG(!(row_begin == 0 && row_count == 2 * N), "the row-range variant takes a range inside one mini-batch");
  // (the feature pointers only serve the alignment test of the shared implementation)
  const float* aligned = reinterpret_cast<const float*>(stack);
  return matching_grad_impl(aligned, aligned, N, D, D, lambda, iters, row_begin, row_count, K_pre, grad_a, grad_b, ldo, entropy,
                            dist, stats, workspace, workspace_bytes, stream, stack);
}

size_t otgan_matching_grad_workspace_bytes(int N, int D) {
  if (N <= 0 || D <= 0) return 0;
  return carve_match(nullptr, 0, 6, N, D, 2 * N, true).bytes;
}
int otgan_matching_two_batch_grad_f32(const float* fa, const float* fb, int N, int D, long ldf, float lambda, int iters,
                                      float* grad_a, float* grad_b, long ldo, float* entropy, double* dist,
                                      double* stats, void* workspace, size_t workspace_bytes, void* stream) {
  return matching_grad_impl(fa, fb, N, D, ldf, lambda, iters, 0, 2 * N, nullptr, grad_a, grad_b, ldo, entropy, dist, stats,
                            workspace, workspace_bytes, stream);
}
int otgan_matching_two_batch_rows_grad_f32(const float* fa, const float* fb, int N, int D, long ldf, float lambda,
                                           int iters, int row_begin, int row_count, const float* K_pre, float* grad_a,
                                           float* grad_b, long ldo, float* entropy, double* dist, double* stats,
                                           void* workspace, size_t workspace_bytes, void* stream) {
  OTGAN_CHECK_ARG(!(row_begin == 0 && row_count == 2 * N), "the row-range variant takes a range inside one mini-batch");
  return matching_grad_impl(fa, fb, N, D, ldf, lambda, iters, row_begin, row_count, K_pre, grad_a, grad_b, ldo, entropy,
                            dist, stats, workspace, workspace_bytes, stream);
}

int otgan_matching_single_batch_f32(const float* fa, const float* fb, int n, int D, long ldf,
                                    float lambda, int iters, float* f_aa, float* f_bb,
                                    float* f_ab, float* f_ba, long ldo, float* entropy,
                                    double* dist, double* stats, void* workspace,
                                    size_t workspace_bytes, void* stream) {
  OTGAN_CHECK_ARG(fa && fb && f_aa && f_bb && f_ab && f_ba && entropy && dist, "null pointer");
  OTGAN_CHECK_ARG(n > 0 && D > 0 && ldf >= D && ldo >= D && iters >= 0, "bad sizes n=%d D=%d", n, D);
  OTGAN_CHECK_ARG(ldf == D && ldo == D, "feature arrays must be contiguous (ld == D)");
  hipStream_t s = (hipStream_t)stream;
  MatchWs w = carve_match(workspace, workspace_bytes, 3, n, D, n);
  if (!workspace || workspace_bytes < w.bytes) {
    otgan_set_error("workspace too small: need %zu bytes, got %zu", w.bytes, workspace_bytes);
    return OTGAN_ERR_WORKSPACE;
  }
  const float* X[3] = {fa, fb, fa};
  const float* Y[3] = {fa, fb, fb};
  const float diag[3] = {999.f, 999.f, 0.f};  // matching.py:109-110
  const bool x3 = w.x3 && ldf % 4 == 0 && ldo % 4 == 0 && aligned16(fa) && aligned16(fb);
  int rc;
  if (x3) {   // stacked rows: a [0,n), b [n,2n); problems aa, bb, ab
    const long xrow[3] = {0, n, 0};
    const long yrow[3] = {0, n, n};
    ProfScope ps(OTGAN_PROF_COST_GEMM, 6.0 * n * (double)n * D, 8.0 * n * (double)D, s);
    x3_split_features(w, fa, fb, n, D, ldf, s);
    rc = launch_cost_x3(w.FP, w.planeF, 2L * n, xrow, yrow, diag, 3, n, n, D, lambda, w.partial, w.K, s);
  } else {
    rc = launch_cost(X, Y, nullptr, nullptr, diag, 3, n, n, D, ldf, lambda, OTGAN_COST_COSINE,
                     w.partial, w.K, s);
  }
  if (rc) return rc;
  rc = launch_sinkhorn(w.K, 3, n, n, iters, lambda, w.plan, w.planT, w.stats, w.fg, s);
  if (rc) return rc;
  const size_t nn = (size_t)n * n;
  ApplyBlock blk[4];
  memset(blk, 0, sizeof(blk));
  auto set1 = [&](int i, float* out, const float* P0, const float* F0) {
    blk[i].out = out; blk[i].rows = n; blk[i].nterms = 1; blk[i].alpha = 1.f;
    blk[i].t[0] = ApplyTerm{P0, F0, (long)n, n};
  };
  set1(0, f_aa, w.plan, fa);            // :131
  set1(1, f_bb, w.plan + nn, fb);       // :132
  set1(2, f_ab, w.plan + 2 * nn, fb);   // :133
  set1(3, f_ba, w.planT + 2 * nn, fa);  // :134
  if (x3) {
    const int orderM[3] = {0, 1, 2};
    const float alpha[3] = {1.f, 1.f, 1.f};
    x3_split_plans(w, w.plan, w.planT, 3, n, orderM, alpha, s);
    const long n1 = n;
    const X3ApplyBlock xb[4] = {
        {w.PT, 0 * n1, 0 * n1, n, f_aa},     // M_aa . a
        {w.PT, 1 * n1, 1 * n1, n, f_bb},     // M_bb . b
        {w.PT, 2 * n1, 1 * n1, n, f_ab},     // M_ab . b
        {w.PM, 2 * n1, 0 * n1, n, f_ba},     // M_ab^T . a
    };
    rc = launch_apply_x3(xb, 4, w.PT < w.PM ? w.PT : w.PM, x3_hdr(w.PT), w.planeP, n, w.FP, w.planeF, 0, n, D, ldo, s);
  } else {
    rc = launch_apply(blk, 4, n, D, ldf, ldo, s);
  }
  if (rc) return rc;
  hipLaunchKernelGGL(entropy_finalize_kernel, dim3(1), dim3(1), 0, s, w.stats, 3, n, entropy);
  rc = launch_distance(fa, fb, f_aa, f_bb, f_ab, (long)n * D, 2.0 * n, dist, w.dot3, s, w.stats, 3);
  if (rc) return rc;
  if (stats) hipMemcpyAsync(stats, w.stats, sizeof(double) * 12, hipMemcpyDeviceToDevice, s);
  return OTGAN_OK;
}

// Training-mode single-batch matching: the injected gradients f_aa - f_ab (train.py:111) and f_bb - f_ba (train.py:125-126)
// of --single_batch directly, for all rows or for the rows of one data-parallel rank, with the three log-kernels
// optionally precomputed (the rank's row slices of matching.py:99-104, all-gathered: K_pre [3][n][n] in the order a-a
// (+999 I), b-b (+999 I), a-b).  g(a) = M_aa a - M_ab b, g(b) = M_bb b - M_ab^T a: two-term plan applications.
static int single_grad_impl(const float* fa, const float* fb, int n, int D, long ldf, float lambda, int iters, int row_begin,
                            int row_count, const float* K_pre, float* grad_a, float* grad_b, long ldo, float* entropy,
                            double* dist, double* stats, void* workspace, size_t workspace_bytes, void* stream) {
  OTGAN_CHECK_ARG(fa && fb && grad_a && entropy && dist, "null pointer");
  OTGAN_CHECK_ARG(n > 0 && D > 0 && ldf >= D && ldo >= D && iters >= 0, "bad sizes n=%d D=%d", n, D);
  OTGAN_CHECK_ARG(row_begin >= 0 && row_count > 0 && row_begin + row_count <= n,
                  "row range [%d, %d) outside [0, %d)", row_begin, row_begin + row_count, n);
  hipStream_t s = (hipStream_t)stream;
  MatchWs w = carve_match(workspace, workspace_bytes, 3, n, D, n, true);
  if (!workspace || workspace_bytes < w.bytes) {
    otgan_set_error("workspace too small: need %zu bytes, got %zu", w.bytes, workspace_bytes);
    return OTGAN_ERR_WORKSPACE;
  }
  const float* X[3] = {fa, fb, fa};
  const float* Y[3] = {fa, fb, fb};
  const float diag[3] = {999.f, 999.f, 0.f};  // matching.py:109-110
  const bool x3 = w.x3 && ldf % 4 == 0 && ldo % 4 == 0 && aligned16(fa) && aligned16(fb) && aligned16(grad_a) &&
                  aligned16(grad_b) && row_begin % 16 == 0;
  int rc = OTGAN_OK;
  if (x3) x3_split_features(w, fa, fb, n, D, ldf, s);   // stacked rows: a [0, n), b [n, 2n)
  const float* Kuse = K_pre;
  if (!K_pre) {
    if (x3) {
      const long xrow[3] = {0, n, 0};
      const long yrow[3] = {0, n, n};
      ProfScope ps(OTGAN_PROF_COST_GEMM, 6.0 * n * (double)n * D, 8.0 * n * (double)D, s);
      rc = launch_cost_x3(w.FP, w.planeF, 2L * n, xrow, yrow, diag, 3, n, n, D, lambda, w.partial, w.K, s);
    } else {
      rc = launch_cost(X, Y, nullptr, nullptr, diag, 3, n, n, D, ldf, lambda, OTGAN_COST_COSINE, w.partial, w.K, s);
    }
    if (rc) return rc;
    Kuse = w.K;
  }
  rc = launch_sinkhorn(Kuse, 3, n, n, iters, lambda, w.plan, w.planT, w.stats, w.fg, s);
  if (rc) return rc;
  const size_t nn = (size_t)n * n;
  const float *M[3], *T[3];
  for (int p = 0; p < 3; ++p) { M[p] = w.plan + p * nn; T[p] = w.planT + p * nn; }
  const int nblk = grad_b ? 2 : 1;
  float* outp[2] = {grad_a, grad_b};
  if (x3) {
    // plan operand (A = the TRANSPOSE of the plan applied: out = A^T F), per difference two n-row blocks in the order
    // of the feature stack [a; b]:   g(a): T_aa, -T_ab        g(b): -M_ab, T_bb
    SplitSrc sp;
    memset(&sp, 0, sizeof(sp));
    const float* src[2][2] = {{T[0], T[2]}, {M[2], T[1]}};
    const float scl[2][2] = {{1.f, -1.f}, {-1.f, 1.f}};
    sp.n = 2 * nblk;
    for (int z = 0; z < nblk; ++z)
      for (int t = 0; t < 2; ++t) {
        const int i = 2 * z + t;
        sp.src[i] = src[z][t]; sp.ld[i] = n; sp.row0[i] = (long)i * n; sp.scale[i] = scl[z][t];
      }
    x3_split(sp, n, n, w.PT, w.planeP, x3_plan_exp(sp), s);
    X3ApplyBlock xb[2];
    for (int z = 0; z < nblk; ++z) xb[z] = X3ApplyBlock{w.PT, 2L * z * n, 0, 2 * n, outp[z]};
    rc = launch_apply_x3(xb, nblk, w.PT, x3_hdr(w.PT), w.planeP, n, w.FP, w.planeF, row_begin, row_count, D, ldo, s);
  } else {
    ApplyBlock blk[2];
    memset(blk, 0, sizeof(blk));
    const long ro = (long)row_begin * n;
    const float *P1[2] = {M[2], T[2]}, *F1[2] = {fb, fa}, *P0[2] = {M[0], M[1]}, *F0[2] = {fa, fb};
    for (int z = 0; z < nblk; ++z) {
      blk[z].out = outp[z]; blk[z].rows = row_count; blk[z].nterms = 2; blk[z].alpha = 1.f;
      blk[z].t[0] = ApplyTerm{P1[z] + ro, F1[z], (long)n, n};
      blk[z].t[1] = ApplyTerm{P0[z] + ro, F0[z], (long)n, n};
      blk[z].rescale[1] = -1.f;     // (-1)(M_ab b) + M_aa a
    }
    rc = launch_apply(blk, nblk, row_count, D, ldf, ldo, s);
  }
  if (rc) return rc;
  hipLaunchKernelGGL(entropy_finalize_kernel, dim3(1), dim3(1), 0, s, w.stats, 3, n, entropy);
  hipLaunchKernelGGL(closed_form_single_distance_kernel, dim3(1), dim3(256), 0, s, w.stats, w.plan, n, (double)diag[0], dist);
  OTGAN_CHECK_LAUNCH("single-batch matching finalize");
  if (stats) hipMemcpyAsync(stats, w.stats, sizeof(double) * 12, hipMemcpyDeviceToDevice, s);
  return OTGAN_OK;
}

size_t otgan_matching_single_batch_grad_workspace_bytes(int n, int D) {
  if (n <= 0 || D <= 0) return 0;
  return carve_match(nullptr, 0, 3, n, D, n, true).bytes;
}
int otgan_matching_single_batch_grad_f32(const float* fa, const float* fb, int n, int D, long ldf, float lambda, int iters,
                                         float* grad_a, float* grad_b, long ldo, float* entropy, double* dist,
                                         double* stats, void* workspace, size_t workspace_bytes, void* stream) {
  return single_grad_impl(fa, fb, n, D, ldf, lambda, iters, 0, n, nullptr, grad_a, grad_b, ldo, entropy, dist, stats, workspace,
                          workspace_bytes, stream);
}
int otgan_matching_single_batch_rows_grad_f32(const float* fa, const float* fb, int n, int D, long ldf, float lambda,
                                              int iters, int row_begin, int row_count, const float* K_pre, float* grad_a,
                                              float* grad_b, long ldo, float* entropy, double* dist, double* stats,
                                              void* workspace, size_t workspace_bytes, void* stream) {
  return single_grad_impl(fa, fb, n, D, ldf, lambda, iters, row_begin, row_count, K_pre, grad_a, grad_b, ldo, entropy, dist,
                          stats, workspace, workspace_bytes, stream);
}

// workspace of the (batched) staged cost entry point: split-K partial sums, the toy cost's row statistics and --
// on the split-precision path -- the stacked operand of the distinct X / Y blocks (at most 2 P of them)
static size_t cost_batched_ws(int P, int n, int m, int D, bool* x3_out) {
  const bool x3 = x3_shape_ok(n, m, D);
  if (x3_out) *x3_out = x3;
  int nsplit = plan_cost(P, n, m, D).nsplit;
  if (x3 && x3_plan_cost(P, n, m, D).nsplit > nsplit) nsplit = x3_plan_cost(P, n, m, D).nsplit;
  size_t b = align_up(sizeof(float) * (size_t)P * n * m * nsplit, 256) + (size_t)P * (align_up(sizeof(float) * n, 256) +
                                                                                      align_up(sizeof(float) * m, 256));
  if (x3) b += align_up(kX3HdrBytes + sizeof(u16) * X3_NP * x3_plane_elems((size_t)P * ((size_t)n + m), D), 256);
  return b;
}

size_t otgan_cost_matrix_workspace_bytes(int n, int m, int D) {
  if (n <= 0 || m <= 0 || D <= 0) return 0;
  return cost_batched_ws(1, n, m, D, nullptr);
}

size_t otgan_cost_matrix_batched_workspace_bytes(int P, int n, int m, int D) {
  if (P <= 0 || P > kMaxProb || n <= 0 || m <= 0 || D <= 0) return 0;
  return cost_batched_ws(P, n, m, D, nullptr);
}

int otgan_cost_matrix_batched_f32(const float* const* X, const float* const* Y, int P, int n, int m, int D, long ldf,
                                  float lambda, int cost_kind, const float* diag_add, float* K, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  OTGAN_CHECK_ARG(X && Y && K, "null pointer");
  OTGAN_CHECK_ARG(P > 0 && P <= kMaxProb, "1 <= P <= %d problems per call", kMaxProb);
  OTGAN_CHECK_ARG(n > 0 && m > 0 && D > 0 && ldf >= D, "bad sizes");
  OTGAN_CHECK_ARG(cost_kind == OTGAN_COST_COSINE || cost_kind == OTGAN_COST_SQEUCLID_MEAN, "unknown cost kind %d", cost_kind);
  for (int p = 0; p < P; ++p) OTGAN_CHECK_ARG(X[p] && Y[p], "null block pointer");
  bool x3 = false;
  const size_t need = cost_batched_ws(P, n, m, D, &x3);
  if (!workspace || workspace_bytes < need) {
    otgan_set_error("workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
    return OTGAN_ERR_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  int nsplit = plan_cost(P, n, m, D).nsplit;
  if (x3 && x3_plan_cost(P, n, m, D).nsplit > nsplit) nsplit = x3_plan_cost(P, n, m, D).nsplit;
  Carver c(workspace, workspace_bytes);
  float* partial = (float*)c.take(sizeof(float) * (size_t)P * n * m * nsplit);
  const float *xp[kMaxProb], *yp[kMaxProb];
  for (int p = 0; p < P; ++p) {
    xp[p] = (float*)c.take(sizeof(float) * n);
    yp[p] = (float*)c.take(sizeof(float) * m);
  }
  x3 = x3 && cost_kind == OTGAN_COST_COSINE && ldf % 4 == 0;
  for (int p = 0; p < P && x3; ++p) x3 = aligned16(X[p]) && aligned16(Y[p]);
  if (x3) {
    // every distinct block is split once into a stacked operand (a rank's slices share their X, two of them a Y)
    u16* FP = (u16*)((char*)c.take(kX3HdrBytes + sizeof(u16) * X3_NP * x3_plane_elems((size_t)P * ((size_t)n + m), D)) + kX3HdrBytes);
    const long plane = (long)x3_plane_elems((size_t)P * ((size_t)n + m), D);
    const float* uniq[2 * kMaxProb];
    long urow[2 * kMaxProb];
    int ulen[2 * kMaxProb];
    int nu = 0;
    long rows = 0, xrow[kMaxProb], yrow[kMaxProb];
    SplitSrc sx, sy;
    memset(&sx, 0, sizeof(sx));
    memset(&sy, 0, sizeof(sy));
    auto place = [&](const float* ptr, int r, SplitSrc& ss) -> long {
      for (int i = 0; i < nu; ++i)
        if (uniq[i] == ptr && ulen[i] >= r) return urow[i];   // same block, at least as many rows already split
      uniq[nu] = ptr; urow[nu] = rows; ulen[nu] = r;
      ss.src[ss.n] = ptr; ss.ld[ss.n] = ldf; ss.row0[ss.n] = rows; ss.scale[ss.n] = 1.f;
      ++ss.n; ++nu;
      rows += r;          // n and m are multiples of 32: every block starts on a row-block boundary
      return rows - r;
    };
    for (int p = 0; p < P; ++p) xrow[p] = place(X[p], n, sx);
    for (int p = 0; p < P; ++p) yrow[p] = place(Y[p], m, sy);   // (a pointer used as X with fewer rows is placed again)
    ProfScope ps(OTGAN_PROF_COST_GEMM, 2.0 * P * n * (double)m * D, 4.0 * P * ((double)n + m) * D, s);
    X3SplitJob jobs[2] = {{sx, n, D, FP, plane}, {sy, m, D, FP, plane}};
    x3_split_group(jobs, 2, x3_hdr(FP), kX3FeatureExp, s);
    return launch_cost_x3(FP, plane, rows, xrow, yrow, diag_add, P, n, m, D, lambda, partial, K, s);
  }
  if (cost_kind == OTGAN_COST_SQEUCLID_MEAN) {
    for (int p = 0; p < P; ++p) {
      hipLaunchKernelGGL(row_halfmeansq_kernel, dim3(ceil_div(n, 4)), dim3(256), 0, s, X[p], ldf, n, D, (float*)xp[p]);
      hipLaunchKernelGGL(row_halfmeansq_kernel, dim3(ceil_div(m, 4)), dim3(256), 0, s, Y[p], ldf, m, D, (float*)yp[p]);
    }
  }
  return launch_cost(X, Y, cost_kind == OTGAN_COST_SQEUCLID_MEAN ? xp : nullptr,
                     cost_kind == OTGAN_COST_SQEUCLID_MEAN ? yp : nullptr, diag_add, P, n, m, D, ldf,
                     lambda, cost_kind, partial, K, s);
}

int otgan_cost_matrix_f32(const float* X, const float* Y, int n, int m, int D, long ldf,
                          float lambda, int cost_kind, float diag_add, float* K, void* workspace,
                          size_t workspace_bytes, void* stream) {
  OTGAN_CHECK_ARG(X && Y && K, "null pointer");
  const float* Xp[1] = {X};
  const float* Yp[1] = {Y};
  const float dg[1] = {diag_add};
  return otgan_cost_matrix_batched_f32(Xp, Yp, 1, n, m, D, ldf, lambda, cost_kind, dg, K, workspace, workspace_bytes, stream);
}

size_t otgan_sinkhorn_workspace_bytes(int P, int n, int m) {
  if (P <= 0 || n <= 0 || m <= 0) return 0;
  return align_up(sizeof(unsigned long long) * (size_t)P * (n + m) + 64, 256);  // tagged potentials + fail word
}

int otgan_sinkhorn_plan_f32(const float* K, int P, int n, int m, int iters, float lambda,
                            float* plan, float* planT, double* stats, void* workspace,
                            size_t workspace_bytes, void* stream) {
  OTGAN_CHECK_ARG(K && plan && planT && stats, "null pointer");
  OTGAN_CHECK_ARG(P > 0 && n > 0 && m > 0 && iters >= 0 && lambda != 0.f, "bad sizes");
  const size_t need = otgan_sinkhorn_workspace_bytes(P, n, m);
  if (!workspace || workspace_bytes < need) {
    otgan_set_error("workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
    return OTGAN_ERR_WORKSPACE;
  }
  return launch_sinkhorn(K, P, n, m, iters, lambda, plan, planT, stats, (float*)workspace,
                         (hipStream_t)stream);
}

int otgan_plan_apply_f32(const float* plan, long ldp, int rows, int kdim, const float* feat,
                         long ldf, int D, float alpha, float* out, long ldo, void* stream) {
  OTGAN_CHECK_ARG(plan && feat && out, "null pointer");
  OTGAN_CHECK_ARG(rows > 0 && kdim > 0 && D > 0 && ldp >= kdim && ldf >= D && ldo >= D, "bad sizes");
  ApplyBlock blk;
  memset(&blk, 0, sizeof(blk));
  blk.out = out; blk.rows = rows; blk.nterms = 1; blk.alpha = alpha;
  blk.t[0] = ApplyTerm{plan, feat, ldp, kdim};
  return launch_apply(&blk, 1, rows, D, ldf, ldo, (hipStream_t)stream, /*bounded_plans=*/false);   // (any matrix: the exact-fp32 kernel)
}

int otgan_calc_distance_f32(const float* a, const float* b, const float* aa, const float* bb,
                            const float* ab, long rows, int D, double denom, double* dist,
                            double* scratch3, void* stream) {
  OTGAN_CHECK_ARG(a && b && aa && bb && ab && dist && scratch3, "null pointer");
  OTGAN_CHECK_ARG(rows > 0 && D > 0 && denom != 0.0, "bad sizes");
  return launch_distance(a, b, aa, bb, ab, rows * (long)D, denom, dist, scratch3,
                         (hipStream_t)stream);
}


// Sweep statistics of the Sinkhorn kernels (DESIGN section 3 "Sinkhorn sweeps without exponentials"; tools/soak_sinkhorn.py):
// on = 1 allocates (once) and zeroes eight device counters that every problem solved from now on adds to -- [0] problems,
// [1] log-domain sweeps, [2] linear sweeps, [3] entries into the linear form, [4] fold-backs, [5] sum of first-entry sweeps,
// [6] problems that never entered; on = 0 detaches them.  otgan_sinkhorn_counters_read synchronises the device.
int otgan_sinkhorn_counters(int on) {
  static unsigned long long* buf = nullptr;
  if (on && !buf) {
    void* p = nullptr;
    if (hipMalloc(&p, 8 * sizeof(unsigned long long)) != hipSuccess) {
      otgan_set_error("otgan_sinkhorn_counters: allocation failed");
      return OTGAN_ERR_LAUNCH;
    }
    buf = static_cast<unsigned long long*>(p);
  }
  if (on) (void)hipMemset(buf, 0, 8 * sizeof(unsigned long long));
  g_sweep_counters = on ? buf : nullptr;
  return OTGAN_OK;
}
int otgan_sinkhorn_counters_read(long long* out8, int reset) {
  OTGAN_CHECK_ARG(out8, "null pointer");
  if (!g_sweep_counters) {
    for (int i = 0; i < 8; ++i) out8[i] = 0;
    return OTGAN_OK;
  }
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(out8, g_sweep_counters, 8 * sizeof(long long), hipMemcpyDeviceToHost) != hipSuccess) {
    otgan_set_error("otgan_sinkhorn_counters_read: copy failed");
    return OTGAN_ERR_LAUNCH;
  }
  if (reset) (void)hipMemset(g_sweep_counters, 0, 8 * sizeof(unsigned long long));
  return OTGAN_OK;
}

}  // extern "C"
