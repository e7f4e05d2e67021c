// Prototype 3: fp32-accurate NT GEMM on the bf16 MFMA pipe with PRE-SPLIT operands (three bf16 planes
// per fp32 tensor: x = hi + mid + lo), six v_mfma_f32_32x32x16_bf16 per 16-wide k slab
// (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid).  Batched like the Winograd-domain GEMMs.
//   C[f][M][N] = A[f][M][K] . B[f][N][K]^T,   planes Ap[3][F][M][K], Bp[3][F][N][K]  (bf16, k contiguous)
// LDS: per (operand, piece) tile [128 rows][BK*2 + 16 bytes]: row stride = odd number of 16-byte
// slots -> conflict-free ds_read_b128 fragment reads (lane = row, 8 consecutive k per lane).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/ablate/gemm_bf16x3_v2.hip -o /tmp/gemm_bf16x3_v2
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned bf16_rne(float x) {
  unsigned u = __float_as_uint(x);
  u += 0x7fffu + ((u >> 16) & 1u);
  return u >> 16;
}
__global__ void split_kernel(const float* __restrict__ x, long n, u16* __restrict__ p1, u16* __restrict__ p2,
                             u16* __restrict__ p3) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = x[i];
    const unsigned a1 = bf16_rne(v);
    const float r1 = v - __uint_as_float(a1 << 16);
    const unsigned a2 = bf16_rne(r1);
    const float r2 = r1 - __uint_as_float(a2 << 16);
    p1[i] = (u16)a1;
    p2[i] = (u16)a2;
    p3[i] = (u16)bf16_rne(r2);
  }
}

struct Args {
  const u16* Ap;   // [3][F][M][K]
  const u16* Bp;   // [3][F][N][K]
  float* C;        // [F][M][N]
  int M, N, K, F;
};

template <int BK, bool DB, int NTERMS>
__global__ __launch_bounds__(256) void gemm_kernel(Args a) {
  constexpr int BM = 128, BN = 128, RS = BK * 2 + 16, CPR = BK / 8;
  constexpr int TILE = BM * RS;                       // bytes of one (operand, piece) tile
  constexpr int CHUNKS = 6 * BM * CPR, PER = CHUNKS / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, g = lane >> 5;
  const int f = blockIdx.z;
  const long planeA = (long)a.F * a.M * a.K, planeB = (long)a.F * a.N * a.K;
  const u16* Ab = a.Ap + ((long)f * a.M + (long)blockIdx.x * BM) * a.K;
  const u16* Bb = a.Bp + ((long)f * a.N + (long)blockIdx.y * BN) * a.K;
  u32x4 rg[PER];
  auto gload = [&](int kt) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int id = i * 256 + tid;
      const int c = id % CPR, row = (id / CPR) % BM, tile = id / (BM * CPR);   // tile = operand*3 + piece
      const u16* src = (tile < 3 ? Ab + tile * planeA : Bb + (tile - 3) * planeB) + (long)row * a.K + kt * BK + c * 8;
      rg[i] = *reinterpret_cast<const u32x4*>(src);
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int id = i * 256 + tid;
      const int c = id % CPR, row = (id / CPR) % BM, tile = id / (BM * CPR);
      *reinterpret_cast<u32x4*>(smem + buf * 6 * TILE + tile * TILE + row * RS + c * 16) = rg[i];
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
  const int nk = a.K / BK;
  gload(0);
  sstore(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = DB ? (kt & 1) : 0;
    if (kt + 1 < nk) gload(kt + 1);
    const unsigned char* base = smem + cur * 6 * TILE;
    const unsigned char* pa = base + (wm * 64 + r) * RS + 16 * g;
    const unsigned char* pb = base + 3 * TILE + (wn * 64 + r) * RS + 16 * g;
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      bf16x8 A[2][3], B[2][3];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          A[t][p] = *reinterpret_cast<const bf16x8*>(pa + p * TILE + t * 32 * RS + 32 * s);
          B[t][p] = *reinterpret_cast<const bf16x8*>(pb + p * TILE + t * 32 * RS + 32 * s);
        }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (NTERMS >= 6) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][2], B[j][0], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][0], B[j][2], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][1], B[j][1], acc[i][j], 0, 0, 0);
          }
          if (NTERMS >= 3) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][1], B[j][0], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][0], B[j][1], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][0], B[j][0], acc[i][j], 0, 0, 0);
        }
    }
    if (DB) {
      if (kt + 1 < nk) sstore(cur ^ 1);
      __syncthreads();
    } else {
      __syncthreads();
      if (kt + 1 < nk) {
        sstore(0);
        __syncthreads();
      }
    }
  }
  float* C = a.C + (long)f * a.M * a.N;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int rr = (q & 3) + 8 * (q >> 2) + 4 * g;
        const long m = (long)blockIdx.x * BM + wm * 64 + i * 32 + rr;
        const long n = (long)blockIdx.y * BN + wn * 64 + j * 32 + r;
        C[m * a.N + n] = acc[i][j][q];
      }
}

template <int BK, bool DB, int NTERMS>
void run(const char* name, const Args& a, const std::vector<float>& hA, const std::vector<float>& hB) {
  constexpr int RS = BK * 2 + 16;
  const size_t lds = (size_t)(DB ? 2 : 1) * 6 * 128 * RS;
  auto kern = gemm_kernel<BK, DB, NTERMS>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  dim3 grid(a.M / 128, a.N / 128, a.F);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, grid, dim3(256), lds, 0, a);
  hipEventRecord(e0);
  const int reps = 5;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, grid, dim3(256), lds, 0, a);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  const double tf = 2.0 * a.F * (double)a.M * a.N * a.K / (ms * 1e-3) / 1e12;
  // relative L2 error of one 64x64 corner of batch 0 against fp64
  std::vector<float> c(64 * 64);
  for (int m = 0; m < 64; ++m) hipMemcpy(&c[m * 64], a.C + (long)m * a.N, 64 * 4, hipMemcpyDeviceToHost);
  double num = 0, den = 0;
  for (int m = 0; m < 64; ++m)
    for (int n = 0; n < 64; ++n) {
      double s = 0;
      for (int k = 0; k < a.K; ++k) s += (double)hA[(long)m * a.K + k] * (double)hB[(long)n * a.K + k];
      num += (s - c[m * 64 + n]) * (s - c[m * 64 + n]);
      den += s * s;
    }
  printf("%-26s grid %4dx%-3dx%-2d lds %6zu  %8.3f ms  %7.1f TF/s-equiv  relL2 %.2e\n", name, grid.x, grid.y, grid.z,
         lds, ms, tf, sqrt(num / den));
}


// 8-wave variant: block tile (WM*MT*32) x (WN*NT*32), single LDS buffer, BK = 32.
template <int WM, int WN, int MT, int NT>
__global__ __launch_bounds__(WM * WN * 64) void gemm_big_kernel(Args a) {
  constexpr int BK = 32, RS = BK * 2 + 16, CPR = BK / 8, THREADS = WM * WN * 64;
  constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
  constexpr int TA = BM * RS, TB = BN * RS;            // bytes of one piece tile of A / B
  constexpr int CH_A = 3 * BM * CPR, CH_B = 3 * BN * CPR;
  constexpr int PER_A = CH_A / THREADS, PER_B = CH_B / THREADS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sA = smem;            // [3][BM][RS]
  unsigned char* sB = smem + 3 * TA;   // [3][BN][RS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, g = lane >> 5;
  const int f = blockIdx.z;
  const long planeA = (long)a.F * a.M * a.K, planeB = (long)a.F * a.N * a.K;
  const u16* Ab = a.Ap + ((long)f * a.M + (long)blockIdx.x * BM) * a.K;
  const u16* Bb = a.Bp + ((long)f * a.N + (long)blockIdx.y * BN) * a.K;
  u32x4 ra[PER_A], rb[PER_B];
  auto gload = [&](int kt) {
#pragma unroll
    for (int i = 0; i < PER_A; ++i) {
      const int id = i * THREADS + tid;
      const int c = id % CPR, row = (id / CPR) % BM, piece = id / (BM * CPR);
      ra[i] = *reinterpret_cast<const u32x4*>(Ab + piece * planeA + (long)row * a.K + kt * BK + c * 8);
    }
#pragma unroll
    for (int i = 0; i < PER_B; ++i) {
      const int id = i * THREADS + tid;
      const int c = id % CPR, row = (id / CPR) % BN, piece = id / (BN * CPR);
      rb[i] = *reinterpret_cast<const u32x4*>(Bb + piece * planeB + (long)row * a.K + kt * BK + c * 8);
    }
  };
  auto sstore = [&]() {
#pragma unroll
    for (int i = 0; i < PER_A; ++i) {
      const int id = i * THREADS + tid;
      const int c = id % CPR, row = (id / CPR) % BM, piece = id / (BM * CPR);
      *reinterpret_cast<u32x4*>(sA + piece * TA + row * RS + c * 16) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < PER_B; ++i) {
      const int id = i * THREADS + tid;
      const int c = id % CPR, row = (id / CPR) % BN, piece = id / (BN * CPR);
      *reinterpret_cast<u32x4*>(sB + piece * TB + row * RS + c * 16) = rb[i];
    }
  };
  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
  const int nk = a.K / BK;
  gload(0);
  sstore();
  __syncthreads();
  const unsigned char* pa = sA + (wm * MT * 32 + r) * RS + 16 * g;
  const unsigned char* pb = sB + (wn * NT * 32 + r) * RS + 16 * g;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) gload(kt + 1);
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      bf16x8 A[MT][3], B[NT][3];
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int p = 0; p < 3; ++p) A[t][p] = *reinterpret_cast<const bf16x8*>(pa + p * TA + t * 32 * RS + 32 * s);
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int p = 0; p < 3; ++p) B[t][p] = *reinterpret_cast<const bf16x8*>(pb + p * TB + t * 32 * RS + 32 * s);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][2], B[j][0], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][0], B[j][2], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][1], B[j][1], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][1], B[j][0], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][0], B[j][1], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][0], B[j][0], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
    if (kt + 1 < nk) {
      sstore();
      __syncthreads();
    }
  }
  float* C = a.C + (long)f * a.M * a.N;
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int rr = (q & 3) + 8 * (q >> 2) + 4 * g;
        const long m = (long)blockIdx.x * BM + (wm * MT + i) * 32 + rr;
        const long n = (long)blockIdx.y * BN + (wn * NT + j) * 32 + r;
        C[m * a.N + n] = acc[i][j][q];
      }
}

template <int WM, int WN, int MT, int NT>
void run_big(const char* name, const Args& a, const std::vector<float>& hA, const std::vector<float>& hB) {
  constexpr int BM = WM * MT * 32, BN = WN * NT * 32, RS = 80;
  if (a.M % BM || a.N % BN) { printf("%-26s skipped (shape)\n", name); return; }
  const size_t lds = (size_t)3 * (BM + BN) * RS;
  auto kern = gemm_big_kernel<WM, WN, MT, NT>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  dim3 grid(a.M / BM, a.N / BN, a.F);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), lds, 0, a);
  hipEventRecord(e0);
  const int reps = 5;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), lds, 0, a);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  const double tf = 2.0 * a.F * (double)a.M * a.N * a.K / (ms * 1e-3) / 1e12;
  std::vector<float> c(64 * 64);
  for (int m = 0; m < 64; ++m) hipMemcpy(&c[m * 64], a.C + (long)m * a.N, 64 * 4, hipMemcpyDeviceToHost);
  double num = 0, den = 0;
  for (int m = 0; m < 64; ++m)
    for (int n = 0; n < 64; ++n) {
      double s = 0;
      for (int k = 0; k < a.K; ++k) s += (double)hA[(long)m * a.K + k] * (double)hB[(long)n * a.K + k];
      num += (s - c[m * 64 + n]) * (s - c[m * 64 + n]);
      den += s * s;
    }
  printf("%-26s grid %4dx%-3dx%-2d lds %6zu  %8.3f ms  %7.1f TF/s-equiv  relL2 %.2e\n", name, grid.x, grid.y, grid.z,
         lds, ms, tf, sqrt(num / den));
}

// DMA-staged variant: global_load_lds (16 bytes per lane, lane-linear LDS image) with an XOR swizzle
// applied on the SOURCE address and on the fragment READ; BK = 16 (32-byte rows), two LDS buffers,
// one barrier per K step, no VGPR staging and no ds_write pass.
template <int WM, int WN, int MT, int NT>
__global__ __launch_bounds__(WM * WN * 64) void gemm_glds_kernel(Args a) {
  constexpr int BK = 16, RB = BK * 2, THREADS = WM * WN * 64, WAVES = WM * WN;
  constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
  constexpr int TA = BM * RB, TB = BN * RB, BUF = 3 * (TA + TB);
  // one glds instruction of a wave fills 64 consecutive 16-byte slots = 32 rows x 2 chunks
  constexpr int ROWS_PER_INST = 32;
  constexpr int INST_A = 3 * BM / ROWS_PER_INST, INST_B = 3 * BN / ROWS_PER_INST;   // per K step, whole block
  constexpr int PER_WAVE = (INST_A + INST_B) / WAVES;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, g = lane >> 5;
  const int f = blockIdx.z;
  const long planeA = (long)a.F * a.M * a.K, planeB = (long)a.F * a.N * a.K;
  const u16* Ab = a.Ap + ((long)f * a.M + (long)blockIdx.x * BM) * a.K;
  const u16* Bb = a.Bp + ((long)f * a.N + (long)blockIdx.y * BN) * a.K;
  // lane -> (row within the 32-row group, physical chunk); logical chunk = physical ^ swz(row)
  const int lrow = lane >> 1, pc = lane & 1;
  auto issue = [&](int kt, int buf) {
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
      const int inst = wave * PER_WAVE + i;            // 0 .. INST_A + INST_B - 1
      const bool isA = inst < INST_A;
      const int li = isA ? inst : inst - INST_A;
      const int rows_per_piece = (isA ? BM : BN) / ROWS_PER_INST;
      const int piece = li / rows_per_piece, rg = li % rows_per_piece;
      const int row = rg * ROWS_PER_INST + lrow;
      const int c = pc ^ ((row >> 3) & 1);
      const u16* src = (isA ? Ab + piece * planeA : Bb + piece * planeB) + (long)row * a.K + kt * BK + c * 8;
      unsigned char* dst = smem + buf * BUF + (isA ? piece * TA : 3 * TA + piece * TB) + rg * ROWS_PER_INST * RB;
      __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  };
  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
  const int nk = a.K / BK;
  issue(0, 0);
  __syncthreads();
  // fragment byte offset inside a piece tile: row * 32 + 16 * (g ^ swz(row)); rows of a 32-row MFMA tile
  const int sw = (r >> 3) & 1;
  const int fa = (wm * MT * 32 + r) * RB + 16 * (g ^ sw);
  const int fb = (wn * NT * 32 + r) * RB + 16 * (g ^ sw);
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) issue(kt + 1, cur ^ 1);
    const unsigned char* base = smem + cur * BUF;
    bf16x8 A[MT][3], B[NT][3];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int p = 0; p < 3; ++p) A[t][p] = *reinterpret_cast<const bf16x8*>(base + p * TA + fa + t * 32 * RB);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int p = 0; p < 3; ++p) B[t][p] = *reinterpret_cast<const bf16x8*>(base + 3 * TA + p * TB + fb + t * 32 * RB);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][2], B[j][0], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][0], B[j][2], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][1], B[j][1], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][1], B[j][0], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][0], B[j][1], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][0], B[j][0], acc[i][j], 0, 0, 0);
      }
    __syncthreads();
  }
  float* C = a.C + (long)f * a.M * a.N;
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int rr = (q & 3) + 8 * (q >> 2) + 4 * g;
        const long m = (long)blockIdx.x * BM + (wm * MT + i) * 32 + rr;
        const long n = (long)blockIdx.y * BN + (wn * NT + j) * 32 + r;
        C[m * a.N + n] = acc[i][j][q];
      }
}

template <int WM, int WN, int MT, int NT>
void run_glds(const char* name, const Args& a, const std::vector<float>& hA, const std::vector<float>& hB) {
  constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
  if (a.M % BM || a.N % BN) { printf("%-26s skipped (shape)\n", name); return; }
  const size_t lds = (size_t)2 * 3 * (BM + BN) * 32;
  auto kern = gemm_glds_kernel<WM, WN, MT, NT>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  dim3 grid(a.M / BM, a.N / BN, a.F);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), lds, 0, a);
  hipEventRecord(e0);
  const int reps = 5;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), lds, 0, a);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  const double tf = 2.0 * a.F * (double)a.M * a.N * a.K / (ms * 1e-3) / 1e12;
  std::vector<float> c(64 * 64);
  for (int m = 0; m < 64; ++m) hipMemcpy(&c[m * 64], a.C + (long)m * a.N, 64 * 4, hipMemcpyDeviceToHost);
  double num = 0, den = 0;
  for (int m = 0; m < 64; ++m)
    for (int n = 0; n < 64; ++n) {
      double s = 0;
      for (int k = 0; k < a.K; ++k) s += (double)hA[(long)m * a.K + k] * (double)hB[(long)n * a.K + k];
      num += (s - c[m * 64 + n]) * (s - c[m * 64 + n]);
      den += s * s;
    }
  printf("%-26s grid %4dx%-3dx%-2d lds %6zu  %8.3f ms  %7.1f TF/s-equiv  relL2 %.2e\n", name, grid.x, grid.y, grid.z,
         lds, ms, tf, sqrt(num / den));
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 16384, N = argc > 2 ? atoi(argv[2]) : 1024, K = argc > 3 ? atoi(argv[3]) : 256;
  const int F = argc > 4 ? atoi(argv[4]) : 16;
  const long nA = (long)F * M * K, nB = (long)F * N * K;
  std::vector<float> hA(nA), hB(nB);
  for (auto& x : hA) x = (rand() / (float)RAND_MAX) * 2 - 1;
  for (auto& x : hB) x = (rand() / (float)RAND_MAX) * 2 - 1;
  float *dA, *dB, *dC;
  u16 *pA, *pB;
  hipMalloc(&dA, nA * 4); hipMalloc(&dB, nB * 4); hipMalloc(&dC, (size_t)F * M * N * 4);
  hipMalloc(&pA, nA * 6); hipMalloc(&pB, nB * 6);
  hipMemcpy(dA, hA.data(), nA * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, hB.data(), nB * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(split_kernel, dim3(4096), dim3(256), 0, 0, dA, nA, pA, pA + nA, pA + 2 * nA);
  hipLaunchKernelGGL(split_kernel, dim3(4096), dim3(256), 0, 0, dB, nB, pB, pB + nB, pB + 2 * nB);
  hipDeviceSynchronize();
  Args a{pA, pB, dC, M, N, K, F};
  printf("M=%d N=%d K=%d F=%d\n", M, N, K, F);
  run<32, false, 6>("bk32 single 6t", a, hA, hB);
  run<32, true, 6>("bk32 double 6t", a, hA, hB);
  run<16, true, 6>("bk16 double 6t", a, hA, hB);
  run<16, false, 6>("bk16 single 6t", a, hA, hB);
  run_big<2, 4, 4, 2>("256x256 8w (128x64/wave)", a, hA, hB);
  hipMemset(a.C, 0, (size_t)a.F * a.M * a.N * 4);
  run_glds<2, 4, 4, 2>("glds 256x256 8w bk16", a, hA, hB);
  run_glds<2, 2, 2, 2>("glds 128x128 4w bk16", a, hA, hB);
  run_big<4, 2, 2, 4>("256x256 8w (64x128/wave)", a, hA, hB);
  run_big<2, 4, 2, 2>("128x256 8w (64x64/wave)", a, hA, hB);
  run_big<4, 2, 2, 2>("256x128 8w (64x64/wave)", a, hA, hB);
  run_big<2, 2, 4, 2>("256x128 4w (128x64/wave)", a, hA, hB);
  run<32, false, 3>("bk32 single 3t", a, hA, hB);
  run<32, false, 1>("bk32 single 1t", a, hA, hB);
  return 0;
}
