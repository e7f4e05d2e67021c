// Prototype 4: the split-precision NT GEMM with ONE wave per SIMD.
//   256 x 256 block tile, 4 waves (2 x 2), wave tile 128 x 128 = 16 accumulator tiles (256 AGPRs),
//   BK = 16 per stage, three LDS stages filled by global_load_lds (no VGPR staging, no ds_write pass),
//   fragments double-buffered in VGPRs: the 24 ds_read_b128 of stage k+1 are interleaved with the
//   96 MFMAs of stage k (one read per four MFMAs), one barrier per stage.
// Motivation: in the 8-wave kernel (252 VGPRs, two waves per SIMD) the fragments are read just in time,
// so every few MFMAs wait for an LDS round trip (ISA: ds_read ... s_waitcnt lgkmcnt(0) ... v_mfma).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/ablate/gemm_bf16x3_v3.hip -o /tmp/gemm_bf16x3_v3
#include <type_traits>
#define main main_v2
#include "gemm_bf16x3_v2.hip"
#undef main

namespace v3 {
constexpr int BK = 16, RB = 32, MT = 4, NT = 4, BM = 256, BN = 256, NBUF = 3;
constexpr int TA = BM * RB, TB = BN * RB, BUF = 3 * (TA + TB);
constexpr int INST = 3 * (BM + BN) / 32, PER_WAVE = INST / 4;

struct Frags {
  bf16x8 a[MT][3];
  bf16x8 b[NT][3];
};

template <int ORDER>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_v3_kernel(Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, g = lane >> 5;
  const int f = blockIdx.z;
  const long planeA = (long)a.F * a.M * a.K, planeB = (long)a.F * a.N * a.K;
  // ORDER >= 2: every block reads tile (0, 0) of frequency 0 (all operand loads hit in L2): separates the
  // in-core limit from the memory-system limit
  const u16* Ab = a.Ap + ((ORDER == 2 || ORDER == 3 || ORDER >= 5) ? 0 : ((long)f * a.M + (long)blockIdx.x * BM) * a.K);
  const u16* Bb = a.Bp + ((ORDER == 2 || ORDER == 3 || ORDER >= 5) ? 0 : ((long)f * a.N + (long)blockIdx.y * BN) * a.K);
  const int lrow = lane >> 1, pc = lane & 1;
  // per-lane byte offset inside a 32-row group (same for every instruction): row * K * 2 + chunk * 16
  const unsigned voff = (unsigned)lrow * (unsigned)a.K * 2u + (unsigned)((pc ^ ((lrow >> 3) & 1)) * 16);
  // waves 0,1 stage A (24 instructions of 32 rows: 3 pieces x 8 row groups), waves 2,3 stage B
  const bool isA = wave < 2;
  const u16* opb = isA ? Ab : Bb;
  const long plane = isA ? planeA : planeB;
  const int half = wave & 1;
  auto issue = [&](int kt, int buf) {
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
      const int li = half * PER_WAVE + i;          // 0..23 within the operand
      const int piece = li >> 3, rg = li & 7;
      const u16* sb = opb + piece * plane + (long)rg * 32 * a.K + kt * BK;
      if (ORDER >= 4) {
        // timing experiment: operands stored tile-blocked [row block of 32][k block of 16][32 rows][16 k], one
        // instruction = one contiguous 1 KiB chunk (the data read differ from the row-major case: timing only)
        sb = opb + piece * plane + ((long)rg * (a.K / BK) + kt) * 512 - (voff >> 1) + lane * 8;
      }
      unsigned char* dst = smem + buf * BUF + (isA ? 0 : 3 * TA) + piece * TA + rg * 32 * RB;
      __builtin_amdgcn_global_load_lds(reinterpret_cast<const unsigned char*>(sb) + voff,
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  };
  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
  const int nk = a.K / BK;   // even, >= 4
  const int sw = (r >> 3) & 1;
  const int fa = (wm * MT * 32 + r) * RB + 16 * (g ^ sw);
  const int fb = 3 * TA + (wn * NT * 32 + r) * RB + 16 * (g ^ sw);

  auto load_frags = [&](Frags& F, int buf) {
    const unsigned char* pa = smem + buf * BUF + fa;
    const unsigned char* pb = smem + buf * BUF + fb;
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int p = 0; p < 3; ++p) F.a[t][p] = *reinterpret_cast<const bf16x8*>(pa + p * TA + t * 32 * RB);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int p = 0; p < 3; ++p) F.b[t][p] = *reinterpret_cast<const bf16x8*>(pb + p * TB + t * 32 * RB);
  };
  auto mfmas = [&](const Frags& F) {
    constexpr int TA_[6] = {2, 0, 1, 1, 0, 0}, TB_[6] = {0, 2, 1, 0, 1, 0};   // smallest terms first
    if (ORDER == 0 || ORDER == 2) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.a[i][TA_[t]], F.b[j][TB_[t]], acc[i][j], 0, 0, 0);
    } else {
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.a[i][TA_[t]], F.b[j][TB_[t]], acc[i][j], 0, 0, 0);
    }
  };
  // one stage: wait for stage kt+1 in LDS, refill the buffer stage kt used, then 96 MFMAs on F with the
  // 24 fragment reads of stage kt+1 interleaved.  ISSUE: stage kt+3 exists; PEND: stage kt+2 is in flight;
  // LOAD: stage kt+1 exists.
  auto stage = [&](int kt, int bufn, const Frags& F, Frags& G, auto issue_c, auto pend_c, auto load_c) {
    constexpr bool ISSUE = decltype(issue_c)::value, PEND = decltype(pend_c)::value, LOAD = decltype(load_c)::value;
    if (ORDER == 7 || ORDER == 8) {
    } else if (PEND) __builtin_amdgcn_s_waitcnt(0x0f70 | (PER_WAVE & 15) | ((PER_WAVE >> 4) << 14));   // vmcnt(12)
    else __builtin_amdgcn_s_waitcnt(0x0f70);                                                      // vmcnt(0)
    // in-core breakdown experiments (wrong results, timing only): 6 = no barrier, 7 = no refills, 8 = no fragment reads
    if (ORDER != 6) __builtin_amdgcn_s_barrier();
    if (ISSUE && ORDER != 7 && ORDER != 8) issue(kt + 3, bufn == 0 ? 2 : bufn - 1);   // the buffer stage kt was read from
    if (LOAD && ORDER != 8) load_frags(G, bufn);
    mfmas(F);
    if (LOAD && ORDER != 8) {
#pragma unroll
      for (int q = 0; q < 24; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // one DS read
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);   // four MFMAs
      }
    }
  };
  using T = std::true_type;
  using Fl = std::false_type;
  Frags F0, F1;
  issue(0, 0);
  issue(1, 1);
  issue(2, 2);
  __builtin_amdgcn_s_waitcnt(0x0f70 | ((2 * PER_WAVE) & 15) | (((2 * PER_WAVE) >> 4) << 14));
  __builtin_amdgcn_s_barrier();
  load_frags(F0, 0);
  int kt = 0, bufn = 1;   // bufn = (kt + 1) % 3
  for (; kt + 6 <= nk; kt += 2) {
    stage(kt, bufn, F0, F1, T{}, T{}, T{});
    bufn = bufn == 2 ? 0 : bufn + 1;
    stage(kt + 1, bufn, F1, F0, T{}, T{}, T{});
    bufn = bufn == 2 ? 0 : bufn + 1;
  }
  // tail: stages nk-4 .. nk-1
  stage(kt, bufn, F0, F1, T{}, T{}, T{});
  bufn = bufn == 2 ? 0 : bufn + 1;
  stage(kt + 1, bufn, F1, F0, Fl{}, T{}, T{});
  bufn = bufn == 2 ? 0 : bufn + 1;
  stage(kt + 2, bufn, F0, F1, Fl{}, Fl{}, T{});
  bufn = bufn == 2 ? 0 : bufn + 1;
  stage(kt + 3, bufn, F1, F0, Fl{}, Fl{}, Fl{});
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

__global__ void diff_kernel(const float* x, const float* y, long n, double* out) {
  double num = 0, den = 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const double d = (double)x[i] - (double)y[i];
    num += d * d;
    den += (double)y[i] * (double)y[i];
  }
  atomicAdd(out, num);
  atomicAdd(out + 1, den);
}

template <int ORDER>
float run(const char* name, const Args& a, const float* Cref) {
  const size_t lds = (size_t)NBUF * BUF;
  auto kern = gemm_v3_kernel<ORDER>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  dim3 grid(a.M / BM, a.N / BN, a.F);
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
  double* d;
  hipMalloc(&d, 16);
  hipMemset(d, 0, 16);
  hipLaunchKernelGGL(diff_kernel, dim3(1024), dim3(256), 0, 0, a.C, Cref, (long)a.F * a.M * a.N, d);
  double h[2];
  hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  printf("%-26s grid %4dx%-3dx%-2d lds %6zu  %8.3f ms  %7.1f TF/s-equiv  vs 8-wave kernel relL2 %.2e (%s)\n", name, grid.x,
         grid.y, grid.z, lds, ms, tf, sqrt(h[0] / h[1]), hipGetErrorString(hipGetLastError()));
  return ms;
}
}  // namespace v3

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 16384, N = argc > 2 ? atoi(argv[2]) : 1024, K = argc > 3 ? atoi(argv[3]) : 256;
  const int F = argc > 4 ? atoi(argv[4]) : 16;
  const long nA = (long)F * M * K, nB = (long)F * N * K;
  std::vector<float> hA(nA), hB(nB);
  for (auto& x : hA) x = (rand() / (float)RAND_MAX) * 2 - 1;
  for (auto& x : hB) x = (rand() / (float)RAND_MAX) * 2 - 1;
  float *dA, *dB, *dC, *dRef;
  u16 *pA, *pB;
  hipMalloc(&dA, nA * 4); hipMalloc(&dB, nB * 4); hipMalloc(&dC, (size_t)F * M * N * 4); hipMalloc(&dRef, (size_t)F * M * N * 4);
  hipMalloc(&pA, nA * 6); hipMalloc(&pB, nB * 6);
  hipMemcpy(dA, hA.data(), nA * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, hB.data(), nB * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(split_kernel, dim3(4096), dim3(256), 0, 0, dA, nA, pA, pA + nA, pA + 2 * nA);
  hipLaunchKernelGGL(split_kernel, dim3(4096), dim3(256), 0, 0, dB, nB, pB, pB + nB, pB + 2 * nB);
  hipDeviceSynchronize();
  Args ref{pA, pB, dRef, M, N, K, F};
  printf("M=%d N=%d K=%d F=%d\n", M, N, K, F);
  run_big<2, 4, 4, 2>("256x256 8w (128x64/wave)", ref, hA, hB);
  Args a{pA, pB, dC, M, N, K, F};
  hipMemset(dC, 0, (size_t)F * M * N * 4);
  v3::run<0>("v3 4w 128x128 i-major", a, dRef);
  hipMemset(dC, 0, (size_t)F * M * N * 4);
  v3::run<1>("v3 4w 128x128 term-major", a, dRef);
  v3::run<3>("v3 term-major, L2-resident", a, dRef);
  v3::run<4>("v3 term-major, blocked operands", a, dRef);
  v3::run<5>("v3 blocked + L2-resident", a, dRef);
  v3::run<6>("  .. no barrier", a, dRef);
  v3::run<7>("  .. no refills", a, dRef);
  v3::run<8>("  .. MFMA only", a, dRef);
  return 0;
}
