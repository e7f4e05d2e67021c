// Prototype 2: bf16x3 GEMM with PRE-SPLIT operands (three bf16 planes per fp32 tensor, produced
// once by a separate pass) -- the hot loop has no conversion work.  Dev tool / experiment.
// C[M,N] = A[M,K] . B[N,K]^T ; planes Ap[3][M][K], Bp[3][N][K] (bf16).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

__device__ __forceinline__ unsigned bf16_rne(float x) {
  unsigned u = __float_as_uint(x);
  u += 0x7fffu + ((u >> 16) & 1u);
  return u >> 16;
}
__global__ void split_kernel(const float* __restrict__ x, long n, u16* __restrict__ p1,
                             u16* __restrict__ p2, u16* __restrict__ p3) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = x[i];
    const unsigned a1 = bf16_rne(v);
    const float r1 = v - __uint_as_float(a1 << 16);
    const unsigned a2 = bf16_rne(r1);
    const float r2 = r1 - __uint_as_float(a2 << 16);
    p1[i] = (u16)a1; p2[i] = (u16)a2; p3[i] = (u16)bf16_rne(r2);
  }
}

template <int WM_, int WN_, int MT_, int NT_, int BK_>
struct Cfg {
  static constexpr int WM = WM_, WN = WN_, MT = MT_, NT = NT_, BK = BK_;
  static constexpr int THREADS = 64 * WM * WN, BM = WM * MT * 32, BN = WN * NT * 32;
  static constexpr int ROWB = 3 * BK * 2 + 16;   // bytes per LDS row: 3 planes x BK bf16 (+pad)
};

template <class C, int NTERMS>
__global__ __launch_bounds__(C::THREADS) void gemm_kernel(const u16* __restrict__ Ap, const u16* __restrict__ Bp,
                                                         float* __restrict__ Cm, int M, int N, int K) {
  constexpr int BM = C::BM, BN = C::BN, BK = C::BK, ROWB = C::ROWB, MT = C::MT, NT = C::NT;
  constexpr int CPR = BK / 8;                      // 16-byte chunks per row per plane
  constexpr int A_CHUNKS = BM * 3 * CPR, B_CHUNKS = BN * 3 * CPR;
  constexpr int A_PER = (A_CHUNKS + C::THREADS - 1) / C::THREADS, B_PER = (B_CHUNKS + C::THREADS - 1) / C::THREADS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sA = smem;
  unsigned char* sB = smem + 2 * BM * ROWB;
  const int tm = blockIdx.x, tn = blockIdx.y;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave / C::WN, wn = wave % C::WN;
  const int li = lane & 31, lh = lane >> 5;
  const long planeA = (long)M * K, planeB = (long)N * K;
  uint4 ra[A_PER], rb[B_PER];
  auto gload = [&](int kt) {
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const int id = t + i * C::THREADS;          // chunk id: (row, plane, c)
      const int c = id % CPR, pl = (id / CPR) % 3, row = id / (3 * CPR);
      if (id < A_CHUNKS)
        ra[i] = *reinterpret_cast<const uint4*>(Ap + pl * planeA + (long)(tm * BM + row) * K + kt * BK + c * 8);
    }
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
      const int id = t + i * C::THREADS;
      const int c = id % CPR, pl = (id / CPR) % 3, row = id / (3 * CPR);
      if (id < B_CHUNKS)
        rb[i] = *reinterpret_cast<const uint4*>(Bp + pl * planeB + (long)(tn * BN + row) * K + kt * BK + c * 8);
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const int id = t + i * C::THREADS;
      const int c = id % CPR, pl = (id / CPR) % 3, row = id / (3 * CPR);
      if (id < A_CHUNKS) *reinterpret_cast<uint4*>(sA + buf * BM * ROWB + row * ROWB + pl * BK * 2 + c * 16) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
      const int id = t + i * C::THREADS;
      const int c = id % CPR, pl = (id / CPR) % 3, row = id / (3 * CPR);
      if (id < B_CHUNKS) *reinterpret_cast<uint4*>(sB + buf * BN * ROWB + row * ROWB + pl * BK * 2 + c * 16) = rb[i];
    }
  };
  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int nkt = K / BK;
  gload(0);
  lstore(0);
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nkt) gload(kt + 1);
    const unsigned char* pa = sA + cur * BM * ROWB + (wm * MT * 32 + li) * ROWB + lh * 16;
    const unsigned char* pb = sB + cur * BN * ROWB + (wn * NT * 32 + li) * ROWB + lh * 16;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8 fa[MT][3], fb[NT][3];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int i = 0; i < MT; ++i) fa[i][p] = *reinterpret_cast<const bf16x8*>(pa + i * 32 * ROWB + p * BK * 2 + ks * 32);
#pragma unroll
        for (int j = 0; j < NT; ++j) fb[j][p] = *reinterpret_cast<const bf16x8*>(pb + j * 32 * ROWB + p * BK * 2 + ks * 32);
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          f32x16 c = acc[i][j];
          if (NTERMS >= 6) {
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][2], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][2], fb[j][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][1], c, 0, 0, 0);
          }
          if (NTERMS >= 3) {
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][0], c, 0, 0, 0);
          }
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][0], c, 0, 0, 0);
          acc[i][j] = c;
        }
    }
    if (kt + 1 < nkt) lstore(cur ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = tm * BM + wm * MT * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int col = tn * BN + wn * NT * 32 + j * 32 + li;
        Cm[(long)row * N + col] = acc[i][j][r];
      }
}

template <class C, int NTERMS>
void run(const char* name, const u16* Ap, const u16* Bp, float* Cm, int M, int N, int K,
         const std::vector<double>& ref, const std::vector<int>& samp) {
  if (M % C::BM || N % C::BN) { printf("%-26s skipped\n", name); return; }
  size_t lds = 2 * (C::BM + C::BN) * C::ROWB;
  auto kern = gemm_kernel<C, NTERMS>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  dim3 grid(M / C::BM, N / C::BN);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, grid, dim3(C::THREADS), lds, 0, Ap, Bp, Cm, M, N, K);
  hipEventRecord(e0);
  const int reps = 5;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, grid, dim3(C::THREADS), lds, 0, Ap, Bp, Cm, M, N, K);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  double num = 0, den = 0;
  for (size_t s = 0; s < samp.size(); ++s) {
    float h; hipMemcpy(&h, Cm + samp[s], sizeof(float), hipMemcpyDeviceToHost);
    num += (h - ref[s]) * (h - ref[s]); den += ref[s] * ref[s];
  }
  printf("%-26s grid %4dx%-3d lds %6zu  %8.3f ms  %7.1f TF/s-equiv   relL2 %.3e\n", name, grid.x, grid.y,
         lds, ms, 2.0 * M * N * (double)K / (ms * 1e-3) / 1e12, sqrt(num / den));
}

int main(int argc, char** argv) {
  int M = argc > 1 ? atoi(argv[1]) : 16384, N = argc > 2 ? atoi(argv[2]) : 512, K = argc > 3 ? atoi(argv[3]) : 4608;
  std::vector<float> ha((size_t)M * K), hb((size_t)N * K);
  srand(1);
  for (auto& x : ha) x = (rand() / (float)RAND_MAX) * 2 - 1;
  for (auto& x : hb) x = ((rand() / (float)RAND_MAX) * 2 - 1) * 0.05f;
  float *A, *B, *Cm;
  u16 *Ap, *Bp;
  hipMalloc(&A, sizeof(float) * ha.size()); hipMalloc(&B, sizeof(float) * hb.size());
  hipMalloc(&Cm, sizeof(float) * (size_t)M * N);
  hipMalloc(&Ap, 6 * ha.size()); hipMalloc(&Bp, 6 * hb.size());
  hipMemcpy(A, ha.data(), sizeof(float) * ha.size(), hipMemcpyHostToDevice);
  hipMemcpy(B, hb.data(), sizeof(float) * hb.size(), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(split_kernel, dim3(2048), dim3(256), 0, 0, A, (long)ha.size(), Ap, Ap + ha.size(), Ap + 2 * ha.size());
  hipLaunchKernelGGL(split_kernel, dim3(2048), dim3(256), 0, 0, B, (long)hb.size(), Bp, Bp + hb.size(), Bp + 2 * hb.size());
  std::vector<int> samp; std::vector<double> ref;
  for (int s = 0; s < 256; ++s) {
    const int r = (int)(((long)s * 7919 + 13) % M), c = (int)(((long)s * 104729 + 7) % N);
    samp.push_back(r * N + c);
    double acc = 0;
    for (int k = 0; k < K; ++k) acc += (double)ha[(size_t)r * K + k] * (double)hb[(size_t)c * K + k];
    ref.push_back(acc);
  }
  printf("M=%d N=%d K=%d  (pre-split operands)\n", M, N, K);
  run<Cfg<2, 2, 2, 2, 16>, 6>("128x128 bk16 4w 6t", Ap, Bp, Cm, M, N, K, ref, samp);
  run<Cfg<2, 2, 2, 2, 32>, 6>("128x128 bk32 4w 6t", Ap, Bp, Cm, M, N, K, ref, samp);
  run<Cfg<2, 2, 2, 2, 64>, 6>("128x128 bk64 4w 6t", Ap, Bp, Cm, M, N, K, ref, samp);
  run<Cfg<4, 2, 2, 2, 32>, 6>("256x128 bk32 8w 6t", Ap, Bp, Cm, M, N, K, ref, samp);
  run<Cfg<2, 2, 4, 2, 32>, 6>("256x128 bk32 4w 6t", Ap, Bp, Cm, M, N, K, ref, samp);
  run<Cfg<2, 2, 4, 4, 16>, 6>("256x256 bk16 4w 6t", Ap, Bp, Cm, M, N, K, ref, samp);
  run<Cfg<4, 2, 2, 4, 16>, 6>("256x256 bk16 8w 6t", Ap, Bp, Cm, M, N, K, ref, samp);
  run<Cfg<2, 2, 2, 2, 32>, 3>("128x128 bk32 4w 3t", Ap, Bp, Cm, M, N, K, ref, samp);
  run<Cfg<2, 2, 2, 2, 32>, 1>("128x128 bk32 4w 1t", Ap, Bp, Cm, M, N, K, ref, samp);
  return 0;
}
