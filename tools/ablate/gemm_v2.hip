// Micro-benchmark of a k-contiguous ("b128") operand path for the fp32 MFMA block GEMM
// (dev tool, GPU box):  C[M,N] = A[M,K] . B[N,K]^T.
//   LDS tiles [rows][BK + 4] (k contiguous): global -> LDS is a straight float4 copy
//   (ds_write_b128), fragments are ds_read_b128 = four k values per lane, and the contraction
//   index is permuted so that MFMA i of a group contracts k = {8s + i, 8s + 4 + i}.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/ablate/gemm_v2.hip -o /tmp/gemm_v2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MT, int NT, int BK, int PF>
__global__ __launch_bounds__(256) void gemm_v2(const float* __restrict__ A, const float* __restrict__ B,
                                               float* __restrict__ C, int M, int N, int K) {
  constexpr int BM = 2 * MT * 32, BN = 2 * NT * 32, LD = BK + 4, QPR = BK / 4;
  constexpr int NA = BM * QPR / 256, NB = BN * QPR / 256;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sA = smem;                  // [2][BM][LD]
  float* sB = smem + 2 * BM * LD;    // [2][BN][LD]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int row = lane & 31, h = lane >> 5;
  const float* Ab = A + (long)blockIdx.x * BM * K;
  const float* Bb = B + (long)blockIdx.y * BN * K;
  f32x4 ra[NA], rb[NB];
  auto gload = [&](int kt) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int id = i * 256 + tid;
      ra[i] = *reinterpret_cast<const f32x4*>(Ab + (long)(id / QPR) * K + kt * BK + 4 * (id % QPR));
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int id = i * 256 + tid;
      rb[i] = *reinterpret_cast<const f32x4*>(Bb + (long)(id / QPR) * K + kt * BK + 4 * (id % QPR));
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int id = i * 256 + tid;
      *reinterpret_cast<f32x4*>(sA + (buf * BM + id / QPR) * LD + 4 * (id % QPR)) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int id = i * 256 + tid;
      *reinterpret_cast<f32x4*>(sB + (buf * BN + id / QPR) * LD + 4 * (id % QPR)) = rb[i];
    }
  };
  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int nk = K / BK;
  gload(0);
  sstore(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) gload(kt + 1);
    const float* pa = sA + (cur * BM + (wm * MT) * 32 + row) * LD + 4 * h;
    const float* pb = sB + (cur * BN + (wn * NT) * 32 + row) * LD + 4 * h;
    f32x4 a[2][MT], b[2][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) a[0][i] = *reinterpret_cast<const f32x4*>(pa + i * 32 * LD);
#pragma unroll
    for (int j = 0; j < NT; ++j) b[0][j] = *reinterpret_cast<const f32x4*>(pb + j * 32 * LD);
#pragma unroll
    for (int ks = 0; ks < BK / 8; ++ks) {
      const int c = ks & 1;
      if (PF && ks + 1 < BK / 8) {
#pragma unroll
        for (int i = 0; i < MT; ++i) a[c ^ 1][i] = *reinterpret_cast<const f32x4*>(pa + i * 32 * LD + 8 * (ks + 1));
#pragma unroll
        for (int j = 0; j < NT; ++j) b[c ^ 1][j] = *reinterpret_cast<const f32x4*>(pb + j * 32 * LD + 8 * (ks + 1));
      }
      if (!PF && ks > 0) {
#pragma unroll
        for (int i = 0; i < MT; ++i) a[c][i] = *reinterpret_cast<const f32x4*>(pa + i * 32 * LD + 8 * ks);
#pragma unroll
        for (int j = 0; j < NT; ++j) b[c][j] = *reinterpret_cast<const f32x4*>(pb + j * 32 * LD + 8 * ks);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][i][q], b[c][j][q], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) sstore(cur ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rr = (r & 3) + 8 * (r >> 2) + 4 * h;
        const long m = (long)blockIdx.x * BM + (wm * MT + i) * 32 + rr;
        const long n = (long)blockIdx.y * BN + (wn * NT + j) * 32 + row;
        C[m * N + n] = acc[i][j][r];
      }
}

template <int MT, int NT, int BK, int PF>
void run(const char* name, const float* A, const float* B, float* C, int M, int N, int K, std::vector<float>* ref) {
  constexpr int BM = 2 * MT * 32, BN = 2 * NT * 32;
  size_t lds = sizeof(float) * 2 * (BM + BN) * (BK + 4);
  auto kern = gemm_v2<MT, NT, BK, PF>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  dim3 grid(M / BM, N / BN);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, grid, dim3(256), lds, 0, A, B, C, M, N, K);
  hipEventRecord(e0);
  const int reps = 5;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, grid, dim3(256), lds, 0, A, B, C, M, N, K);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  double tf = 2.0 * M * N * (double)K / (ms * 1e-3) / 1e12;
  // spot check against a host fp64 dot of a few entries
  std::vector<float> c(4);
  double err = 0;
  if (ref) {
    const long idx[4] = {0, 12345 % ((long)M * N), (long)M * N / 2 + 77, (long)M * N - 1};
    for (int t = 0; t < 4; ++t) {
      hipMemcpy(&c[t], C + idx[t], 4, hipMemcpyDeviceToHost);
      const long m = idx[t] / N, n = idx[t] % N;
      double s = 0;
      for (int k = 0; k < K; ++k) s += (double)ref[0][m * K + k] * (double)ref[1][n * K + k];
      err = fmax(err, fabs(s - c[t]) / (fabs(s) + 1e-6));
    }
  }
  printf("%-34s grid %4dx%-3d lds %6zu  %8.3f ms  %6.1f TF/s  relerr %.1e\n", name, grid.x, grid.y, lds, ms, tf, err);
}

int main(int argc, char** argv) {
  int M = argc > 1 ? atoi(argv[1]) : 16384, N = argc > 2 ? atoi(argv[2]) : 512, K = argc > 3 ? atoi(argv[3]) : 4608;
  float *A, *B, *C;
  hipMalloc(&A, sizeof(float) * (size_t)M * K); hipMalloc(&B, sizeof(float) * (size_t)N * K);
  hipMalloc(&C, sizeof(float) * (size_t)M * N);
  std::vector<float> ref[2];
  ref[0].resize((size_t)M * K); ref[1].resize((size_t)N * K);
  for (auto& x : ref[0]) x = (rand() / (float)RAND_MAX) * 2 - 1;
  for (auto& x : ref[1]) x = (rand() / (float)RAND_MAX) * 2 - 1;
  hipMemcpy(A, ref[0].data(), sizeof(float) * (size_t)M * K, hipMemcpyHostToDevice);
  hipMemcpy(B, ref[1].data(), sizeof(float) * (size_t)N * K, hipMemcpyHostToDevice);
  printf("M=%d N=%d K=%d\n", M, N, K);
  run<2, 2, 32, 1>("v2 128x128 bk32 prefetch", A, B, C, M, N, K, ref);
  run<2, 2, 32, 0>("v2 128x128 bk32 no-prefetch", A, B, C, M, N, K, ref);
  run<2, 2, 16, 1>("v2 128x128 bk16 prefetch", A, B, C, M, N, K, ref);
  run<2, 2, 64, 1>("v2 128x128 bk64 prefetch", A, B, C, M, N, K, ref);
  run<4, 2, 32, 1>("v2 256x128 bk32 prefetch", A, B, C, M, N, K, ref);
  run<2, 4, 32, 1>("v2 128x256 bk32 prefetch", A, B, C, M, N, K, ref);
  run<1, 2, 32, 1>("v2 64x128 bk32 prefetch", A, B, C, M, N, K, ref);
  run<2, 4, 32, 0>("v2 128x256 bk32 no-prefetch", A, B, C, M, N, K, ref);
  run<2, 4, 16, 0>("v2 128x256 bk16 no-prefetch", A, B, C, M, N, K, ref);
  run<2, 4, 16, 1>("v2 128x256 bk16 prefetch", A, B, C, M, N, K, ref);
  run<4, 2, 16, 0>("v2 256x128 bk16 no-prefetch", A, B, C, M, N, K, ref);
  run<4, 4, 16, 0>("v2 256x256 bk16 no-prefetch", A, B, C, M, N, K, ref);
  run<2, 2, 16, 0>("v2 128x128 bk16 no-prefetch", A, B, C, M, N, K, ref);
  return 0;
}
