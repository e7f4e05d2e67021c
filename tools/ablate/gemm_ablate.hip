// Micro-benchmark of the fp32 MFMA block-GEMM engine (dev tool, run on the GPU box):
// C[M,N] = A[M,K] . B[N,K]^T with the engine of csrc/gemm_tile.h and ablation switches.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I ot-gan_amd/csrc tools/ablate/gemm_ablate.hip -o /tmp/gemm_ablate
#include <vector>
#include <cstdio>
#include <cstdlib>
#include "gemm_tile.h"

void otgan_set_error(const char*, ...) {}
void otgan_prof_begin(int, double, double, hipStream_t) {}
void otgan_prof_end(int, hipStream_t) {}

template <class Cfg, int MODE>
__global__ __launch_bounds__(Cfg::THREADS) void gemm_kernel(const float* A, const float* B, float* C,
                                                           int M, int N, int K) {
  using LA = MatLoaderK<Cfg, Cfg::BM, true>;
  using LB = MatLoaderK<Cfg, Cfg::BN, true>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tm = blockIdx.x, tn = blockIdx.y;
  LA la; LB lb;
  la.init(A + (long)tm * Cfg::BM * K, K, M - tm * Cfg::BM, K);
  lb.init(B + (long)tn * Cfg::BN * K, K, N - tn * Cfg::BN, K);
  typename Cfg::acc_t acc[Cfg::MT][Cfg::NT];
  zero_acc<Cfg>(acc);
  gemm_mainloop<Cfg>(la, lb, K / Cfg::BK, smem, acc);
  foreach_acc<Cfg>(acc, [&](int r, int c, int, int, int, float v) {
    C[(long)(tm * Cfg::BM + r) * N + tn * Cfg::BN + c] = v;
  });
}

template <class Cfg>
double run(const char* name, const float* A, const float* B, float* C, int M, int N, int K, int extra_lds) {
  using LA = MatLoaderK<Cfg, Cfg::BM, true>;
  using LB = MatLoaderK<Cfg, Cfg::BN, true>;
  size_t lds = sizeof(float) * (2 * LA::FLOATS + 2 * LB::FLOATS) + extra_lds;
  auto kern = gemm_kernel<Cfg, 0>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  dim3 grid(M / Cfg::BM, N / Cfg::BN);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, grid, dim3(Cfg::THREADS), lds, 0, A, B, C, M, N, K);
  hipEventRecord(e0);
  const int reps = 5;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, grid, dim3(Cfg::THREADS), lds, 0, A, B, C, M, N, K);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  double tf = 2.0 * M * N * (double)K / (ms * 1e-3) / 1e12;
  printf("%-28s grid %4dx%-3d lds %6zu  %8.3f ms  %6.1f TF/s\n", name, grid.x, grid.y, lds, ms, tf);
  return tf;
}

int main(int argc, char** argv) {
  int M = argc > 1 ? atoi(argv[1]) : 16384, N = argc > 2 ? atoi(argv[2]) : 512, K = argc > 3 ? atoi(argv[3]) : 4608;
  float *A, *B, *C;
  hipMalloc(&A, sizeof(float) * (size_t)M * K); hipMalloc(&B, sizeof(float) * (size_t)N * K);
  hipMalloc(&C, sizeof(float) * (size_t)M * N);
  std::vector<float> h((size_t)M * K);
  for (auto& x : h) x = (rand() / (float)RAND_MAX) * 2 - 1;
  hipMemcpy(A, h.data(), sizeof(float) * (size_t)M * K, hipMemcpyHostToDevice);
  h.resize((size_t)N * K);
  for (auto& x : h) x = (rand() / (float)RAND_MAX) * 2 - 1;
  hipMemcpy(B, h.data(), sizeof(float) * (size_t)N * K, hipMemcpyHostToDevice);
  printf("M=%d N=%d K=%d\n", M, N, K);
  run<GemmCfg<2, 2, 2, 2, 16>>("128x128 bk16 (3 blk/CU)", A, B, C, M, N, K, 0);
  run<GemmCfg<2, 2, 2, 2, 32>>("128x128 bk32", A, B, C, M, N, K, 0);
  run<GemmCfg<2, 2, 2, 2, 64>>("128x128 bk64", A, B, C, M, N, K, 0);
  run<GemmCfg<2, 2, 4, 2, 32>>("256x128 bk32 4 waves", A, B, C, M, N, K, 0);
  run<GemmCfg<2, 2, 2, 4, 32>>("128x256 bk32 4 waves", A, B, C, M, N, K, 0);
  run<GemmCfg<4, 2, 2, 2, 32>>("256x128 bk32 8 waves", A, B, C, M, N, K, 0);
  run<GemmCfg<2, 2, 3, 2, 16>>("192x128 bk16 4 waves", A, B, C, M, N, K, 0);
  run<GemmCfg<1, 4, 2, 1, 32>>("64x128 bk32", A, B, C, M, N, K, 0);
  run<GemmCfg<1, 4, 2, 2, 32>>("64x256 bk32", A, B, C, M, N, K, 0);
  run<GemmCfg<2, 2, 1, 2, 32>>("64x128 bk32 (2x2 waves)", A, B, C, M, N, K, 0);
  return 0;
}
