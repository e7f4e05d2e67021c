// Prototype: fp32-accurate GEMM on the bf16 MFMA pipe ("bf16x3", 6-term product expansion).
//   a = a1 + a2 + a3 (three bf16 pieces, 24 mantissa bits), b likewise;
//   a*b ~= a1b1 + (a1b2 + a2b1) + (a1b3 + a2b2 + a3b1)      (dropped terms <= 2^-23 |ab|)
// Six v_mfma_f32_32x32x16_bf16 (32 cycles each) replace eight v_mfma_f32_32x32x2_f32 (64 cycles
// each) per K=16 slab: 2.67x the fp32-MFMA rate at fp32-class accuracy.
// C[M,N] = A[M,K] . B[N,K]^T, A and B fp32 row-major (K contiguous).  Dev tool / experiment.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/ablate/gemm_bf16x3.hip -o /tmp/gemm_bf16x3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

constexpr int BM = 128, BN = 128, BK = 16, THREADS = 256;
// LDS tile: [rows][3 planes][16 bf16]  -> 96 bytes per row (+ pad to break bank periodicity)
constexpr int ROWB = 3 * 32 + 16;  // bytes per row (112: 28 dwords, odd multiple of 4 dwords)
constexpr int TILEB = BM * ROWB;

__device__ __forceinline__ unsigned bf16_rne(float x) {  // returns the 16-bit pattern
  unsigned u = __float_as_uint(x);
  u += 0x7fffu + ((u >> 16) & 1u);
  return u >> 16;
}
// split x into three bf16 pieces
__device__ __forceinline__ void split3(float x, unsigned& p1, unsigned& p2, unsigned& p3) {
  p1 = bf16_rne(x);
  const float r1 = x - __uint_as_float(p1 << 16);
  p2 = bf16_rne(r1);
  const float r2 = r1 - __uint_as_float(p2 << 16);
  p3 = bf16_rne(r2);
}

__device__ __forceinline__ unsigned cvt_pk(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
// two floats -> three packed bf16 pairs (hardware RNE conversion)
__device__ __forceinline__ void split3_pair(float x, float y, unsigned& p1, unsigned& p2, unsigned& p3) {
  p1 = cvt_pk(x, y);
  const float rx = x - __uint_as_float(p1 << 16), ry = y - __uint_as_float(p1 & 0xffff0000u);
  p2 = cvt_pk(rx, ry);
  const float sx = rx - __uint_as_float(p2 << 16), sy = ry - __uint_as_float(p2 & 0xffff0000u);
  p3 = cvt_pk(sx, sy);
}

template <int NTERMS>
__global__ __launch_bounds__(THREADS) void gemm_bf16x3_kernel(const float* __restrict__ A,
                                                             const float* __restrict__ B,
                                                             float* __restrict__ C, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sA = smem;                 // 2 buffers
  unsigned char* sB = smem + 2 * TILEB;
  const int tm = blockIdx.x, tn = blockIdx.y;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  // staging map: thread -> (row = t / 4 + 64*p, k-quad c = t % 4)
  const int c4 = t & 3, r0 = t >> 2;
  const float* Ap = A + (long)(tm * BM + r0) * K + 4 * c4;
  const float* Bp = B + (long)(tn * BN + r0) * K + 4 * c4;
  float4 ra[2], rb[2];
  auto gload = [&](int kt) {
    ra[0] = *reinterpret_cast<const float4*>(Ap + kt * BK);
    ra[1] = *reinterpret_cast<const float4*>(Ap + (long)64 * K + kt * BK);
    rb[0] = *reinterpret_cast<const float4*>(Bp + kt * BK);
    rb[1] = *reinterpret_cast<const float4*>(Bp + (long)64 * K + kt * BK);
  };
  auto split_store = [&](unsigned char* tile, int row, float4 v) {
    unsigned a1[2], a2[2], a3[2];
    split3_pair(v.x, v.y, a1[0], a2[0], a3[0]);
    split3_pair(v.z, v.w, a1[1], a2[1], a3[1]);
    unsigned char* d = tile + row * ROWB + c4 * 8;
    *reinterpret_cast<uint2*>(d) = make_uint2(a1[0], a1[1]);
    *reinterpret_cast<uint2*>(d + 32) = make_uint2(a2[0], a2[1]);
    *reinterpret_cast<uint2*>(d + 64) = make_uint2(a3[0], a3[1]);
  };
  auto lstore = [&](int buf) {
    split_store(sA + buf * TILEB, r0, ra[0]);
    split_store(sA + buf * TILEB, r0 + 64, ra[1]);
    split_store(sB + buf * TILEB, r0, rb[0]);
    split_store(sB + buf * TILEB, r0 + 64, rb[1]);
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int nkt = K / BK;
  gload(0);
  lstore(0);
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nkt) gload(kt + 1);
    const unsigned char* pa = sA + cur * TILEB + (wm * 64 + li) * ROWB + lh * 16;
    const unsigned char* pb = sB + cur * TILEB + (wn * 64 + li) * ROWB + lh * 16;
    bf16x8 fa[2][3], fb[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        fa[i][p] = *reinterpret_cast<const bf16x8*>(pa + i * 32 * ROWB + p * 32);
        fb[i][p] = *reinterpret_cast<const bf16x8*>(pb + i * 32 * ROWB + p * 32);
      }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x16 c = acc[i][j];
        // small terms first
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
    if (kt + 1 < nkt) lstore(cur ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = tm * BM + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int col = tn * BN + wn * 64 + j * 32 + li;
        C[(long)row * N + col] = acc[i][j][r];
      }
}

template <int NTERMS>
void run(const char* name, const float* A, const float* B, float* C, int M, int N, int K,
         const std::vector<double>& ref, const std::vector<int>& samp) {
  size_t lds = 4 * TILEB;
  auto kern = gemm_bf16x3_kernel<NTERMS>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  dim3 grid(M / BM, N / BN);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, grid, dim3(THREADS), lds, 0, A, B, C, M, N, K);
  hipEventRecord(e0);
  const int reps = 5;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, grid, dim3(THREADS), lds, 0, A, B, C, M, N, K);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  std::vector<float> h(samp.size());
  double num = 0, den = 0, mx = 0;
  for (size_t s = 0; s < samp.size(); ++s) {
    hipMemcpy(&h[s], C + samp[s], sizeof(float), hipMemcpyDeviceToHost);
    const double d = h[s] - ref[s];
    num += d * d;
    den += ref[s] * ref[s];
    mx = fmax(mx, fabs(d));
  }
  printf("%-12s grid %4dx%-3d lds %6zu  %8.3f ms  %7.1f TF/s-equiv   relL2 %.3e  maxabs %.3e\n", name,
         grid.x, grid.y, lds, ms, 2.0 * M * N * (double)K / (ms * 1e-3) / 1e12, sqrt(num / den), mx);
}

int main(int argc, char** argv) {
  int M = argc > 1 ? atoi(argv[1]) : 16384, N = argc > 2 ? atoi(argv[2]) : 512, K = argc > 3 ? atoi(argv[3]) : 4608;
  std::vector<float> ha((size_t)M * K), hb((size_t)N * K);
  srand(1);
  for (auto& x : ha) x = (rand() / (float)RAND_MAX) * 2 - 1;
  for (auto& x : hb) x = ((rand() / (float)RAND_MAX) * 2 - 1) * 0.05f;
  float *A, *B, *C;
  hipMalloc(&A, sizeof(float) * ha.size());
  hipMalloc(&B, sizeof(float) * hb.size());
  hipMalloc(&C, sizeof(float) * (size_t)M * N);
  hipMemcpy(A, ha.data(), sizeof(float) * ha.size(), hipMemcpyHostToDevice);
  hipMemcpy(B, hb.data(), sizeof(float) * hb.size(), hipMemcpyHostToDevice);
  // fp64 reference on sampled outputs (asymmetric positions: catches transposes)
  std::vector<int> samp;
  std::vector<double> ref;
  for (int s = 0; s < 512; ++s) {
    const int r = (int)(((long)s * 7919 + 13) % M), c = (int)(((long)s * 104729 + 7) % N);
    samp.push_back(r * N + c);
    double acc = 0;
    for (int k = 0; k < K; ++k) acc += (double)ha[(size_t)r * K + k] * (double)hb[(size_t)c * K + k];
    ref.push_back(acc);
  }
  printf("M=%d N=%d K=%d\n", M, N, K);
  {  // what the exact-fp32 MFMA path produces: a k-ordered fmaf chain
    double num = 0, den = 0;
    for (size_t s = 0; s < samp.size(); ++s) {
      const int r = samp[s] / N, c = samp[s] % N;
      float acc = 0.f;
      for (int k = 0; k < K; ++k) acc = fmaf(ha[(size_t)r * K + k], hb[(size_t)c * K + k], acc);
      num += (acc - ref[s]) * (acc - ref[s]);
      den += ref[s] * ref[s];
    }
    printf("fp32 fmaf chain (== v_mfma_f32_32x32x2_f32)                              relL2 %.3e\n", sqrt(num / den));
  }
  run<6>("bf16x3 6t", A, B, C, M, N, K, ref, samp);
  run<3>("bf16x3 3t", A, B, C, M, N, K, ref, samp);
  run<1>("bf16 1t", A, B, C, M, N, K, ref, samp);
  return 0;
}
