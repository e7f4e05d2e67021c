// Round 5 prototype: a Winograd-domain GEMM whose workgroup owns a (tile block x Cout block) for ALL 36
// frequencies, so that the output transform can run in its epilogue and the fp32 product tensor [36][T][N] is never written.
//
// What bounds such a kernel is the register file: 36 accumulator planes of Tb x Nb fp32 must fit 512 KiB per compute unit
// (Tb x Nb <= ~2800), and operands of different frequencies share nothing.  Design measured here:
//   * 4 waves per workgroup, one per SIMD; wave w owns frequencies 9 w .. 9 w + 8 of the SAME 64 x 32 block
//     (2 x v_mfma_f32_32x32x16_f16 accumulator tiles per frequency = 288 accumulator registers);
//   * no LDS and no barriers in the K loop: a fragment of the blocked operand layout is one contiguous 1 KiB chunk, loaded
//     straight into the MFMA operand registers (global_load_dwordx4), three (frequency, k16) stages in flight;
//   * operands: two scaled fp16 pieces (hi, lo), three MFMAs per product, as in csrc/gemm_x3.h;
//   * epilogue (optional, -DEPI=1): the 36 planes meet in LDS, quarter by quarter, and every thread applies the output
//     transform A^T M A (6x6 -> 4x4) to its (tile, channel) elements and stores y once.
// L2 -> CU operand demand at the full matrix rate: (64 + 32) rows x 64 B per 6 MFMAs x 32 clocks per wave = 32 B/clk/wave =
// 128 B/clk/CU against the L2's ~64 B/clk/CU (34.5 TB/s): the matrix pipe can be at most half busy.
//
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/ablate/fused_out_proto.hip -o tools/ablate/bin/fused_out_proto
// run:   fused_out_proto [T N K]      (default: the DCGAN generator's last gated layer, forward: 4096 1024 256)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef EPI
#define EPI 1
#endif
#ifndef FPW
#define FPW 9        // frequencies per wave.  9: all 36 (288 accumulator registers: two tiles live in VGPRs and the compiler
#endif               // moves them through the AGPR file inside the loop); 8: 32 of 36, all accumulators in AGPRs (clean timing x 36/32)
constexpr int NF = 36;

struct Args {
  const uint4* A;      // [2 pieces][36][T/32][K/16] chunks of 64 x 16 B (lane l = row l % 32, k 8 (l / 32) ..)
  const uint4* B;      // [2][36][N/32][K/16]
  float* Y;            // EPI: [T][16][N] (tile, output position, channel); else [T/64 * N/32][4 waves][2][64 lanes][16]
  int TB32, NB32, KB;  // T / 32, N / 32, K / 16
  int map;             // 0: channel blocks fastest; 1: XCD-blocked (an XCD's 32 resident workgroups = 4 tile blocks x 8 channel blocks)
};

struct Stage {
  uint4 a[2][2];       // [piece][row block]
  uint4 b[2];          // [piece]
};

__device__ __forceinline__ half8 as_h8(const uint4& v) { return __builtin_bit_cast(half8, v); }

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void fused_kernel(Args g) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ntb = g.TB32 / 2, nnb = g.NB32;
  int tb, nb;
  if (g.map == 0) {
    nb = blockIdx.x % nnb;
    tb = blockIdx.x / nnb;
  } else {
    // workgroup x runs on XCD x % 8 (tools/ablate/xcc_probe.hip); j = x / 8 counts the XCD's workgroups in dispatch order.
    // XCD c works through super-blocks of 4 tile blocks x 8 channel blocks, 32 consecutive j each.
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int sb = j >> 5, in = j & 31;
    const int sbn = nnb / 8;                       // super-blocks along the channels
    const int sid = sb * 8 + xcd;                  // global super-block index
    nb = (sid % sbn) * 8 + (in & 7);
    tb = (sid / sbn) * 4 + (in >> 3);
    if (tb >= ntb) return;
  }
  const long KB = g.KB;
  const long pA = (long)NF * g.TB32 * KB * 64, pB = (long)NF * g.NB32 * KB * 64;   // piece strides in uint4
  const uint4* Aw = g.A + ((long)(wave * FPW) * g.TB32 + 2 * tb) * KB * 64 + lane;
  const uint4* Bw = g.B + ((long)(wave * FPW) * g.NB32 + nb) * KB * 64 + lane;
  const long fA = (long)g.TB32 * KB * 64, fB = (long)g.NB32 * KB * 64;             // frequency strides

  auto load = [&](Stage& s, int fi, int kb) {
    const uint4* a = Aw + fi * fA + kb * 64;
    const uint4* b = Bw + fi * fB + kb * 64;
    s.a[0][0] = a[0];
    s.a[0][1] = a[KB * 64];
    s.a[1][0] = a[pA];
    s.a[1][1] = a[pA + KB * 64];
    s.b[0] = b[0];
    s.b[1] = b[pB];
  };

  f32x16 acc[FPW][2];
#pragma unroll
  for (int i = 0; i < FPW; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  Stage st[3];
  load(st[0], 0, 0);
  load(st[1], 1, 0);
  for (int kb = 0; kb < g.KB; ++kb) {
#pragma unroll
    for (int fi = 0; fi < FPW; ++fi) {
      {
        const int nf = (fi + 2) % FPW;
        int nkb = kb + ((fi + 2) >= FPW ? 1 : 0);
        nkb = nkb < g.KB ? nkb : g.KB - 1;         // (the last two prefetches re-read a loaded chunk)
        load(st[(fi + 2) % 3], nf, nkb);
      }
      const Stage& s = st[fi % 3];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        acc[fi][r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(s.a[0][r]), as_h8(s.b[0]), acc[fi][r], 0, 0, 0);
        acc[fi][r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(s.a[0][r]), as_h8(s.b[1]), acc[fi][r], 0, 0, 0);
        acc[fi][r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(s.a[1][r]), as_h8(s.b[0]), acc[fi][r], 0, 0, 0);
      }
    }
  }

#if EPI
  // The 36 planes of the 64 x 32 block meet in LDS one row block (32 tiles) and one 8-row group of the accumulator layout
  // at a time: plane f, tile row t (0..31), channel c (0..31) -> lds[f][t_sub][c].  An accumulator register q of lane l
  // is element (row 8 (q / 4) + 4 (l / 32) + q % 4, column l % 32).  Per pass p = (row block r, q-group qg = q / 4): every
  // wave stores 9 planes x 8 rows x 32 channels (4 registers per lane), then 256 threads = 8 rows x 32 channels transform
  // one (tile, channel) each: 36 LDS reads, 16 stores.
  __shared__ float lds[NF][8][33];
  if (FPW < 9) for (int i = threadIdx.x; i < (NF - 4 * FPW) * 8 * 33; i += 256) (&lds[4 * FPW][0][0])[i] = 0.f;
  const int c = lane & 31, hi = lane >> 5;
  // output transform matrix A^T (4 x 6) for the points {0, 1, -1, 1/2, -2, inf} (winograd.hip)
  const float AT[4][6] = {{1.f, 1.f, 1.f, 1.f, 1.f, 0.f},
                          {0.f, 1.f, -1.f, 0.5f, -2.f, 0.f},
                          {0.f, 1.f, 1.f, 0.25f, 4.f, 0.f},
                          {0.f, 1.f, -1.f, 0.125f, -8.f, 1.f}};
  const int tr = threadIdx.x >> 5, tc = threadIdx.x & 31;     // this thread's (row of the pass, channel)
#pragma unroll
  for (int r = 0; r < 2; ++r) {
#pragma unroll
    for (int qg = 0; qg < 4; ++qg) {
      __syncthreads();
#pragma unroll
      for (int fi = 0; fi < FPW; ++fi)
#pragma unroll
        for (int q = 0; q < 4; ++q) lds[wave * FPW + fi][4 * hi + q][c] = acc[fi][r][4 * qg + q];
      __syncthreads();
      float m[6][6];
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) m[i][j] = lds[i * 6 + j][tr][tc];
      float z[4][6];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          float s = 0.f;
#pragma unroll
          for (int i = 0; i < 6; ++i) s += AT[a][i] * m[i][j];
          z[a][j] = s;
        }
      const long tile = (long)tb * 64 + r * 32 + qg * 8 + tr;
      float* yp = g.Y + (tile * 16) * ((long)g.NB32 * 32) + nb * 32 + tc;
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          float s = 0.f;
#pragma unroll
          for (int j = 0; j < 6; ++j) s += z[a][j] * AT[b][j];
          yp[(long)(a * 4 + b) * g.NB32 * 32] = s;
        }
    }
  }
#else
  // no epilogue: sum the planes (keeps every accumulator live) and store one tile pair per wave
  f32x16 s0 = acc[0][0], s1 = acc[0][1];
#pragma unroll
  for (int fi = 1; fi < FPW; ++fi) {
    s0 += acc[fi][0] * (float)(fi + 1);
    s1 += acc[fi][1] * (float)(fi + 1);
  }
  float* o = g.Y + (((long)(tb * nnb + nb) * 4 + wave) * 2 * 64 + lane) * 16;
#pragma unroll
  for (int q = 0; q < 16; q += 4) {
    *reinterpret_cast<f32x4*>(o + q) = f32x4{s0[q], s0[q + 1], s0[q + 2], s0[q + 3]};
    *reinterpret_cast<f32x4*>(o + 64 * 16 + q) = f32x4{s1[q], s1[q + 1], s1[q + 2], s1[q + 3]};
  }
#endif
}

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      printf("%s failed: %s\n", #x, hipGetErrorString(e_));                   \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

static inline unsigned short f2h(float f) {
  _Float16 h = (_Float16)f;
  unsigned short u;
  __builtin_memcpy(&u, &h, 2);
  return u;
}
static inline float h2f(unsigned short u) {
  _Float16 h;
  __builtin_memcpy(&h, &u, 2);
  return (float)h;
}

int main(int argc, char** argv) {
  int T = 4096, N = 1024, K = 256;
  if (argc >= 4) { T = atoi(argv[1]); N = atoi(argv[2]); K = atoi(argv[3]); }
  const int TB32 = T / 32, NB32 = N / 32, KB = K / 16;
  const size_t nA = (size_t)2 * NF * T * K, nB = (size_t)2 * NF * N * K;     // halves
  std::vector<unsigned short> hA(nA), hB(nB);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.f - 0.5f; };
  // hi piece ~ U(-0.5, 0.5), lo piece 2^-11 of that: the magnitudes of a real split
  for (size_t i = 0; i < nA; ++i) hA[i] = f2h(rnd() * (i < nA / 2 ? 1.f : 1.f / 2048));
  for (size_t i = 0; i < nB; ++i) hB[i] = f2h(rnd() * (i < nB / 2 ? 1.f : 1.f / 2048));
  uint4 *dA, *dB;
  float* dY;
#if EPI
  const size_t nY = (size_t)T * 16 * N;
#else
  const size_t nY = (size_t)(T / 64) * NB32 * 4 * 2 * 64 * 16;
#endif
  CK(hipMalloc(&dA, nA * 2));
  CK(hipMalloc(&dB, nB * 2));
  CK(hipMalloc(&dY, nY * 4));
  CK(hipMemcpy(dA, hA.data(), nA * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, hB.data(), nB * 2, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const double flop = 2.0 * (4 * FPW) * T * (double)N * K;     // (of the frequencies computed)
  for (int map = 0; map < 2; ++map) {
    Args g{dA, dB, dY, TB32, NB32, KB, map};
    const int ntb = T / 64;
    int grid = ntb * NB32;
    if (map == 1) {
      if (NB32 % 8) continue;
      const int sbt = (ntb + 3) / 4, sbn = NB32 / 8;
      const int nsb = ((sbt * sbn + 7) / 8) * 8;
      grid = nsb * 32;
    }
    CK(hipMemset(dY, 0, nY * 4));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(fused_kernel, dim3(grid), dim3(256), 0, 0, g);
    CK(hipDeviceSynchronize());
    float best = 1e9f, tot = 0.f;
    const int reps = 20;
    for (int i = 0; i < reps; ++i) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(fused_kernel, dim3(grid), dim3(256), 0, 0, g);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      best = ms < best ? ms : best;
      tot += ms;
    }
    printf("fused_out_proto EPI=%d FPW=%d T=%d N=%d K=%d map=%d grid=%d: best %.1f us, mean %.1f us; products %.0f TFLOP/s, executed fp16 "
           "%.0f TFLOP/s = %.3f of 2500\n", EPI, FPW, T, N, K, map, grid, best * 1e3, tot / reps * 1e3, flop / (tot / reps * 1e-3) / 1e12,
           3 * flop / (tot / reps * 1e-3) / 1e12, 3 * flop / (tot / reps * 1e-3) / 1e12 / 2500.0);
  }
#if EPI
  // check tile 5 / channel 7 and tile T-3 / channel N-2 against a host evaluation of the same sums
  std::vector<float> hY(nY);
  CK(hipMemcpy(hY.data(), dY, nY * 4, hipMemcpyDeviceToHost));
  const float AT[4][6] = {{1, 1, 1, 1, 1, 0}, {0, 1, -1, 0.5f, -2, 0}, {0, 1, 1, 0.25f, 4, 0}, {0, 1, -1, 0.125f, -8, 1}};
  auto elem = [&](const std::vector<unsigned short>& v, int rows32, int p, int f, int row, int k) {
    const size_t chunk = (((size_t)p * NF + f) * rows32 + row / 32) * KB + k / 16;
    const int l = row % 32 + 32 * ((k % 16) / 8);
    return h2f(v[chunk * 512 + l * 8 + k % 8]);
  };
  double worst = 0;
  const int probes[2][2] = {{5, 7}, {T - 3, N - 2}};
  for (auto& pr : probes) {
    const int t = pr[0], c = pr[1];
    double m[6][6];
    for (int f = 0; f < NF; ++f) {
      double acc = 0;
      for (int k = 0; k < (f < 4 * FPW ? K : 0); ++k) {
        const double ah = elem(hA, TB32, 0, f, t, k), al = elem(hA, TB32, 1, f, t, k);
        const double bh = elem(hB, NB32, 0, f, c, k), bl = elem(hB, NB32, 1, f, c, k);
        acc += ah * bh + ah * bl + al * bh;
      }
      m[f / 6][f % 6] = acc;
    }
    for (int a = 0; a < 4; ++a)
      for (int b = 0; b < 4; ++b) {
        double y = 0;
        for (int i = 0; i < 6; ++i)
          for (int j = 0; j < 6; ++j) y += AT[a][i] * m[i][j] * AT[b][j];
        const double got = hY[((size_t)t * 16 + a * 4 + b) * N + c];
        worst = fmax(worst, fabs(got - y) / (fabs(y) + 1e-3));
      }
  }
  printf("check: worst relative difference of two probed (tile, channel) outputs %.2e %s\n", worst, worst < 1e-3 ? "OK" : "MISMATCH");
#endif
  return 0;
}
