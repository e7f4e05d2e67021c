// x3_phase.hip -- where does a workgroup of wino_bgemm_x3_kernel spend its time?  (dev tool, GPU box)
// Builds the production kernel from ot-gan_amd/csrc/gemm_x3.h with -DX3_TIMING (s_memtime stamps per phase kept in
// spare LDS) and runs it on the Winograd-domain GEMM shapes of the DCGAN layers with random operands.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DX3_TIMING -DX3_STREAM_TOOL -I ot-gan_amd/csrc -I tools/ablate tools/ablate/x3_phase.hip -o tools/ablate/bin/x3_phase
//   tools/ablate/bin/x3_phase M N K [reps]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>

#include <algorithm>
#include <vector>

#include "gemm_x3.h"
#include "gemm_x3_stream.h"     // the stream-K kernel (left the library in round 5); build with -DX3_STREAM_TOOL

void otgan_set_error(const char*, ...) {}
void otgan_prof_begin(int, double, double, hipStream_t) {}
void otgan_prof_end(int, hipStream_t) {}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

static unsigned build_fmap_plain(BgArgs& b) {
  int cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  memset(b.fmap, -1, sizeof(b.fmap));
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

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 4096, N = argc > 2 ? atoi(argv[2]) : 1024, K = argc > 3 ? atoi(argv[3]) : 256;
  const int reps = argc > 4 ? atoi(argv[4]) : 5;
  BgArgs b;
  memset(&b, 0, sizeof(b));
  b.M = M; b.N = N; b.K = K;
  b.ldc = N; b.sC = (long)M * N;
  b.tiles_m = (M + X3_BM - 1) / X3_BM;
  b.tiles_n = (N + X3_BN - 1) / X3_BN;
  b.kt_per_split = K / X3_BK;
  b.sAp = op_fstride(M, K); b.sBp = op_fstride(N, K);
  b.pA = kWF * b.sAp; b.pB = kWF * b.sBp;
  b.rbA = (M + 31) / 32; b.rbB = (N + 31) / 32; b.kblocks = K / 16;
  const size_t na = (size_t)X3_NP * b.pA, nb = (size_t)X3_NP * b.pB, nc = (size_t)kWF * M * N;
  std::vector<u16> h(std::max(na, nb));
  unsigned lcg = 12345u;
  const bool zeros = getenv("X3_ZEROS") != nullptr;   // all-zero operands: the same instruction stream at low power
  for (auto& v : h) {
    lcg = lcg * 1664525u + 1013904223u;
    // finite, all mantissa bits toggling: bf16 +-[0.0078, 0.0156); fp16 (two-piece build) +-[0.125, 1)
    v = zeros ? (u16)0 : X3_NP == 3 ? (u16)(0x3c00u | ((lcg >> 16) & 0x83ffu)) : (u16)((0x3000u + ((lcg >> 16) & 0x0bffu)) | ((lcg >> 16) & 0x8000u));
  }
  u16 *dA, *dB;
  float* dC;
  unsigned long long* dbg;
  CK(hipMalloc(&dA, na * 2)); CK(hipMalloc(&dB, nb * 2)); CK(hipMalloc(&dC, nc * 4));
  CK(hipMemcpy(dA, h.data(), na * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, h.data(), nb * 2, hipMemcpyHostToDevice));
  b.Ap = dA; b.Bp = dB; b.C = dC;
  const unsigned grid = build_fmap_plain(b);
  CK(hipMalloc(&dbg, (size_t)grid * 64 * 8));
  CK(hipMemset(dbg, 0, (size_t)grid * 64 * 8));
  b.dbg = dbg;
  const size_t lds = X3_LDS + 512;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(wino_bgemm_x3_kernel<true, false>),
                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f, sum = 0.f;
  for (int r = 0; r < reps + 2; ++r) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((wino_bgemm_x3_kernel<true, false>), dim3(grid, 1, 1), dim3(X3_THREADS), lds, 0, b);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (r >= 2) { best = std::min(best, ms); sum += ms; }
  }
  const double flop = (double)X3_NTERM * 2.0 * kWF * (double)M * N * K;
  printf("M=%d N=%d K=%d grid=%u tiles/XCD=%.1f: avg %.1f us best %.1f us  %.0f TFLOP/s (%.3f of 2500)\n", M, N, K, grid,
         b.tiles_m * b.tiles_n * 4.5, sum / reps * 1e3, best * 1e3, flop / (sum / reps * 1e-3) / 1e12,
         flop / (sum / reps * 1e-3) / 2.5e15);
  // ---- the persistent stream kernel on the same problem: time and compare ----
  {
    const int nw = argc > 5 ? atoi(argv[5]) : 32;
    std::vector<float> ref(nc), got(nc);
    CK(hipMemcpy(ref.data(), dC, nc * 4, hipMemcpyDeviceToHost));
    CK(hipMemset(dC, 0xff, nc * 4));
    float* skws;
    CK(hipMalloc(&skws, x3_stream_ws_floats(8 * nw) * 4));
    CK(hipMemset(skws, 0x5a, x3_stream_ws_floats(8 * nw) * 4));
    BgArgs c = b;
    c.sk_partial = skws;
    const int stagger = argc > 6 ? atoi(argv[6]) : 0;
    if (!x3_plan_stream(c, nw, stagger)) { printf("stream plan refused\n"); return 1; }
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(wino_bgemm_x3_stream_kernel<false>),
                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)X3_SK_LDS));
    unsigned long long* sdbg;
    CK(hipMalloc(&sdbg, (size_t)8 * nw * 64 * 8));
    CK(hipMemset(sdbg, 0, (size_t)8 * nw * 64 * 8));
    c.dbg = sdbg;
    float sbest = 1e9f, ssum = 0.f;
    for (int r = 0; r < reps + 2; ++r) {
      c.sk_epoch = 16 * (1000 + r);
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL((wino_bgemm_x3_stream_kernel<false>), dim3(8 * nw), dim3(X3_THREADS), X3_SK_LDS, 0, c);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (r >= 2) { sbest = std::min(sbest, ms); ssum += ms; }
    }
    CK(hipMemcpy(got.data(), dC, nc * 4, hipMemcpyDeviceToHost));
    double maxd = 0, maxr = 0;
    size_t bad = 0;
    for (size_t i = 0; i < nc; ++i) {
      const double d = fabs((double)got[i] - ref[i]);
      if (!(d <= 1e-3 * (fabs(ref[i]) + 1e-3))) ++bad;
      maxd = std::max(maxd, d);
      maxr = std::max(maxr, (double)fabs(ref[i]));
    }
    int splits = 0;
    for (int x = 0; x < 8; ++x)
      for (int w = 1; w < nw; ++w) splits += (c.sk_bound[x][w] & 0xffffu) ? 1 : 0;
    printf("stream (%d workgroups, %d cuts inside tiles): avg %.1f us best %.1f us  %.0f TFLOP/s (%.3f)  max|diff| %.3g of max|C| %.3g, %zu bad\n",
           8 * nw, splits, ssum / reps * 1e3, sbest * 1e3, flop / (ssum / reps * 1e-3) / 1e12, flop / (ssum / reps * 1e-3) / 2.5e15,
           maxd, maxr, bad);
    // stamps of workgroup 8 * (nw / 2) + 3: [0] start, per piece: fragments ready, loop done, write-out issued, stores <= 63
    std::vector<unsigned long long> st((size_t)8 * nw * 64);
    CK(hipMemcpy(st.data(), sdbg, st.size() * 8, hipMemcpyDeviceToHost));
    {
      unsigned long long t0 = ~0ull, t1 = 0;
      for (int w = 0; w < 8 * nw; ++w) {
        if (!st[(size_t)w * 64]) continue;
        t0 = std::min(t0, st[(size_t)w * 64]);
        t1 = std::max(t1, st[(size_t)w * 64 + 1]);
      }
      printf("  stream kernel span %llu cycles (last launch)\n", t1 - t0);
    }
    for (int wsel : {3, 8 * (nw / 2) + 3, 8 * (nw - 1) + 3}) {
      const unsigned long long* w = &st[(size_t)wsel * 64];
      printf("  workgroup %d: first fragments after %llu;", wsel, w[2] - w[0]);
      for (int i = 2; i + 3 < 64 && w[i]; i += 4)
        printf(" [loop %llu out %llu drain %llu next %llu]", w[i + 1] - w[i], w[i + 2] - w[i + 1], w[i + 3] - w[i + 2], w[i + 4] ? w[i + 4] - w[i + 3] : 0ull);
      printf(" total %llu\n", w[1] - w[0]);
    }
  }
#if X3_PIECES == 2
  // ---- the 256 x 128 tile kernel (two workgroups per compute unit) on the same problem ----
  {
    std::vector<float> ref(nc), got(nc);
    hipLaunchKernelGGL((wino_bgemm_x3_kernel<true, false>), dim3(grid, 1, 1), dim3(X3_THREADS), lds, 0, b);   // exact reference
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(ref.data(), dC, nc * 4, hipMemcpyDeviceToHost));
    CK(hipMemset(dC, 0xff, nc * 4));
    BgArgs c = b;
    c.tiles_n = (N + X3N_BN - 1) / X3N_BN;
    const unsigned ngrid = build_fmap_plain(c);
    unsigned long long* ndbg;
    CK(hipMalloc(&ndbg, (size_t)ngrid * 64 * 8));
    CK(hipMemset(ndbg, 0, (size_t)ngrid * 64 * 8));
    c.dbg = ndbg;
    c.x_total = ngrid;        // (round 4: the kernel walks queue positions up to x_total)
    const size_t nlds = X3N_LDS + 512;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(wino_bgemm_x3n_kernel<false>),
                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)nlds));
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, wino_bgemm_x3n_kernel<false>, X3_THREADS, nlds));
    float nbest = 1e9f, nsum = 0.f;
    for (int r = 0; r < reps + 2; ++r) {
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL((wino_bgemm_x3n_kernel<false>), dim3(ngrid, 1, 1), dim3(X3_THREADS), nlds, 0, c);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (r >= 2) { nbest = std::min(nbest, ms); nsum += ms; }
    }
    CK(hipMemcpy(got.data(), dC, nc * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    double maxd = 0;
    for (size_t i = 0; i < nc; ++i) {
      const double d = fabs((double)got[i] - ref[i]);
      if (!(d == 0)) {
        if (bad < 12 || (bad % 100003) == 0) {
          const size_t fm = i / N;
          printf("    differ at f=%zu m=%zu n=%zu: got %g ref %g\n", fm / M, fm % M, i % N, got[i], ref[i]);
        }
        ++bad;
      }
      maxd = std::max(maxd, d);
    }
    printf("narrow 256x128 (grid %u, %d workgroups/CU): avg %.1f us best %.1f us  %.0f TFLOP/s (%.3f)  max|diff| %.3g, %zu differ\n",
           ngrid, occ, nsum / reps * 1e3, nbest * 1e3, flop / (nsum / reps * 1e-3) / 1e12, flop / (nsum / reps * 1e-3) / 2.5e15, maxd, bad);
    std::vector<unsigned long long> t((size_t)ngrid * 64);
    CK(hipMemcpy(t.data(), ndbg, t.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> pro, loop, epi, total;
    for (unsigned x = 0; x < ngrid; ++x) {
      const unsigned long long* w = &t[(size_t)x * 64];
      if (!w[0] || !w[63]) continue;
      pro.push_back((double)(w[1] - w[0])); loop.push_back((double)(w[2] - w[1])); epi.push_back((double)(w[3] - w[2]));
      total.push_back((double)(w[63] - w[0]));
    }
    auto stat = [](std::vector<double> v, const char* name) {
      std::sort(v.begin(), v.end());
      double s = 0;
      for (double x : v) s += x;
      printf("  narrow %-10s mean %8.0f  p10 %8.0f  p50 %8.0f  p90 %8.0f  max %8.0f\n", name, s / v.size(), v[v.size() / 10],
             v[v.size() / 2], v[v.size() * 9 / 10], v.back());
    };
    stat(pro, "prologue"); stat(loop, "main loop"); stat(epi, "epi issue"); stat(total, "total");
    CK(hipMemcpy(dC, ref.data(), nc * 4, hipMemcpyHostToDevice));
  }
#endif
  std::vector<unsigned long long> t((size_t)grid * 64);
  CK(hipMemcpy(t.data(), dbg, t.size() * 8, hipMemcpyDeviceToHost));
  // per workgroup: prologue (entry -> first fragments), main loop, epilogue issue, store drain; cycles of s_memtime
  // (100 MHz constant clock on gfx9: 10 ns units)
  unsigned long long tmin = ~0ull, tmax = 0;
  std::vector<double> pro, loop, epi, drain, total, start;
  for (unsigned x = 0; x < grid; ++x) {
    const unsigned long long* w = &t[(size_t)x * 64];
    if (!w[0] || !w[63]) continue;
    tmin = std::min(tmin, w[0]);
    tmax = std::max(tmax, w[63]);
  }
  for (unsigned x = 0; x < grid; ++x) {
    const unsigned long long* w = &t[(size_t)x * 64];
    if (!w[0] || !w[63]) continue;
    pro.push_back((double)(w[1] - w[0]));
    loop.push_back((double)(w[2] - w[1]));
    epi.push_back((double)(w[3] - w[2]));
    drain.push_back((double)(w[63] - w[3]));
    total.push_back((double)(w[63] - w[0]));
    start.push_back((double)(w[0] - tmin));
  }
  auto stat = [](std::vector<double> v, const char* name) {
    std::sort(v.begin(), v.end());
    double s = 0;
    for (double x : v) s += x;
    printf("  %-10s mean %8.0f  p10 %8.0f  p50 %8.0f  p90 %8.0f  max %8.0f\n", name, s / v.size(), v[v.size() / 10],
           v[v.size() / 2], v[v.size() * 9 / 10], v.back());
  };
  printf("one-tile kernel: workgroups that ran: %zu; kernel span %.0f cycles (last launch)\n", total.size(), (double)(tmax - tmin));
  stat(pro, "prologue"); stat(loop, "main loop"); stat(epi, "epi issue"); stat(drain, "drain"); stat(total, "total");
  stat(start, "start at");
  // stage-pair durations of one mid-launch workgroup
  for (unsigned x = grid / 2; x < grid; ++x) {
    const unsigned long long* w = &t[(size_t)x * 64];
    if (!w[0] || !w[63]) continue;
    printf("workgroup %u stage pairs:", x);
    unsigned long long prev = w[1];
    for (int i = 4; i < 60 && w[i]; ++i) { printf(" %llu", w[i] - prev); prev = w[i]; }
    printf("\n");
    break;
  }
  return 0;
}
