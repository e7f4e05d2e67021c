// dev: minimal reproducer hunt for the co-run miscompute (tests/test_corun_gpu.py).  A victim kernel runs chains of
// v_pk_fma_f32 with one operand broadcast through op_sel (what `acc4 += w4 * x` compiles to) on one stream while an
// aggressor kernel keeps the compute units busy on another: matrix instructions only, LDS-DMA only, or both (the shape of
// wino_bgemm_x3n_kernel's main loop).  The victim's result is compared bit for bit with its own result on an idle GPU.
//   hipcc -O3 --offload-arch=gfx950 tools/debug/pkfma_corun.hip -o /tmp/pkfma_corun && /tmp/pkfma_corun
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// victim: 8 packed accumulators per lane, `iters` rounds of acc[i] = w[i] * {x, x} + acc[i]
template <bool PACKED, bool FROM_LDS>
__global__ __launch_bounds__(256) void victim_kernel(const float* __restrict__ wsrc, const float* __restrict__ xsrc, int iters,
                                                     float* __restrict__ out) {
  __shared__ float4 xs[64];
  const int tid = blockIdx.x * 256 + threadIdx.x;
  f32x2 w[8], acc[8];
  for (int i = 0; i < 8; ++i) {
    w[i] = f32x2{wsrc[(tid * 16 + 2 * i) & 4095], wsrc[(tid * 16 + 2 * i + 1) & 4095]};
    acc[i] = f32x2{0.f, 0.f};
  }
  if (threadIdx.x < 64) xs[threadIdx.x] = make_float4(xsrc[threadIdx.x], xsrc[64 + threadIdx.x], xsrc[128 + threadIdx.x], xsrc[192 + threadIdx.x]);
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
    float4 xv;
    if (FROM_LDS) xv = xs[it & 63];                              // broadcast ds_read_b128
    else xv = make_float4(xsrc[it & 63], xsrc[64 + (it & 63)], xsrc[128 + (it & 63)], xsrc[192 + (it & 63)]);
    const float xq[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float x = xq[i & 3];
      if (PACKED) {
        const f32x2 xx = {x, 12345.f};   // op_sel_hi:[1,0,1]: both halves take xx[0]
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[i]) : "v"(w[i]), "v"(xx));
      } else {
        float a0 = acc[i][0], a1 = acc[i][1];
        asm volatile("v_fma_f32 %0, %2, %4, %0\n\tv_fma_f32 %1, %3, %4, %1" : "+v"(a0), "+v"(a1) : "v"(w[i][0]), "v"(w[i][1]), "v"(x));
        acc[i] = f32x2{a0, a1};
      }
    }
  }
  for (int i = 0; i < 8; ++i) {
    out[(long)tid * 16 + 2 * i] = acc[i][0];
    out[(long)tid * 16 + 2 * i + 1] = acc[i][1];
  }
}

// aggressor: MODE bit 0 = matrix instructions, bit 1 = LDS-DMA refills (global_load_lds_dwordx4), bit 2 = ds_read_b128
template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void aggressor_kernel(const _Float16* __restrict__ src, int iters, float* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // 72 KiB: two workgroups per compute unit
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f32x4 acc[32];
  for (int i = 0; i < 32; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  f16x8 a = *reinterpret_cast<const f16x8*>(src + (threadIdx.x & 255) * 8);
  f16x8 b = *reinterpret_cast<const f16x8*>(src + 2048 + (threadIdx.x & 255) * 8);
  const unsigned ldsbase = (unsigned)(size_t)lds;
  for (int it = 0; it < iters; ++it) {
    if (MODE & 2) {
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        const _Float16* g = src + (((it * 6 + c) * 256 + wave * 64) & 8191) * 8;   // wave-uniform base, lanes 16 bytes apart
        const unsigned dst = ldsbase + (unsigned)(((it % 3) * 24 + wave * 6 + c) * 1024);
        const unsigned voff = (unsigned)lane * 16u;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(g), "s"(dst) : "memory");
      }
    }
    if (MODE & 4) {
      const f16x8 t = *reinterpret_cast<const f16x8*>(lds + ((it & 15) * 4096 + threadIdx.x * 16));
      a = a + t * (_Float16)0.0f;
    }
    if (MODE & 1) {
#pragma unroll
      for (int i = 0; i < 32; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
    }
    if (MODE & 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  f32x4 s = acc[0];
  for (int i = 1; i < 32; ++i) s += acc[i];
  if (s[0] == 123.456f) sink[blockIdx.x * 256 + threadIdx.x] = s[1] + s[2] + s[3] + (float)lane;
}

template <int MODE>
static void launch_aggressor(const _Float16* src, int iters, float* sink, hipStream_t s) {
  static bool once = (hipFuncSetAttribute((const void*)aggressor_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 73728), true);
  (void)once;
  hipLaunchKernelGGL(aggressor_kernel<MODE>, dim3(512), dim3(256), 73728, s, src, iters, sink);
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 40;
  const int vblocks = 2048, viters = 2000;
  float *w, *x, *out, *ref, *sink;
  _Float16* src;
  CHECK(hipMalloc(&w, 4096 * 4)); CHECK(hipMalloc(&x, 256 * 4));
  CHECK(hipMalloc(&out, (size_t)vblocks * 256 * 16 * 4)); CHECK(hipMalloc(&ref, (size_t)vblocks * 256 * 16 * 4));
  CHECK(hipMalloc(&sink, 512 * 256 * 4)); CHECK(hipMalloc(&src, 8192 * 8 * 2));
  std::vector<float> hw(4096), hx(256);
  srand(1);
  for (auto& v : hw) v = (rand() / (float)RAND_MAX - 0.5f) * 0.1f;
  for (auto& v : hx) v = rand() / (float)RAND_MAX - 0.5f;
  std::vector<_Float16> hs(8192 * 8);
  for (auto& v : hs) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 0.01f);
  CHECK(hipMemcpy(w, hw.data(), 4096 * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(x, hx.data(), 256 * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(src, hs.data(), hs.size() * 2, hipMemcpyHostToDevice));
  hipStream_t sa, sb;
  CHECK(hipStreamCreate(&sa)); CHECK(hipStreamCreate(&sb));
  const size_t nout = (size_t)vblocks * 256 * 16;
  std::vector<float> hout(nout), href(nout);
  const char* vnames[4] = {"packed, x from LDS", "packed, x from global", "scalar, x from LDS", "scalar, x from global"};
  const char* anames[5] = {"matrix only", "LDS-DMA only", "matrix + LDS-DMA", "matrix + LDS-DMA + ds_read", "idle"};
  for (int v = 0; v < 4; ++v) {
    auto run_victim = [&](float* dst, hipStream_t s) {
      if (v == 0) hipLaunchKernelGGL((victim_kernel<true, true>), dim3(vblocks), dim3(256), 0, s, w, x, viters, dst);
      if (v == 1) hipLaunchKernelGGL((victim_kernel<true, false>), dim3(vblocks), dim3(256), 0, s, w, x, viters, dst);
      if (v == 2) hipLaunchKernelGGL((victim_kernel<false, true>), dim3(vblocks), dim3(256), 0, s, w, x, viters, dst);
      if (v == 3) hipLaunchKernelGGL((victim_kernel<false, false>), dim3(vblocks), dim3(256), 0, s, w, x, viters, dst);
    };
    run_victim(ref, sb);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(href.data(), ref, nout * 4, hipMemcpyDeviceToHost));
    for (int ag = 0; ag < 5; ++ag) {
      int bad = 0;
      long wrong = 0, lane_hist[4] = {0, 0, 0, 0}, half_hist[2] = {0, 0};
      for (int r = 0; r < reps; ++r) {
        if (ag == 0) launch_aggressor<1>(src, 3000, sink, sa);
        if (ag == 1) launch_aggressor<2>(src, 3000, sink, sa);
        if (ag == 2) launch_aggressor<3>(src, 3000, sink, sa);
        if (ag == 3) launch_aggressor<7>(src, 3000, sink, sa);
        for (int k = 0; k < 4; ++k) run_victim(out, sb);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(hout.data(), out, nout * 4, hipMemcpyDeviceToHost));
        if (memcmp(hout.data(), href.data(), nout * 4) != 0) {
          ++bad;
          for (size_t i = 0; i < nout; ++i)
            if (memcmp(&hout[i], &href[i], 4) != 0) { ++wrong; ++lane_hist[((i / 16) & 63) >> 4]; ++half_hist[i & 1]; }
        }
      }
      printf("victim [%s] beside [%s]: %d of %d runs differ", vnames[v], anames[ag], bad, reps);
      if (bad) printf("  (%ld elements; by lane quarter %ld %ld %ld %ld; low / high half %ld %ld)", wrong, lane_hist[0], lane_hist[1], lane_hist[2], lane_hist[3], half_hist[0], half_hist[1]);
      printf("\n");
      fflush(stdout);
    }
  }
  return 0;
}
