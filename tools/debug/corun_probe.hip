// dev: instrumented victims for the co-residency miscompute (DESIGN section 3 "Four hazards", item 3).
// Built on its own (default compiler flags, i.e. packed fp32 ON) into tools/debug/bin/libcorun_probe.so and driven by
// tools/debug/corun_probe.py, which runs the REAL Winograd-domain GEMM of libotgan_hip.so on a second stream.
//
//   probe_rgbin(variant, ...)   the round-2 form of conv_rgbin_fwd_kernel (weights in registers, pixel broadcast from LDS,
//                               `acc = fma(w2, {x, x}, acc)` = v_pk_fma_f32 with op_sel), KS = 5, TR = 8, W = 32
//        variant 0: as it was                          -> does it still fail beside the GEMM?
//        variant 1: every product ALSO as two scalar v_fma_f32 from the SAME registers; packed and scalar accumulators
//                   compared per pixel pair in the kernel; on a mismatch the pixel is recomputed (LDS re-read) both ways;
//                   at the end the weight registers are compared with a fresh load and the LDS tile (plus canary words
//                   behind it) with its source -> tells a wrong INPUT (LDS / register corruption) from a wrong RESULT
//   probe_canary(...)           fills `lds_bytes` of LDS with an address pattern and re-reads it `iters` times with plain
//                               ds_read_b32: any foreign write into this workgroup's LDS (an LDS-DMA of the neighbour landing
//                               outside its own allocation) is caught with address and value
//   probe_reduce(...)           a neighbour shaped like a collective's reduction kernel: out = a + b staged through LDS,
//                               float4, default flags (packed adds)
// Records: dbg[0] = number of records (atomic), then 16 words per record.
#include <hip/hip_runtime.h>
#include <cstdint>

typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned hw_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
  return v;
}
__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v;
}
__device__ unsigned* dbg_record(unsigned* dbg, unsigned cap) {
  const unsigned i = atomicAdd(dbg, 1u);
  return i < cap ? dbg + 16 + 16 * (size_t)i : nullptr;
}
__device__ __forceinline__ unsigned fbits(float f) { return __float_as_uint(f); }

struct RgbArgs {
  const float* x;    // [N, H, W, 3]
  const float* wT;   // [128][75]
  float* y;          // [N, H, W, 128]
  unsigned* dbg;
  unsigned cap;
  int N, H, W;
};
constexpr int KS = 5, TR = 8, CANARY = 64;   // canary float4s behind the tile

template <int VARIANT>
__global__ __launch_bounds__(256) void probe_rgbin_kernel(RgbArgs a) {
  extern __shared__ __attribute__((aligned(16))) float4 s_in[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int co = lane;
  const int tiles_per_img = a.H / TR;
  const int n = blockIdx.x / tiles_per_img, r0 = (blockIdx.x - n * tiles_per_img) * TR;
  const int LW = a.W + KS - 1, LH = TR + KS - 1;
  const long img = (long)n * a.H * a.W;
  auto src_pixel = [&](int i) {
    const int lr = i / LW, lc = i - lr * LW;
    const int ih = r0 + lr - 2, iw = lc - 2;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W) {
      const float* px = a.x + (img + (long)ih * a.W + iw) * 3;
      v = make_float4(px[0], px[1], px[2], 0.f);
    }
    return v;
  };
  for (int i = tid; i < LH * LW; i += 256) s_in[i] = src_pixel(i);
  if (VARIANT == 1 && tid < CANARY) s_in[LH * LW + tid] = make_float4(__uint_as_float(0xC0DE0000u + tid), 1.f, 2.f, 3.f);
  f32x2 w[KS * KS * 3];
  const float* wlo = a.wT + (long)co * 75;
  const float* whi = wlo + 64L * 75;
#pragma unroll
  for (int i = 0; i < KS * KS * 3; ++i) {
    w[i] = f32x2{wlo[i], whi[i]};
    if (VARIANT == 1) asm volatile("" : "+v"(w[i]));   // one copy of every weight: the pair the packed FMAs read
  }
  __syncthreads();
  const int rows = TR >> 2, rbase = wave * rows, pairs = a.W >> 1;
  for (int pi = 0; pi < rows * pairs; ++pi) {
    const int r = rbase + pi / pairs, c0 = (pi % pairs) * 2;
    f32x2 acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f};
    float s00 = 0.f, s01 = 0.f, s10 = 0.f, s11 = 0.f;
#pragma unroll
    for (int kh = 0; kh < KS; ++kh) {
      float4 xs[KS + 1];
      const float4* src = s_in + (r + kh) * LW + c0;
#pragma unroll
      for (int j = 0; j < KS + 1; ++j) xs[j] = src[j];
#pragma unroll
      for (int kw = 0; kw < KS; ++kw) {
        const f32x2* wt = w + (kh * KS + kw) * 3;
        acc0 = __builtin_elementwise_fma(wt[0], f32x2{xs[kw].x, xs[kw].x}, acc0);
        acc0 = __builtin_elementwise_fma(wt[1], f32x2{xs[kw].y, xs[kw].y}, acc0);
        acc0 = __builtin_elementwise_fma(wt[2], f32x2{xs[kw].z, xs[kw].z}, acc0);
        acc1 = __builtin_elementwise_fma(wt[0], f32x2{xs[kw + 1].x, xs[kw + 1].x}, acc1);
        acc1 = __builtin_elementwise_fma(wt[1], f32x2{xs[kw + 1].y, xs[kw + 1].y}, acc1);
        acc1 = __builtin_elementwise_fma(wt[2], f32x2{xs[kw + 1].z, xs[kw + 1].z}, acc1);
        if (VARIANT == 1) {
#define SFMA(s, wv, xv) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s) : "v"(wv), "v"(xv))
          SFMA(s00, wt[0][0], xs[kw].x); SFMA(s01, wt[0][1], xs[kw].x);
          SFMA(s00, wt[1][0], xs[kw].y); SFMA(s01, wt[1][1], xs[kw].y);
          SFMA(s00, wt[2][0], xs[kw].z); SFMA(s01, wt[2][1], xs[kw].z);
          SFMA(s10, wt[0][0], xs[kw + 1].x); SFMA(s11, wt[0][1], xs[kw + 1].x);
          SFMA(s10, wt[1][0], xs[kw + 1].y); SFMA(s11, wt[1][1], xs[kw + 1].y);
          SFMA(s10, wt[2][0], xs[kw + 1].z); SFMA(s11, wt[2][1], xs[kw + 1].z);
        }
      }
    }
    if (VARIANT == 1) {
      const bool bad = fbits(acc0[0]) != fbits(s00) || fbits(acc0[1]) != fbits(s01) || fbits(acc1[0]) != fbits(s10) ||
                       fbits(acc1[1]) != fbits(s11);
      if (bad) {
        // once more, both ways, operands re-read from LDS
        f32x2 p0 = {0.f, 0.f};
        float q0 = 0.f, q1 = 0.f;
        for (int kh = 0; kh < KS; ++kh)
          for (int kw = 0; kw < KS; ++kw) {
            const float4 xv = s_in[(r + kh) * LW + c0 + kw];
            const float xq[3] = {xv.x, xv.y, xv.z};
#pragma unroll
            for (int j = 0; j < 3; ++j) {
              f32x2 ww;
              // (dynamic index into w[] would spill: reload the weight instead)
              ww = f32x2{wlo[(kh * KS + kw) * 3 + j], whi[(kh * KS + kw) * 3 + j]};
              p0 = __builtin_elementwise_fma(ww, f32x2{xq[j], xq[j]}, p0);
              SFMA(q0, ww[0], xq[j]); SFMA(q1, ww[1], xq[j]);
            }
          }
        unsigned* rec = dbg_record(a.dbg, a.cap);
        if (rec) {
          rec[0] = 1u; rec[1] = blockIdx.x; rec[2] = tid; rec[3] = pi; rec[4] = hw_id(); rec[5] = xcc_id();
          rec[6] = fbits(acc0[0]); rec[7] = fbits(s00); rec[8] = fbits(acc0[1]); rec[9] = fbits(s01);
          rec[10] = fbits(acc1[0]); rec[11] = fbits(s10); rec[12] = fbits(acc1[1]); rec[13] = fbits(s11);
          rec[14] = fbits(p0[0]); rec[15] = fbits(q0);
        }
      }
    }
    float* dst = a.y + (img + (long)(r0 + r) * a.W + c0) * 128 + co;
    dst[0] = acc0[0];
    dst[64] = acc0[1];
    dst[128] = acc1[0];
    dst[128 + 64] = acc1[1];
  }
  if (VARIANT == 1) {
    // weight registers against a fresh load
    unsigned wbad = 0, first = 0xffffffffu;
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < KS * KS * 3; ++i) {
      if (i % 4 == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (keeps the fresh loads from being hoisted / batched: registers)
      const bool b = fbits(w[i][0]) != fbits(wlo[i]) || fbits(w[i][1]) != fbits(whi[i]);
      if (b && first == 0xffffffffu) first = i;
      wbad += b;
    }
    if (wbad) {
      unsigned* rec = dbg_record(a.dbg, a.cap);
      if (rec) { rec[0] = 2u; rec[1] = blockIdx.x; rec[2] = tid; rec[3] = wbad; rec[4] = hw_id(); rec[5] = xcc_id(); rec[6] = first; }
    }
    __syncthreads();
    for (int i = tid; i < LH * LW + CANARY; i += 256) {
      const float4 got = s_in[i];
      const float4 want = i < LH * LW ? src_pixel(i) : make_float4(__uint_as_float(0xC0DE0000u + (i - LH * LW)), 1.f, 2.f, 3.f);
      if (fbits(got.x) != fbits(want.x) || fbits(got.y) != fbits(want.y) || fbits(got.z) != fbits(want.z) || fbits(got.w) != fbits(want.w)) {
        unsigned* rec = dbg_record(a.dbg, a.cap);
        if (rec) {
          rec[0] = 3u; rec[1] = blockIdx.x; rec[2] = tid; rec[3] = i; rec[4] = hw_id(); rec[5] = xcc_id();
          rec[6] = fbits(got.x); rec[7] = fbits(want.x); rec[8] = fbits(got.y); rec[9] = fbits(want.y);
          rec[10] = fbits(got.z); rec[11] = fbits(want.z); rec[12] = fbits(got.w); rec[13] = fbits(want.w);
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) void probe_canary_kernel(unsigned* dbg, unsigned cap, int words, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned s_c[];
  const int tid = threadIdx.x;
  const unsigned salt = 0x5A000000u ^ (blockIdx.x << 12);
  for (int i = tid; i < words; i += 256) s_c[i] = salt ^ i;
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
    for (int i = tid; i < words; i += 256) {
      const unsigned got = *(volatile unsigned*)(s_c + i);
      if (got != (salt ^ i)) {
        unsigned* rec = dbg_record(dbg, cap);
        if (rec) { rec[0] = 4u; rec[1] = blockIdx.x; rec[2] = tid; rec[3] = i; rec[4] = hw_id(); rec[5] = xcc_id(); rec[6] = got; rec[7] = salt ^ i; rec[8] = it; }
        s_c[i] = salt ^ i;
      }
    }
    __builtin_amdgcn_s_sleep(8);
  }
}

// out = a + b through an LDS stage, 1024 float4 per workgroup and round (what a collective's reduce kernel does)
__global__ __launch_bounds__(256) void probe_reduce_kernel(const float4* __restrict__ a, const float4* __restrict__ b,
                                                           float4* __restrict__ out, long n4) {
  __shared__ float4 st[2][1024];
  const int tid = threadIdx.x;
  for (long base = (long)blockIdx.x * 1024; base < n4; base += (long)gridDim.x * 1024) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long i = base + k * 256 + tid;
      if (i < n4) { st[0][k * 256 + tid] = a[i]; st[1][k * 256 + tid] = b[i]; }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int j = k * 256 + ((tid + 64) & 255);      // read what another wave staged
      const long i = base + j;
      if (i < n4) {
        const float4 u = st[0][j], v = st[1][j];
        out[i] = make_float4(fmaf(u.x, 1.f, v.x), fmaf(u.y, 1.f, v.y), fmaf(u.z, 1.f, v.z), fmaf(u.w, 1.f, v.w));
      }
    }
    __syncthreads();
  }
}

// Which instruction FORM goes wrong beside the GEMM?  16 accumulator pairs per lane, `iters` rounds of acc = w * x + acc,
// everything pinned in asm.  FORM 0: v_pk_fma_f32 with x broadcast through op_sel_hi, x from a broadcast ds_read_b128;
// 1: the same with x from registers only; 2: v_pk_fma_f32 without op_sel (src1 = {x, x} built by moves); 3: v_pk_mul_f32 +
// v_pk_add_f32; 4: v_fma_f64 (another instruction with 64-bit register-pair operands); 5: two scalar v_fma_f32 (control);
// 6 - 10: one op_sel pattern each (see the branches).
template <int FORM>
__global__ __launch_bounds__(256) void probe_forms_kernel(const float* __restrict__ wsrc, const float* __restrict__ xsrc, int iters,
                                                          float* __restrict__ out) {
  __shared__ float4 xs[64];
  const int tid = blockIdx.x * 256 + threadIdx.x;
  constexpr int NA = 16;
  f32x2 w[NA], acc[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    w[i] = f32x2{wsrc[(tid * 32 + 2 * i) & 8191], wsrc[(tid * 32 + 2 * i + 1) & 8191]};
    acc[i] = f32x2{0.f, 0.f};
    asm volatile("" : "+v"(w[i]), "+v"(acc[i]));
  }
  if (threadIdx.x < 64) xs[threadIdx.x] = make_float4(xsrc[threadIdx.x], xsrc[64 + threadIdx.x], xsrc[128 + threadIdx.x], xsrc[192 + threadIdx.x]);
  __syncthreads();
  float4 xreg = make_float4(xsrc[tid & 63], xsrc[64 + (tid & 63)], xsrc[128 + (tid & 63)], xsrc[192 + (tid & 63)]);
  for (int it = 0; it < iters; ++it) {
    float4 xv = xreg;
    if (FORM == 0) xv = xs[it & 63];
    f32x2 xlo = {xv.x, xv.y}, xhi = {xv.z, xv.w};
    asm volatile("" : "+v"(xlo), "+v"(xhi));
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      f32x2& xp = (i & 2) ? xhi : xlo;
      if (FORM == 0 || FORM == 1) {
        if (i & 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc[i]) : "v"(w[i]), "v"(xp));
        else asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[i]) : "v"(w[i]), "v"(xp));
      } else if (FORM == 6) {        // only the high-half cross-over: the HIGH result half reads src1's LOW register
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[i]) : "v"(w[i]), "v"(xp));
      } else if (FORM == 7) {        // only the low-half cross-over: the LOW result half reads src1's HIGH register
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc[i]) : "v"(w[i]), "v"(xp));
      } else if (FORM == 8) {        // the same cross-over in v_pk_mul_f32
        f32x2 t;
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(t) : "v"(w[i]), "v"(xp));
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(t));
      } else if (FORM == 9) {        // the cross-over on src0
        asm volatile("v_pk_fma_f32 %0, %2, %1, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc[i]) : "v"(w[i]), "v"(xp));
      } else if (FORM == 10) {       // the cross-over on src2 (the accumulator halves swap every round)
        f32x2 xx = (i & 1) ? f32x2{xp[1], xp[1]} : f32x2{xp[0], xp[0]};
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,1] op_sel_hi:[1,1,0]" : "+v"(acc[i]) : "v"(w[i]), "v"(xx));
      } else if (FORM == 2) {
        f32x2 xx = (i & 1) ? f32x2{xp[1], xp[1]} : f32x2{xp[0], xp[0]};
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(w[i]), "v"(xx));
      } else if (FORM == 3) {
        f32x2 xx = (i & 1) ? f32x2{xp[1], xp[1]} : f32x2{xp[0], xp[0]};
        f32x2 t;
        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(w[i]), "v"(xx));
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(t));
      } else if (FORM == 4) {
        double a = __builtin_bit_cast(double, acc[i]), ww = (double)w[i][0], x = (double)((i & 1) ? xp[1] : xp[0]);
        asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a) : "v"(ww), "v"(x));
        acc[i] = __builtin_bit_cast(f32x2, a);
      } else {
        const float x = (i & 1) ? xp[1] : xp[0];
        float a0 = acc[i][0], a1 = acc[i][1];
        asm volatile("v_fma_f32 %0, %2, %4, %0\n\tv_fma_f32 %1, %3, %4, %1" : "+v"(a0), "+v"(a1) : "v"(w[i][0]), "v"(w[i][1]), "v"(x));
        acc[i] = f32x2{a0, a1};
      }
    }
    if ((it & 15) == 15) {   // keep the values bounded
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        float a0 = acc[i][0], a1 = acc[i][1];
        asm volatile("v_mul_f32 %0, 0.125, %0\n\tv_mul_f32 %1, 0.125, %1" : "+v"(a0), "+v"(a1));
        acc[i] = f32x2{a0, a1};
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    out[(long)tid * 32 + 2 * i] = acc[i][0];
    out[(long)tid * 32 + 2 * i + 1] = acc[i][1];
  }
}

extern "C" {
int probe_forms(int form, const float* w, const float* x, int iters, float* out, int blocks, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  switch (form) {
    case 0: hipLaunchKernelGGL(probe_forms_kernel<0>, dim3(blocks), dim3(256), 0, s, w, x, iters, out); break;
    case 1: hipLaunchKernelGGL(probe_forms_kernel<1>, dim3(blocks), dim3(256), 0, s, w, x, iters, out); break;
    case 2: hipLaunchKernelGGL(probe_forms_kernel<2>, dim3(blocks), dim3(256), 0, s, w, x, iters, out); break;
    case 3: hipLaunchKernelGGL(probe_forms_kernel<3>, dim3(blocks), dim3(256), 0, s, w, x, iters, out); break;
    case 4: hipLaunchKernelGGL(probe_forms_kernel<4>, dim3(blocks), dim3(256), 0, s, w, x, iters, out); break;
    case 6: hipLaunchKernelGGL(probe_forms_kernel<6>, dim3(blocks), dim3(256), 0, s, w, x, iters, out); break;
    case 7: hipLaunchKernelGGL(probe_forms_kernel<7>, dim3(blocks), dim3(256), 0, s, w, x, iters, out); break;
    case 8: hipLaunchKernelGGL(probe_forms_kernel<8>, dim3(blocks), dim3(256), 0, s, w, x, iters, out); break;
    case 9: hipLaunchKernelGGL(probe_forms_kernel<9>, dim3(blocks), dim3(256), 0, s, w, x, iters, out); break;
    case 10: hipLaunchKernelGGL(probe_forms_kernel<10>, dim3(blocks), dim3(256), 0, s, w, x, iters, out); break;
    default: hipLaunchKernelGGL(probe_forms_kernel<5>, dim3(blocks), dim3(256), 0, s, w, x, iters, out); break;
  }
  return (int)hipGetLastError();
}
int probe_rgbin(int variant, const float* x, const float* wT, float* y, unsigned* dbg, unsigned cap, int N, int H, int W,
                void* stream) {
  RgbArgs a{x, wT, y, dbg, cap, N, H, W};
  const size_t lds = sizeof(float4) * ((size_t)(TR + KS - 1) * (W + KS - 1) + CANARY);
  const dim3 grid(N * (H / TR));
  if (variant == 0) hipLaunchKernelGGL(probe_rgbin_kernel<0>, grid, dim3(256), lds, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(probe_rgbin_kernel<1>, grid, dim3(256), lds, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}
int probe_canary(unsigned* dbg, unsigned cap, int lds_bytes, int iters, int blocks, void* stream) {
  static int set = 0;
  if (lds_bytes > set) {
    hipFuncSetAttribute((const void*)probe_canary_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    set = lds_bytes;
  }
  hipLaunchKernelGGL(probe_canary_kernel, dim3(blocks), dim3(256), lds_bytes, (hipStream_t)stream, dbg, cap, lds_bytes / 4, iters);
  return (int)hipGetLastError();
}
int probe_reduce(const float* a, const float* b, float* out, long n, int blocks, void* stream) {
  hipLaunchKernelGGL(probe_reduce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float4*)a, (const float4*)b,
                     (float4*)out, n / 4);
  return (int)hipGetLastError();
}
}
