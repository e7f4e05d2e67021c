// common.h -- shared device/host helpers for the OT-GAN gfx950 kernels.
// gfx950 (MI355X / CDNA4) only: wave = 64 lanes, fp32-input MFMA, 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define OTGAN_OK 0
#define OTGAN_ERR_INVALID (-1)
#define OTGAN_ERR_WORKSPACE (-2)
#define OTGAN_ERR_LAUNCH (-3)
#define OTGAN_ERR_UNSUPPORTED (-4)

// Thread-local last-error text (returned by otgan_last_error()).
void otgan_set_error(const char* fmt, ...);

#define OTGAN_CHECK_ARG(cond, ...)            \
  do {                                        \
    if (!(cond)) {                            \
      otgan_set_error(__VA_ARGS__);           \
      return OTGAN_ERR_INVALID;               \
    }                                         \
  } while (0)

#define OTGAN_CHECK_LAUNCH(what)                                              \
  do {                                                                        \
    hipError_t e__ = hipGetLastError();                                       \
    if (e__ != hipSuccess) {                                                  \
      otgan_set_error("%s: launch failed: %s", what, hipGetErrorString(e__)); \
      return OTGAN_ERR_LAUNCH;                                                \
    }                                                                         \
  } while (0)

// ---- optional per-kernel-class HIP-event timing (used by bench.py's roofline leg) -----------
// Classes are small integers; each timed launch records a start/stop event pair on the
// launch stream.  Disabled (zero overhead beyond one branch) unless otgan_prof_enable(1).
enum OtganProfClass {
  OTGAN_PROF_CONV_FWD = 0,
  OTGAN_PROF_CONV_DGRAD = 1,
  OTGAN_PROF_CONV_WGRAD = 2,
  OTGAN_PROF_COST_GEMM = 3,
  OTGAN_PROF_SINKHORN = 4,
  OTGAN_PROF_PLAN_APPLY = 5,
  OTGAN_PROF_POINTWISE = 6,
  OTGAN_PROF_WINO_GEMM = 7,     // the batched Winograd-domain GEMM alone, fp32 MFMA engine (nested inside the conv classes)
  OTGAN_PROF_WINO_GEMM_X3 = 8,  // the same on the bf16 pipe with split-precision operands; FLOP = executed bf16 FLOP (6 per fp32 product)
  OTGAN_PROF_NCLASS = 9
};
void otgan_prof_begin(int cls, double flops, double bytes, hipStream_t s);
void otgan_prof_end(int cls, hipStream_t s);

struct ProfScope {
  int cls;
  hipStream_t s;
  ProfScope(int c, double flops, double bytes, hipStream_t st) : cls(c), s(st) {
    otgan_prof_begin(c, flops, bytes, st);
  }
  ~ProfScope() { otgan_prof_end(cls, s); }
};

#ifdef __HIPCC__
#define OTGAN_HD __host__ __device__
#else
#define OTGAN_HD
#endif
OTGAN_HD static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
OTGAN_HD static inline long ceil_div_l(long a, long b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

#ifdef __HIPCC__
// ---- wave (64 lanes) reductions ---------------------------------------------------------
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// ---- amax records filled by the kernel that WRITES a tensor (otgan_layers.h: "amax records") -------------
// A record is OTGAN_AMAX_RECORD_FLOATS = 512 floats: 16 sub-slots at a stride of 32 floats (one cache line each); its
// value is the largest of the 16 entries, each the bit pattern of a non-negative float: for those unsigned order =
// numeric order, infinity sorts above every finite value and NaN above infinity, so a NaN anywhere makes the record
// NaN -- what absmax_kernel reports.  max is order-free: deterministic.  A producing kernel reduces per workgroup
// (shuffles, LDS) and max-accumulates into sub-slot blockIdx.x % 16 with ONE atomic -- and only when it would raise the
// entry (the k-th workgroup does with probability ~1/k).  Same-address atomics serialise in their L2 channel at a few
// ns each: one atomic per WAVE into one word cost a 16 k-wave GLU launch 50 us (32 -> 81 us); per workgroup over 16
// lines it is not measurable.  The caller zeroes the record before the producing launch.
constexpr int kAmaxSub = 16, kAmaxSubStride = 32;
__device__ __forceinline__ unsigned amax_bits(float v) { return __float_as_uint(v) & 0x7fffffffu; }
__device__ __forceinline__ unsigned amax_bits4(f32x4 v, unsigned mb) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const unsigned b = amax_bits(v[k]);
    mb = b > mb ? b : mb;
  }
  return mb;
}
// every thread of the workgroup that is still alive calls this (one-dimensional workgroups of <= 1024 threads; a lane
// that has exited reads as 0 in the shuffles: ds_bpermute of a disabled lane).  Waves that have exited altogether take no
// part: the workgroup's maximum and the committing wave are found among the waves that arrive (round 4 -- until then
// thread 0 read one LDS word per wave of the LAUNCH, stale LDS for a wave that had returned early: a last, partly
// filled workgroup could raise the record by garbage; the shapes of the models fill their workgroups, the 4 x 4 layer of
// a three-image batch did not).
__device__ __forceinline__ void amax_commit(float* rec, unsigned mb) {
  __shared__ unsigned s_amax[2];   // [0] maximum, [1] lowest wave that arrived
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned w = (unsigned)__shfl_xor((int)mb, o, 64);
    mb = w > mb ? w : mb;
  }
  const unsigned wave = threadIdx.x >> 6;
  const bool leader = (int)(threadIdx.x & 63) == __ffsll((long long)__ballot(1)) - 1;
  if (leader) { s_amax[0] = 0u; s_amax[1] = 0xffffffffu; }     // (every arriving wave writes the same two values)
  __syncthreads();
  if (leader) {
    atomicMax(&s_amax[0], mb);
    atomicMin(&s_amax[1], wave);
  }
  __syncthreads();
  if (leader && s_amax[1] == wave) {
    const unsigned m = s_amax[0];
    if (m != 0u) {
      unsigned* p = reinterpret_cast<unsigned*>(rec) + kAmaxSubStride * ((blockIdx.x + blockIdx.y + blockIdx.z) & (kAmaxSub - 1));
      if (m > __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        __hip_atomic_fetch_max(p, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
// value of a record (any thread; plain loads: the record was written by an earlier launch)
__device__ __forceinline__ float amax_record_value(const float* rec) {
  unsigned m = 0u;
#pragma unroll
  for (int k = 0; k < kAmaxSub; ++k) {
    const unsigned b = __float_as_uint(rec[k * kAmaxSubStride]) & 0x7fffffffu;
    m = b > m ? b : m;
  }
  return __uint_as_float(m);
}
// maximum over `count` consecutive records (kAmaxSub * kAmaxSubStride floats apart); bit patterns: a NaN stays a NaN
__device__ __forceinline__ float amax_records_value(const float* rec, int count) {
  unsigned m = 0u;
  for (int r = 0; r < count; ++r) {
    const unsigned b = __float_as_uint(amax_record_value(rec + (long)r * kAmaxSub * kAmaxSubStride));
    m = b > m ? b : m;
  }
  return __uint_as_float(m);
}
// exp(x) for x <= 0 (max-shifted) through the native 2^x unit.
__device__ __forceinline__ float exp_neg(float x) {
  return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f);
}
#endif
