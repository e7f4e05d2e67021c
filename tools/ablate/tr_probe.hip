// Probe of ds_read_b64_tr_b16 (gfx950): LDS holds u16 = element index; lane l reads at byte address addr[l].
// Prints, per lane, the four 16-bit values it receives.  build: hipcc -O2 --offload-arch=gfx950 tr_probe.hip -o tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__global__ void probe(const int* addr, unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const unsigned a = (unsigned)(unsigned long)(__attribute__((address_space(3))) void*)lds + addr[threadIdx.x];
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
  out[threadIdx.x * 4 + 0] = v[0] & 0xffff;
  out[threadIdx.x * 4 + 1] = v[0] >> 16;
  out[threadIdx.x * 4 + 2] = v[1] & 0xffff;
  out[threadIdx.x * 4 + 3] = v[1] >> 16;
}
int main() {
  int h[64];
  int* d; unsigned short* o; unsigned short ho[256];
  hipMalloc(&d, 256); hipMalloc(&o, 512);
  for (int variant = 0; variant < 3; ++variant) {
    // 0: lane l -> l * 8 bytes (contiguous);  1: lane l -> row (l % 16) * 64 bytes + (l / 16) * 8;  2: lane l -> row (l%4)*... 
    for (int l = 0; l < 64; ++l) {
      if (variant == 0) h[l] = l * 8;
      if (variant == 1) h[l] = (l % 16) * 64 + (l / 16) * 8;
      if (variant == 2) h[l] = (l % 4) * 128 + ((l / 4) % 4) * 8 + (l / 16) * 32;
    }
    hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o);
    hipMemcpy(ho, o, 512, hipMemcpyDeviceToHost);
    printf("variant %d (values = u16 element index = byte address / 2)\n", variant);
    for (int l = 0; l < 64; ++l)
      printf("lane %2d addr %4d(elem %4d): %4d %4d %4d %4d\n", l, h[l], h[l] / 2, ho[l * 4], ho[l * 4 + 1], ho[l * 4 + 2], ho[l * 4 + 3]);
  }
  return 0;
}
