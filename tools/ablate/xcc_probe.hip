#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(int* out) {
  // burn a little time so that the grid overlaps
  if (threadIdx.x == 0) {
    const int xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 15;
    long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < 200000) {}
    out[blockIdx.x] = xcc;
  }
}
int main() {
  for (int grid : {256, 2560, 250, 64}) {
    int* d; hipMalloc(&d, grid * 4);
    for (int rep = 0; rep < 3; ++rep) {
      hipLaunchKernelGGL(probe, dim3(grid), dim3(256), 150000, 0, d);
      hipDeviceSynchronize();
    }
    int* h = new int[grid];
    hipMemcpy(h, d, grid * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int x = 0; x < grid; ++x) if (h[x] != h[x & 7]) ++bad;
    printf("grid %d: xcc of blocks 0..15:", grid);
    for (int x = 0; x < 16; ++x) printf(" %d", h[x]);
    printf("  mismatches vs block (x & 7): %d\n", bad);
  }
  return 0;
}
