// Write-bandwidth of the operand producers' store pattern: every thread writes NS streams (48 = 16 frequencies x 3
// planes, far apart) with W bytes per lane and store; a wave's lanes are contiguous within a stream.
// build: hipcc -O3 --offload-arch=gfx950 tools/ablate/store_pattern.hip -o /tmp/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
template <int WORDS, int NS>
__global__ __launch_bounds__(256) void k(unsigned* out, long stream_words, long per_stream_threads) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= per_stream_threads) return;
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    unsigned* p = out + s * stream_words + t * WORDS;
    if (WORDS == 2) *reinterpret_cast<uint2*>(p) = make_uint2((unsigned)t, s);
    if (WORDS == 4) *reinterpret_cast<uint4*>(p) = make_uint4((unsigned)t, s, 1, 2);
    if (WORDS == 1) *p = (unsigned)t;
  }
}
template <int WORDS, int NS>
void run(unsigned* buf, long total_words) {
  const long stream_words = total_words / NS, threads = stream_words / WORDS;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k<WORDS, NS>), dim3((threads + 255) / 256), dim3(256), 0, 0, buf, stream_words, threads);
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<WORDS, NS>), dim3((threads + 255) / 256), dim3(256), 0, 0, buf, stream_words, threads);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  printf("%2d bytes/lane, %2d streams: %.2f TB/s\n", WORDS * 4, NS, threads * WORDS * 4.0 * NS / (ms * 1e-3) / 1e12);
}
int main() {
  const long words = 3L << 28;   // 3 GiB
  unsigned* buf; hipMalloc(&buf, words * 4);
  run<4, 1>(buf, words); run<2, 1>(buf, words); run<1, 1>(buf, words);
  run<4, 48>(buf, words); run<2, 48>(buf, words); run<2, 16>(buf, words); run<4, 16>(buf, words); run<2, 3>(buf, words);
  return 0;
}
