// How fast can ONE workgroup per CU stream an L2-resident buffer into registers?  (bytes/clk/CU)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int DEPTH>
__global__ __launch_bounds__(256) void k(const uint4* __restrict__ w, size_t units_per_wave, int passes, unsigned* sink, unsigned long long* cyc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint4* base = w + (size_t)wave * units_per_wave * 64 + lane;
  uint4 r[DEPTH];
  unsigned acc = 0;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int p = 0; p < passes; ++p) {
    const uint4* q = base;
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) r[i] = q[i * 64];
    q += DEPTH * 64;
    for (size_t u = DEPTH; u < units_per_wave; u += DEPTH) {
#pragma unroll
      for (int i = 0; i < DEPTH; ++i) { acc += (r[i].x ^ r[i].y) + (r[i].z ^ r[i].w); r[i] = q[i * 64]; }
      q += DEPTH * 64;
    }
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) acc += (r[i].x ^ r[i].y) + (r[i].z ^ r[i].w);
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (acc == 0x12345) sink[0] = acc;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int DEPTH> void run(int waves, size_t total_bytes, const uint4* d, unsigned* sink, unsigned long long* cyc, int grid = 256) {
  size_t units_per_wave = total_bytes / 1024 / waves; units_per_wave -= units_per_wave % DEPTH;
  int passes = (int)(200.0e6 / total_bytes) + 1;
  k<DEPTH><<<grid, waves * 64>>>(d, units_per_wave, 2, sink, cyc); hipDeviceSynchronize();
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); hipEventRecord(a);
  k<DEPTH><<<grid, waves * 64>>>(d, units_per_wave, passes, sink, cyc); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  double bytes = (double)units_per_wave * waves * 1024 * passes;
  printf("grid %3d waves %2d depth %2d buf %.1f MB: %.1f B/clk/CU (s_memtime)  %.1f GB/s/CU  chip %.1f TB/s\n", grid, waves, DEPTH, total_bytes / 1e6,
         bytes / (double)c, bytes / (ms * 1e-3) / 1e9, bytes * grid / (ms * 1e-3) / 1e12);
}
int main() {
  size_t N = 8 << 20; uint4* d; hipMalloc(&d, N); hipMemset(d, 1, N); unsigned* sink; hipMalloc(&sink, 4); unsigned long long* cyc; hipMalloc(&cyc, 8);
  for (int grid : {1, 86, 256}) { run<4>(4, 3800 << 10, d, sink, cyc, grid); run<8>(4, 3800 << 10, d, sink, cyc, grid); run<16>(4, 3800 << 10, d, sink, cyc, grid); run<32>(4, 3800 << 10, d, sink, cyc, grid); }
}
