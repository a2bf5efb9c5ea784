// Does issuing loads in bursts (issue N, then consume N) deliver less L2->CU bandwidth than a rolling window?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int N, bool ROLLING>
__global__ __launch_bounds__(256) void k(const u32x4* __restrict__ w, int units_per_wave, int passes, unsigned* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const u32x4* base = w + (size_t)wave * units_per_wave * 64 + lane;
  u32x4 r[N];
  unsigned acc = 0;
  for (int p = 0; p < passes; ++p) {
    const u32x4* q = base;
    if (ROLLING) {
#pragma unroll
      for (int i = 0; i < N; ++i) r[i] = q[i * 64];
      q += N * 64;
      for (int u = N; u < units_per_wave; u += N) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
          acc += (r[i].x ^ r[i].y) + (r[i].z ^ r[i].w);
          r[i] = q[i * 64];
          __builtin_amdgcn_sched_barrier(0);
        }
        q += N * 64;
      }
#pragma unroll
      for (int i = 0; i < N; ++i) acc += (r[i].x ^ r[i].y) + (r[i].z ^ r[i].w);
    } else {
      for (int u = 0; u < units_per_wave; u += N) {
#pragma unroll
        for (int i = 0; i < N; ++i) r[i] = q[i * 64];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < N; ++i) acc += (r[i].x ^ r[i].y) + (r[i].z ^ r[i].w);
        __builtin_amdgcn_sched_barrier(0);
        q += N * 64;
      }
    }
  }
  if (acc == 0x12345) sink[0] = acc;
}
template <int N, bool ROLLING> void run(const u32x4* d, unsigned* sink, int grid) {
  int upw = 928 - 928 % N, passes = 200;
  k<N, ROLLING><<<grid, 256>>>(d, upw, 2, sink); hipDeviceSynchronize();
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); hipEventRecord(a);
  k<N, ROLLING><<<grid, 256>>>(d, upw, passes, sink); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double bytes = (double)upw * 4 * 1024 * passes;
  printf("grid %3d N %2d %s: %.1f GB/s/CU  (%.1f us per 3.8 MB pass)\n", grid, N, ROLLING ? "rolling" : "burst  ", bytes / (ms * 1e-3) / 1e9, ms * 1e3 / passes);
}
int main() {
  u32x4* d; hipMalloc(&d, 8 << 20); hipMemset(d, 1, 8 << 20); unsigned* sink; hipMalloc(&sink, 4);
  for (int grid : {1, 256}) {
    run<4, true>(d, sink, grid); run<8, true>(d, sink, grid); run<16, true>(d, sink, grid); run<32, true>(d, sink, grid);
    run<4, false>(d, sink, grid); run<8, false>(d, sink, grid); run<16, false>(d, sink, grid); run<32, false>(d, sink, grid);
  }
}
