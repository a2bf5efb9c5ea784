// LDS read throughput per CU for the access patterns of k_gemm_bf16: ds_read_b128 with (a) lane-linear addresses,
// (b) the MFMA-fragment pattern (lane (g, j) -> row j, 16 B at column group g) over rows of 144 B, (c) the same over
// rows of 128 B (unpadded), and ds_read_b64 for comparison.  8 waves per CU (2 workgroups of 4), like the GEMM.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(int iters, float* sink) {
  __shared__ __attribute__((aligned(16))) char lds[36864];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, g = lane >> 4, j = lane & 15;
  for (int i = t; i < 36864 / 4; i += 256) reinterpret_cast<float*>(lds)[i] = (float)i;
  __syncthreads();
  int off;
  if (MODE == 0) off = wave * 4096 + lane * 16;
  else if (MODE == 1 || MODE == 3) off = (wave * 16 + j) * 144 + g * 16;
  else off = (wave * 16 + j) * 128 + g * 16;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const char* p = lds + off + ((u * 2304 + it * 64) & 16383);
      if (MODE == 3) { f32x2 v = *reinterpret_cast<const f32x2*>(p); acc[0] += v[0]; acc[1] += v[1]; }
      else { f32x4 v = *reinterpret_cast<const f32x4*>(p); acc += v; }
    }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 0.12345f) sink[0] = acc[0];
}
template <int MODE> void run(float* sink, const char* name) {
  const int iters = 2000, grid = 512;
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, 10, sink); hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, iters, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes_per_cu = 2.0 * 4 * 64 * (MODE == 3 ? 8 : 16) * 8.0 * iters;
  printf("%-34s %.3f ms  %.1f B/clk/CU at 2.4 GHz\n", name, ms, bytes_per_cu / (ms * 1e-3 * 2.4e9));
}
int main() {
  float* sink; hipMalloc(&sink, 4);
  run<0>(sink, "ds_read_b128 lane-linear"); run<1>(sink, "ds_read_b128 fragment, 144 B rows");
  run<2>(sink, "ds_read_b128 fragment, 128 B rows"); run<3>(sink, "ds_read_b64 fragment, 144 B rows");
  return 0;
}
