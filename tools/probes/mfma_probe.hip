// Practical matrix-pipe ceiling: independent v_mfma_f32_16x16x32_bf16 back to back, no memory traffic.
// waves per SIMD = 1, 2, 4 via the grid / block shape; accumulators 16 (no dependent stalls).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NACC>
__global__ __launch_bounds__(256) void k(int iters, float* sink) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = {0.f, 0.f, 0.f, 0.f};
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x & 3); b[e] = (__bf16)1.0f; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0];
  if (s == 0.12345f) sink[0] = s;
}
template <int NACC> void run(int wgs_per_cu, float* sink) {
  const int iters = 4000, grid = 256 * wgs_per_cu;
  hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, 10, sink); hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, iters, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)grid * 4 * iters * NACC * 32768.0;
  printf("acc %2d  %d waves/SIMD: %.3f ms  %.1f TFLOP/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", NACC, wgs_per_cu, ms, flops / ms / 1e9,
         ms * 1e-3 * 2.4e9 / ((double)wgs_per_cu * iters * NACC));
}
int main() {
  float* sink; hipMalloc(&sink, 4);
  run<4>(1, sink); run<16>(1, sink); run<16>(2, sink); run<16>(4, sink); run<32>(1, sink);
  return 0;
}
