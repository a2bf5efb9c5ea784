// The inner loop of k_gemm_bf16 in isolation: LDS tiles filled once, then only ds_read_b128 fragment reads + MFMAs.
// Variants: wave tile TM x TN (in 16-row fragments), fragment double-buffering (PIPE), workgroups per CU (via LDS size).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int LDSK = 72;
template <int TM, int TN, bool PIPE>
__global__ __launch_bounds__(256) void k(int iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned short* As = reinterpret_cast<unsigned short*>(smem);          // [2 * 16 * TM rows][LDSK]
  unsigned short* Ws = As + 2 * 16 * TM * LDSK;                           // [2 * 16 * TN rows][LDSK]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, g = lane >> 4, j = lane & 15, wm = wave >> 1, wn = wave & 1;
  for (int i = t; i < (32 * TM + 32 * TN) * LDSK; i += 256) As[i] = (unsigned short)(0x3c00 + (i & 7));
  __syncthreads();
  const unsigned short* Ab = As + (size_t)(16 * TM * wm + j) * LDSK + 8 * g;
  const unsigned short* Wb = Ws + (size_t)(16 * TN * wn + j) * LDSK + 8 * g;
  f32x4 acc[TN][TM];
  for (int x = 0; x < TN; ++x) for (int y = 0; y < TM; ++y) acc[x][y] = {0.f, 0.f, 0.f, 0.f};
  if (!PIPE) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 wf[TN], af[TM];
#pragma unroll
        for (int x = 0; x < TN; ++x) wf[x] = *reinterpret_cast<const bf16x8*>(Wb + (size_t)(16 * x) * LDSK + 32 * s);
#pragma unroll
        for (int y = 0; y < TM; ++y) af[y] = *reinterpret_cast<const bf16x8*>(Ab + (size_t)(16 * y) * LDSK + 32 * s);
#pragma unroll
        for (int x = 0; x < TN; ++x)
#pragma unroll
          for (int y = 0; y < TM; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[x], af[y], acc[x][y], 0, 0, 0);
      }
      asm volatile("" ::: "memory");
    }
  } else {
    bf16x8 wf[2][TN], af[2][TM];
#pragma unroll
    for (int x = 0; x < TN; ++x) wf[0][x] = *reinterpret_cast<const bf16x8*>(Wb + (size_t)(16 * x) * LDSK);
#pragma unroll
    for (int y = 0; y < TM; ++y) af[0][y] = *reinterpret_cast<const bf16x8*>(Ab + (size_t)(16 * y) * LDSK);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int c = s, n = s ^ 1;
#pragma unroll
        for (int x = 0; x < TN; ++x) wf[n][x] = *reinterpret_cast<const bf16x8*>(Wb + (size_t)(16 * x) * LDSK + 32 * n);
#pragma unroll
        for (int y = 0; y < TM; ++y) af[n][y] = *reinterpret_cast<const bf16x8*>(Ab + (size_t)(16 * y) * LDSK + 32 * n);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int x = 0; x < TN; ++x)
#pragma unroll
          for (int y = 0; y < TM; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[c][x], af[c][y], acc[x][y], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("" ::: "memory");
    }
  }
  float s = 0.f;
  for (int x = 0; x < TN; ++x) for (int y = 0; y < TM; ++y) s += acc[x][y][0];
  if (s == 0.12345f) sink[0] = s;
}
template <int TM, int TN, bool PIPE> void run(float* sink, int lds_bytes, const char* note) {
  const int iters = 2000, grid = 256 * 4;
  const int need = (32 * TM + 32 * TN) * LDSK * 2;
  if (lds_bytes < need) lds_bytes = need;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<TM, TN, PIPE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  hipLaunchKernelGGL((k<TM, TN, PIPE>), dim3(grid), dim3(256), lds_bytes, 0, 10, sink); hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0);
  hipLaunchKernelGGL((k<TM, TN, PIPE>), dim3(grid), dim3(256), lds_bytes, 0, iters, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)grid * 4 * iters * 2 * TM * TN * 16384.0;
  printf("wave tile %3d x %3d  %s  LDS %3d KiB/WG (%s): %.3f ms  %.0f TFLOP/s  %s\n", 16 * TM, 16 * TN, PIPE ? "pipelined" : "plain    ",
         lds_bytes / 1024, note, ms, flop / ms / 1e9, hipGetErrorString(hipGetLastError()));
}
int main() {
  float* sink; hipMalloc(&sink, 4);
  run<4, 4, false>(sink, 72 * 1024, "2 WG/CU"); run<4, 4, true>(sink, 72 * 1024, "2 WG/CU");
  run<4, 4, false>(sink, 40 * 1024, "4 WG/CU"); run<4, 4, true>(sink, 40 * 1024, "4 WG/CU");
  run<8, 4, false>(sink, 72 * 1024, "2 WG/CU"); run<8, 4, true>(sink, 72 * 1024, "2 WG/CU");
  run<8, 8, false>(sink, 100 * 1024, "1 WG/CU"); run<8, 8, true>(sink, 100 * 1024, "1 WG/CU");
  run<4, 4, false>(sink, 100 * 1024, "1 WG/CU");
  return 0;
}
