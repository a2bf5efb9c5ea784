// Loader/consumer wave specialisation probe: 8-wave workgroups, waves 4..7 stream their pair's weight slice
// global -> LDS with global_load_lds_dwordx4 (no VGPR destination, counted vmcnt), waves 0..3 consume the chunks
// from LDS (ds_read_b128 -> MFMA) with an optional dependent-VALU chain that stands in for the latency-bound
// phases of k_sample.  One s_barrier per chunk publishes chunk c and frees the slot of chunk c-1.
// Question: what L2->CU stream rate does this sustain per CU (all 256 CUs reading the same 3.8 MB), and how much
// consumer work per chunk is hidden?  Compare with tools/probes/burst_probe.hip (register ring: ~143 GB/s/CU).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int C, int R, int WORK, bool STREAM>
__global__ __launch_bounds__(512) void k(const uint4* __restrict__ w, int chunks_per_wave, int passes, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int pair = wave & 3;
  const bool loader = wave >= 4;
  const unsigned ring = (unsigned)(size_t)lds + pair * (R * C * 1024);  // LDS byte address of this pair's ring
  const uint4* src = w + (size_t)pair * chunks_per_wave * C * 64 + lane;
  const int total = passes * chunks_per_wave;
  if (loader) {
    int ci = 0;   // chunk index within the pass (stream wraps)
    int slot = 0;
    auto issue = [&]() {
      const uint4* s = src + (size_t)ci * C * 64;
      const unsigned d = ring + slot * (C * 1024);
      if (STREAM) {
#pragma unroll
        for (int u = 0; u < C; ++u) glds16(s + u * 64, __builtin_amdgcn_readfirstlane(d + u * 1024));
      }
      ci = (ci + 1 == chunks_per_wave) ? 0 : ci + 1;
      slot = (slot + 1 == R) ? 0 : slot + 1;
    };
    for (int i = 0; i < R - 1; ++i) issue();
    for (int c = 0; c < total; ++c) {
      wait_vm<(R - 2) * C>();            // chunk c has landed (chunks c+1 .. c+R-2 may still be in flight)
      __builtin_amdgcn_s_barrier();      // publish chunk c; the consumers are done with chunk c-1
      issue();                           // chunk c+R-1 -> the slot of chunk c-1
    }
    wait_vm<0>();
  } else {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
    int slot = 0;
    const bf16x8 xb = {1, 1, 1, 1, 1, 1, 1, 1};
    for (int c = 0; c < total; ++c) {
      __builtin_amdgcn_s_barrier();
      const char* base = lds + pair * (R * C * 1024) + slot * (C * 1024) + lane * 16;
#pragma unroll
      for (int u = 0; u < C; ++u) {
        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(base + u * 1024);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xb, acc, 0, 0, 0);
      }
#pragma unroll 4
      for (int i = 0; i < WORK; ++i) acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xb, xb, acc2, 0, 0, 0);  // dependent chain
      slot = (slot + 1 == R) ? 0 : slot + 1;
    }
    if (acc[0] + acc2[0] == 0.12345f) sink[0] = acc[1];
  }
}

template <int C, int R, int WORK, bool STREAM = true>
void run(const uint4* d, float* sink, int grid) {
  const int upw = 928 - 928 % C, cpw = upw / C, passes = 100;
  const int ldsb = 4 * R * C * 1024;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<C, R, WORK, STREAM>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
  hipLaunchKernelGGL((k<C, R, WORK, STREAM>), dim3(grid), dim3(512), ldsb, 0, d, cpw, 2, sink);
  hipDeviceSynchronize();
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); hipEventRecord(a);
  hipLaunchKernelGGL((k<C, R, WORK, STREAM>), dim3(grid), dim3(512), ldsb, 0, d, cpw, passes, sink);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double bytes = (double)upw * 4 * 1024 * passes;
  printf("%s grid %3d  chunk %2d KiB/wave  ring %d chunks (%3d KiB LDS)  work %4d: %6.1f GB/s/CU  %6.1f us per 3.8 MB pass  (%.0f ns per chunk)  %s\n",
         STREAM ? "stream" : "nostrm", grid, C, R, ldsb / 1024, WORK, bytes / (ms * 1e-3) / 1e9, ms * 1e3 / passes, ms * 1e6 / passes / cpw,
         hipGetErrorString(hipGetLastError()));
}

int main() {
  uint4* d; hipMalloc(&d, 8 << 20); hipMemset(d, 1, 8 << 20);
  float* sink; hipMalloc(&sink, 4);
  for (int grid : {256}) {
    run<4, 6, 0>(d, sink, grid); run<4, 6, 0, false>(d, sink, grid);
    run<4, 6, 4>(d, sink, grid); run<4, 6, 4, false>(d, sink, grid);
    run<4, 6, 8>(d, sink, grid); run<4, 6, 8, false>(d, sink, grid);
    run<4, 6, 16>(d, sink, grid); run<4, 6, 16, false>(d, sink, grid);
    run<4, 6, 32>(d, sink, grid); run<4, 6, 32, false>(d, sink, grid);
    run<8, 4, 16>(d, sink, grid); run<8, 4, 16, false>(d, sink, grid);
    run<8, 4, 32>(d, sink, grid); run<8, 4, 32, false>(d, sink, grid);
    run<8, 4, 64>(d, sink, grid); run<8, 4, 64, false>(d, sink, grid);
  }
  return 0;
}
