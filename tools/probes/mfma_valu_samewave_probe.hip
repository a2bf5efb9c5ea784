// Probe: how many independent v_fma_f32 of the SAME wave hide under one MFMA (one wave per SIMD, independent accumulators, the fillers placed between the MFMAs with
// sched_group_barrier)?  ticks per (MFMA + NF fillers) group for v_mfma_f32_16x16x4_f32 (32-cycle pipe occupancy) against v_mfma_f32_32x32x16_bf16 (32 cycles too).
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_samewave_probe mfma_valu_samewave_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int NF>
__global__ __launch_bounds__(256) void k_sw(unsigned long long* out, int iters, float seed) {
    f32x4 acc[4];
    f32x16 big[2];
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{seed, seed, seed, seed};
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 16; ++j) big[i][j] = seed;
    bf16x8 ab, bb;
    for (int i = 0; i < 8; ++i) { ab[i] = (__bf16)seed; bb[i] = (__bf16)0.5f; }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed + i + threadIdx.x;
    const float a = seed, b = seed * 0.5f, m = seed * 0.999f;
    float c = seed * 0.001f;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if constexpr (KIND == 0) acc[u % 4] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u % 4], 0, 0, 0);
            else big[u % 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, big[u % 2], 0, 0, 0);
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                float& x = v[(u * NF + f) % 8];
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(c));   // (asm: not SLP-packed, not moved)
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0];
    for (int i = 0; i < 2; ++i) s += big[i][0];
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 123.f) out[0] = 0;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int KIND, int NF>
void run(const char* name) {
    unsigned long long* d;
    hipMalloc(&d, 4 * 256 * sizeof(unsigned long long));
    const int iters = 400;
    hipLaunchKernelGGL((k_sw<KIND, NF>), dim3(256), dim3(256), 0, 0, d, 10, 1.0f);
    hipLaunchKernelGGL((k_sw<KIND, NF>), dim3(256), dim3(256), 0, 0, d, iters, 1.0f);
    hipDeviceSynchronize();
    unsigned long long h[4 * 256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double cyc = 0;
    for (int i = 0; i < 4 * 256; ++i) cyc += (double)h[i];
    printf("%-26s + %2d v_fma_f32 per MFMA (same wave, one wave per SIMD): %5.1f ticks per group\n", name, NF, cyc / (4.0 * 256 * iters * 16));
    hipFree(d);
}
int main() {
    run<0, 0>("v_mfma_f32_16x16x4_f32"); run<0, 2>("v_mfma_f32_16x16x4_f32"); run<0, 4>("v_mfma_f32_16x16x4_f32"); run<0, 6>("v_mfma_f32_16x16x4_f32");
    run<0, 8>("v_mfma_f32_16x16x4_f32"); run<0, 12>("v_mfma_f32_16x16x4_f32");
    run<1, 0>("v_mfma_f32_32x32x16_bf16"); run<1, 2>("v_mfma_f32_32x32x16_bf16"); run<1, 4>("v_mfma_f32_32x32x16_bf16"); run<1, 6>("v_mfma_f32_32x32x16_bf16");
    run<1, 8>("v_mfma_f32_32x32x16_bf16"); run<1, 12>("v_mfma_f32_32x32x16_bf16");
    return 0;
}
