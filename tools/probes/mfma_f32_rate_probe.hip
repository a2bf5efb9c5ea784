// Probe: sustained issue rate of the fp32-input MFMAs (v_mfma_f32_16x16x4_f32, v_mfma_f32_32x32x2_f32) and, for scale, v_mfma_f32_16x16x32_bf16:
// shader cycles and nanoseconds per instruction and SIMD with 1 and 2 waves per SIMD, on one workgroup (an idle chip) and on 256 / 512 workgroups (every CU busy:
// what the clock does under load).  Six independent accumulators per wave, no memory traffic.  Decides what "MFMA-bound" means for csrc/k_train_gemm.hip.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_f32_rate_probe mfma_f32_rate_probe.hip      Run: ./mfma_f32_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ __launch_bounds__(512) void k_rate(unsigned long long* out, int iters, float seed) {
    const int wave = threadIdx.x >> 6;
    unsigned long long t0, t1;
    if constexpr (KIND == 0) {
        f32x4 acc[6];
        for (int i = 0; i < 6; ++i) acc[i] = f32x4{seed, seed, seed, seed};
        const float a = seed, b = seed * 0.5f;
        __syncthreads();
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 96; ++u) acc[u % 6] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u % 6], 0, 0, 0);
        }
        t1 = __builtin_readcyclecounter();
        float s = 0.f;
        for (int i = 0; i < 6; ++i) s += acc[i][0];
        if (s == 123.f) out[0] = 0;
    } else if constexpr (KIND == 1) {
        f32x16 acc[3];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 16; ++j) acc[i][j] = seed;
        const float a = seed, b = seed * 0.5f;
        __syncthreads();
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 96; ++u) acc[u % 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u % 3], 0, 0, 0);
        }
        t1 = __builtin_readcyclecounter();
        float s = 0.f;
        for (int i = 0; i < 3; ++i) s += acc[i][0];
        if (s == 123.f) out[0] = 0;
    } else {
        f32x4 acc[6];
        for (int i = 0; i < 6; ++i) acc[i] = f32x4{seed, seed, seed, seed};
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)seed; b[i] = (__bf16)0.5f; }
        __syncthreads();
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 96; ++u) acc[u % 6] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[u % 6], 0, 0, 0);
        }
        t1 = __builtin_readcyclecounter();
        float s = 0.f;
        for (int i = 0; i < 6; ++i) s += acc[i][0];
        if (s == 123.f) out[0] = 0;
    }
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int KIND>
void run(const char* name, double flop_per_instr) {
    unsigned long long* d;
    hipMalloc(&d, 8 * 1024 * sizeof(unsigned long long));
    const int iters = 400;
    for (int blocks : {1, 256, 512}) {
        for (int waves : {4, 8}) {
            if (blocks == 512 && waves == 8) continue;
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(64 * waves), 0, 0, d, 10, 1.0f);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(64 * waves), 0, 0, d, iters, 1.0f);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h(8 * blocks);
            hipMemcpy(h.data(), d, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            double cyc = 0;
            for (int b = 0; b < blocks; ++b)
                for (int w = 0; w < waves; ++w) cyc += (double)h[b * 8 + w];
            cyc /= (double)blocks * waves;
            const double per_wave_instr = 96.0 * iters, wps = waves / 4.0;
            const double instr_total = per_wave_instr * waves * blocks;
            printf("%-28s %3d workgroups x %d waves (%g per SIMD): %6.1f counter ticks per instruction and SIMD; wall %7.1f us -> %5.1f ns per instruction and SIMD, %7.1f TFLOP/s\n", name, blocks,
                   waves, wps, cyc / (per_wave_instr * wps), ms * 1e3, ms * 1e6 / (per_wave_instr * wps * (blocks > 256 ? 2 : 1)), instr_total * flop_per_instr / (ms * 1e-3) * 1e-12);
        }
    }
    hipFree(d);
}

int main() {
    run<0>("v_mfma_f32_16x16x4_f32", 2048.0);
    run<1>("v_mfma_f32_32x32x2_f32", 4096.0);
    run<2>("v_mfma_f32_16x16x32_bf16", 16384.0);
    return 0;
}
