// Probe: do VALU instructions of one wave run UNDER the fp32-input MFMAs of another wave on the same SIMD?  Workgroup of 8 waves (w and w + 4 share a SIMD):
// waves 0-3 issue NM MFMAs per iteration, waves 4-7 NV independent v_fma_f32; cycles of each role alone and together, for v_mfma_f32_16x16x4_f32 and (for scale)
// v_mfma_f32_16x16x32_bf16.  If the fp32 MFMA ran on a matrix pipe of its own, "together" would cost max(alone); if it shares the vector ALUs, the sum.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_coissue_probe mfma_valu_coissue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ __launch_bounds__(512) void k_co(unsigned long long* out, int iters, float seed, int do_mfma, int do_valu) {
    const int wave = threadIdx.x >> 6;
    unsigned long long t0 = 0, t1 = 0;
    __syncthreads();
    if (wave < 4) {
        if (do_mfma) {
            f32x4 acc[6];
            for (int i = 0; i < 6; ++i) acc[i] = f32x4{seed, seed, seed, seed};
            bf16x8 ab, bb;
            for (int i = 0; i < 8; ++i) { ab[i] = (__bf16)seed; bb[i] = (__bf16)0.5f; }
            const float a = seed, b = seed * 0.5f;
            t0 = __builtin_readcyclecounter();
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 48; ++u) {
                    if constexpr (KIND == 0) acc[u % 6] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u % 6], 0, 0, 0);
                    else acc[u % 6] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc[u % 6], 0, 0, 0);
                }
            }
            t1 = __builtin_readcyclecounter();
            float s = 0.f;
            for (int i = 0; i < 6; ++i) s += acc[i][0];
            if (s == 123.f) out[0] = 0;
        }
    } else if (do_valu) {
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = seed + i + threadIdx.x;
        const float m = seed * 0.999f, c = seed * 0.001f;
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 48 * 8; ++u) v[u % 8] = __builtin_fmaf(v[u % 8], m, c);   // 8 VALU per MFMA of the other role
        }
        t1 = __builtin_readcyclecounter();
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += v[i];
        if (s == 123.f) out[0] = 0;
    }
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int KIND>
void run(const char* name) {
    unsigned long long* d;
    hipMalloc(&d, 8 * 256 * sizeof(unsigned long long));
    const int iters = 400;
    for (int mode = 0; mode < 3; ++mode) {
        const int dm = mode != 1, dv = mode != 0;
        hipLaunchKernelGGL(k_co<KIND>, dim3(256), dim3(512), 0, 0, d, 10, 1.0f, dm, dv);
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_co<KIND>, dim3(256), dim3(512), 0, 0, d, iters, 1.0f, dm, dv);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[8 * 256];
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        double cm = 0, cv = 0;
        for (int b = 0; b < 256; ++b)
            for (int w = 0; w < 8; ++w) (w < 4 ? cm : cv) += (double)h[b * 8 + w];
        cm /= 256.0 * 4 * iters * 48;
        cv /= 256.0 * 4 * iters * 48 * 8;
        printf("%-26s %-14s wall %7.1f us;  MFMA waves: %5.1f ticks per MFMA;  VALU waves: %5.2f ticks per v_fma_f32 (8 per MFMA of the other wave)\n", name,
               mode == 0 ? "MFMA alone" : mode == 1 ? "VALU alone" : "both", ms * 1e3, dm ? cm : 0.0, dv ? cv : 0.0);
    }
    hipFree(d);
}
int main() {
    run<0>("v_mfma_f32_16x16x4_f32");
    run<1>("v_mfma_f32_16x16x32_bf16");
    return 0;
}
