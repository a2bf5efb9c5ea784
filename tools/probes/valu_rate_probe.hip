// Probe: issue cost (shader cycles per wave64 instruction) of the VALU instructions the fused decoder's softmax and GELU are
// made of, with 1 and 2 waves per SIMD, alone and beside MFMAs of another wave.  Decides what bounds k_vae_fused: its SIMDs
// issue 8 VALU instructions per MFMA, 84 of them v_exp_f32 per (query tile, head).
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate_probe valu_rate_probe.hip      Run: ./valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define REP16(x) x x x x x x x x x x x x x x x x
// blocks of 128 instructions per loop iteration (the taken branch costs ~30 cycles); 16 independent destinations v[16..47], sources v[2..5]: no dependent chains, no hazards between neighbours
#define BODY(NAME, ASM)                                                                                  \
    __global__ __launch_bounds__(512) void NAME(unsigned long long* out, int iters, int mfma_waves) {    \
        const int wave = threadIdx.x >> 6;                                                               \
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};                                                                \
        bf16x8 a, b;                                                                                     \
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)1.0f; b[i] = (__bf16)0.5f; }                        \
        __syncthreads();                                                                                 \
        const unsigned long long t0 = __builtin_readcyclecounter();                                     \
        if (wave < mfma_waves) {                                                                         \
            for (int it = 0; it < iters; ++it) {                                                         \
                _Pragma("unroll") for (int u = 0; u < 128; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0); \
            }                                                                                            \
        } else {                                                                                         \
            for (int it = 0; it < iters; ++it) {                                                         \
                asm volatile(ASM ASM ASM ASM ASM ASM ASM ASM ::: "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", \
                             "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42",  \
                             "v43", "v44", "v45", "v46", "v47");                                         \
            }                                                                                            \
        }                                                                                                \
        const unsigned long long t1 = __builtin_readcyclecounter();                                     \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = t1 - t0;                               \
        if (acc[0] == 123.f) out[0] = 0;                                                                 \
    }

BODY(k_exp32, "v_exp_f32 v16, v2\n v_exp_f32 v17, v3\n v_exp_f32 v18, v4\n v_exp_f32 v19, v5\n v_exp_f32 v20, v2\n v_exp_f32 v21, v3\n v_exp_f32 v22, v4\n v_exp_f32 v23, v5\n"
               "v_exp_f32 v24, v2\n v_exp_f32 v25, v3\n v_exp_f32 v26, v4\n v_exp_f32 v27, v5\n v_exp_f32 v28, v2\n v_exp_f32 v29, v3\n v_exp_f32 v30, v4\n v_exp_f32 v31, v5\n")
BODY(k_exp16, "v_exp_f16 v16, v2\n v_exp_f16 v17, v3\n v_exp_f16 v18, v4\n v_exp_f16 v19, v5\n v_exp_f16 v20, v2\n v_exp_f16 v21, v3\n v_exp_f16 v22, v4\n v_exp_f16 v23, v5\n"
               "v_exp_f16 v24, v2\n v_exp_f16 v25, v3\n v_exp_f16 v26, v4\n v_exp_f16 v27, v5\n v_exp_f16 v28, v2\n v_exp_f16 v29, v3\n v_exp_f16 v30, v4\n v_exp_f16 v31, v5\n")
BODY(k_fma32, "v_fma_f32 v16, v2, v3, v4\n v_fma_f32 v17, v3, v4, v5\n v_fma_f32 v18, v2, v3, v4\n v_fma_f32 v19, v3, v4, v5\n v_fma_f32 v20, v2, v3, v4\n v_fma_f32 v21, v3, v4, v5\n v_fma_f32 v22, v2, v3, v4\n v_fma_f32 v23, v3, v4, v5\n"
               "v_fma_f32 v24, v2, v3, v4\n v_fma_f32 v25, v3, v4, v5\n v_fma_f32 v26, v2, v3, v4\n v_fma_f32 v27, v3, v4, v5\n v_fma_f32 v28, v2, v3, v4\n v_fma_f32 v29, v3, v4, v5\n v_fma_f32 v30, v2, v3, v4\n v_fma_f32 v31, v3, v4, v5\n")
BODY(k_pkfma32, "v_pk_fma_f32 v[16:17], v[2:3], v[4:5], v[2:3]\n v_pk_fma_f32 v[18:19], v[2:3], v[4:5], v[2:3]\n v_pk_fma_f32 v[20:21], v[2:3], v[4:5], v[2:3]\n v_pk_fma_f32 v[22:23], v[2:3], v[4:5], v[2:3]\n"
                 "v_pk_fma_f32 v[24:25], v[2:3], v[4:5], v[2:3]\n v_pk_fma_f32 v[26:27], v[2:3], v[4:5], v[2:3]\n v_pk_fma_f32 v[28:29], v[2:3], v[4:5], v[2:3]\n v_pk_fma_f32 v[30:31], v[2:3], v[4:5], v[2:3]\n"
                 "v_pk_fma_f32 v[32:33], v[2:3], v[4:5], v[2:3]\n v_pk_fma_f32 v[34:35], v[2:3], v[4:5], v[2:3]\n v_pk_fma_f32 v[36:37], v[2:3], v[4:5], v[2:3]\n v_pk_fma_f32 v[38:39], v[2:3], v[4:5], v[2:3]\n"
                 "v_pk_fma_f32 v[40:41], v[2:3], v[4:5], v[2:3]\n v_pk_fma_f32 v[42:43], v[2:3], v[4:5], v[2:3]\n v_pk_fma_f32 v[44:45], v[2:3], v[4:5], v[2:3]\n v_pk_fma_f32 v[46:47], v[2:3], v[4:5], v[2:3]\n")
BODY(k_pkfma32dep, "v_pk_fma_f32 v[16:17], v[2:3], v[16:17], v[2:3]\n s_nop 0\n v_pk_fma_f32 v[16:17], v[2:3], v[16:17], v[2:3]\n s_nop 0\n v_pk_fma_f32 v[16:17], v[2:3], v[16:17], v[2:3]\n s_nop 0\n v_pk_fma_f32 v[16:17], v[2:3], v[16:17], v[2:3]\n s_nop 0\n"
                    "v_pk_fma_f32 v[16:17], v[2:3], v[16:17], v[2:3]\n s_nop 0\n v_pk_fma_f32 v[16:17], v[2:3], v[16:17], v[2:3]\n s_nop 0\n v_pk_fma_f32 v[16:17], v[2:3], v[16:17], v[2:3]\n s_nop 0\n v_pk_fma_f32 v[16:17], v[2:3], v[16:17], v[2:3]\n s_nop 0\n"
                    "v_pk_fma_f32 v[16:17], v[2:3], v[16:17], v[2:3]\n s_nop 0\n v_pk_fma_f32 v[16:17], v[2:3], v[16:17], v[2:3]\n s_nop 0\n v_pk_fma_f32 v[16:17], v[2:3], v[16:17], v[2:3]\n s_nop 0\n v_pk_fma_f32 v[16:17], v[2:3], v[16:17], v[2:3]\n s_nop 0\n"
                    "v_pk_fma_f32 v[16:17], v[2:3], v[16:17], v[2:3]\n s_nop 0\n v_pk_fma_f32 v[16:17], v[2:3], v[16:17], v[2:3]\n s_nop 0\n v_pk_fma_f32 v[16:17], v[2:3], v[16:17], v[2:3]\n s_nop 0\n v_pk_fma_f32 v[16:17], v[2:3], v[16:17], v[2:3]\n s_nop 0\n")
BODY(k_cvtpk, "v_cvt_pk_bf16_f32 v16, v2, v3\n v_cvt_pk_bf16_f32 v17, v4, v5\n v_cvt_pk_bf16_f32 v18, v2, v3\n v_cvt_pk_bf16_f32 v19, v4, v5\n v_cvt_pk_bf16_f32 v20, v2, v3\n v_cvt_pk_bf16_f32 v21, v4, v5\n v_cvt_pk_bf16_f32 v22, v2, v3\n v_cvt_pk_bf16_f32 v23, v4, v5\n"
               "v_cvt_pk_bf16_f32 v24, v2, v3\n v_cvt_pk_bf16_f32 v25, v4, v5\n v_cvt_pk_bf16_f32 v26, v2, v3\n v_cvt_pk_bf16_f32 v27, v4, v5\n v_cvt_pk_bf16_f32 v28, v2, v3\n v_cvt_pk_bf16_f32 v29, v4, v5\n v_cvt_pk_bf16_f32 v30, v2, v3\n v_cvt_pk_bf16_f32 v31, v4, v5\n")
BODY(k_max3, "v_max3_f32 v16, v2, v3, v4\n v_max3_f32 v17, v3, v4, v5\n v_max3_f32 v18, v2, v3, v4\n v_max3_f32 v19, v3, v4, v5\n v_max3_f32 v20, v2, v3, v4\n v_max3_f32 v21, v3, v4, v5\n v_max3_f32 v22, v2, v3, v4\n v_max3_f32 v23, v3, v4, v5\n"
              "v_max3_f32 v24, v2, v3, v4\n v_max3_f32 v25, v3, v4, v5\n v_max3_f32 v26, v2, v3, v4\n v_max3_f32 v27, v3, v4, v5\n v_max3_f32 v28, v2, v3, v4\n v_max3_f32 v29, v3, v4, v5\n v_max3_f32 v30, v2, v3, v4\n v_max3_f32 v31, v3, v4, v5\n")

template <class K>
void run(const char* name, K kern, int waves, int mfma_waves, unsigned long long* d) {
    const int iters = 1024;
    hipLaunchKernelGGL(kern, dim3(1), dim3(64 * waves), 0, 0, d, iters, mfma_waves);
    hipLaunchKernelGGL(kern, dim3(1), dim3(64 * waves), 0, 0, d, iters, mfma_waves);
    unsigned long long h[8];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    // waves 0..3 sit on SIMDs 0..3, waves 4..7 again on SIMDs 0..3 (two per SIMD)
    printf("%-22s %d waves (%d MFMA waves): ", name, waves, mfma_waves);
    for (int w = 0; w < waves; ++w) printf("%s%.2f", w ? " / " : "", (double)h[w] / (128.0 * iters));
    printf("  cycles per instruction and wave\n");
}

int main() {
    unsigned long long* d;
    hipMalloc(&d, 4096);
#define RUNALL(NAME, K)                   \
    run(NAME, K, 4, 0, d);                \
    run(NAME, K, 8, 0, d);                \
    run(NAME, K, 8, 4, d);
    RUNALL("v_exp_f32", k_exp32)
    RUNALL("v_exp_f16", k_exp16)
    RUNALL("v_fma_f32", k_fma32)
    RUNALL("v_pk_fma_f32", k_pkfma32)
    RUNALL("v_pk_fma_f32 dep+nop", k_pkfma32dep)
    RUNALL("v_cvt_pk_bf16_f32", k_cvtpk)
    RUNALL("v_max3_f32", k_max3)
    return 0;
}
