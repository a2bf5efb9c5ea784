// Probe: does v_mfma_f32_16x16x32_f16 on gfx950 honour f16 SUBNORMAL operands, and does the f32 -> f16 conversion the split
// (x = hi + lo) relies on round to nearest even and produce subnormals?  Decides whether the split-fp16 sampler mode
// (k_sampler.hip PREC_F16X2) needs its weights scaled into the normal range.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
__global__ void k(float a, float b, float* out) {
    f32x8 av, bv;
    for (int i = 0; i < 8; ++i) { av[i] = a; bv[i] = b; }
    const f16x8 ah = __builtin_convertvector(av, f16x8), bh = __builtin_convertvector(bv, f16x8);
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, c, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = c[0]; out[1] = (float)ah[0]; out[2] = (float)bh[0]; }
}
int main() {
    float* d; hipMalloc(&d, 64);
    const float cases[][2] = {{1.0f, 1.0f}, {3e-5f, 1.0f}, {1e-6f, 1.0f}, {3e-5f, 3e-5f}, {5.96e-8f, 1.0f}, {1.00048828125f, 1.0f}, {1.00146484375f, 1.0f}, {65504.f, 1.f}};
    for (auto& cs : cases) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, cs[0], cs[1], d);
        float h[3]; hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
        printf("a=%.9g b=%.9g : f16(a)=%.9g f16(b)=%.9g  mfma sum over K=32 -> %.9g (expected %.9g)\n", cs[0], cs[1], h[1], h[2], h[0], 32.0 * (double)h[1] * (double)h[2]);
    }
    return 0;
}
