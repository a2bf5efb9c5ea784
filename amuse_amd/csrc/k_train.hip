// The row-wise glue of the train_gesture step's transformer layers as fp32 HIP kernels (BASELINE config 4; reference scripts/trainer.py:335-498
// runs models/latent_diffusion/vae.py / denoiser.py under autograd; layer arithmetic utils/cross_attention.py:259-272 (encoder layer, forward_post)
// and :323-345 (decoder layer)): everything between the GEMMs and the attention of a layer's forward AND backward pass, so that a layer is one
// autograd.Function of ~8 + ~18 launches (amuse_amd/train_ops.py) instead of ~14 + ~25 eager ones:
//
//   ln_fwd        out = LayerNorm(x + dropout(y + bias))            the residual branch's bias, dropout1/2/3, the add and norm1/2/3 in one pass;
//                                                                   keeps zhat = (z - mean) rstd and rstd for the backward pass
//   ln_bwd        dz = LayerNorm backward of (dout + dout2); dx = dz, dy = dz . mask / (1 - p), dgamma, dbeta, dbias = column sums of dy
//   bias_gelu_drop_fwd   a = dropout(gelu(h + b))                   linear1's bias, exact-erf GELU, the FFN's inner dropout
//   bias_gelu_drop_bwd   dh = da . mask / (1 - p) . gelu'(h + b), db = column sums of dh
//   colsum        out[c] = sum_rows x[r][c]                         in_proj's bias gradient
//
// Dropout masks are counter-based: element e of a call is draw e % 4 of Philox4x32-10(key = seed, counter = (e / 4, offset)) - nothing is stored,
// the backward kernels regenerate the mask from the same (seed, offset); keep <=> u >= p with u = the draw's top 24 bits / 2^24.
// Column sums are deterministic: every workgroup writes its partial sums, a one-workgroup launch behind it adds them up in a fixed order.  All arrays fp32, row-major [rows][C]; HBM-bound by construction (each array is read or written once).
#include "amuse_dev.hpp"
#include "amuse_host.hpp"

namespace amuse {
namespace {

constexpr int kTrainWgs = 256;   // workgroups (at most) of the kernels with column sums = rows of the partial-sum workspace

__device__ __forceinline__ uint4 drop_bits(uint64_t seed, uint64_t offset, uint64_t e4) {
    uint32_t c[4] = {(uint32_t)e4, (uint32_t)(e4 >> 32), (uint32_t)offset, (uint32_t)(offset >> 32)};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    return uint4{c[0], c[1], c[2], c[3]};
}
// mask . 1 / (1 - p) of the four elements 4 e4 .. 4 e4 + 3 (thr = p 2^24; thr == 0: no dropout)
__device__ __forceinline__ f32x4 drop_scale4(uint64_t seed, uint64_t offset, uint64_t e4, uint32_t thr, float scale) {
    if (thr == 0) return splat4(1.0f);
    const uint4 b = drop_bits(seed, offset, e4);
    return f32x4{(b.x >> 8) >= thr ? scale : 0.f, (b.y >> 8) >= thr ? scale : 0.f, (b.z >> 8) >= thr ? scale : 0.f, (b.w >> 8) >= thr ? scale : 0.f};
}
__device__ __forceinline__ float half_wave_sum(float v) {   // over the 32 lanes that share a row
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m, 32);
    return v;
}

// ---- LayerNorm over C = 128: a half wave per row (lane & 31 = the row's 4-column group), 8 rows per workgroup and iteration
__global__ __launch_bounds__(256) void k_train_ln_fwd(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ bias,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta, uint32_t thr, float scale,
                                                      uint64_t seed, uint64_t offset, long rows, float* __restrict__ out, float* __restrict__ zhat,
                                                      float* __restrict__ rstd_out) {
    const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const f32x4 bi = bias ? ld4(bias + 4 * cg) : splat4(0.f), ga = ld4(gamma + 4 * cg), be = ld4(beta + 4 * cg);
    for (long r = (long)blockIdx.x * 8 + rl; r < rows; r += (long)gridDim.x * 8) {
        const size_t e = (size_t)r * 128 + 4 * cg;
        const f32x4 yv = (ld4(y + e) + bi) * drop_scale4(seed, offset, e >> 2, thr, scale);
        const f32x4 z = x ? ld4(x + e) + yv : yv;
        const float mean = half_wave_sum((z[0] + z[1]) + (z[2] + z[3])) * (1.0f / 128.0f);
        const f32x4 d = z - splat4(mean);
        const float var = half_wave_sum((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3])) * (1.0f / 128.0f);
        const float rstd = 1.0f / sqrtf(var + 1e-5f);
        const f32x4 zh = d * rstd;
        st4(out + e, zh * ga + be);
        if (zhat) st4(zhat + e, zh);
        if (rstd_out && cg == 0) rstd_out[r] = rstd;
    }
}

// ---- column sums: every workgroup (1,024 threads) writes its partial sums ws[workgroup][ncol]; k_train_finalize (one workgroup, the next launch) adds
// them up - ncol / 4 float4 columns x as many groups of partials as fit 1,024 threads, independent loads, then the groups in order.  (A "last
// workgroup adds up" tail inside the same launch needs agent-scope fences, and on this chip - eight L2s - those write the XCD's dirty lines back:
// measured 40-70 us per launch against ~10 for the two launches.)
constexpr int kTrainThreads = 1024;
// red: >= 1,024 float4 of LDS.  outs[j] (nullable) receives columns [j * C, (j + 1) * C)
__device__ __forceinline__ void sum_partials(const float* ws, int n, float* const* outs, int ncol, int C, f32x4* red) {
    const int nc4 = ncol / 4, groups = kTrainThreads / nc4;
    const int c4 = threadIdx.x % nc4, grp = threadIdx.x / nc4;
    f32x4 acc = splat4(0.f);
    if (grp < groups)
        for (int w = grp; w < n; w += 4 * groups) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = w + u * groups < n ? ld4(ws + (size_t)(w + u * groups) * ncol + 4 * c4) : splat4(0.f);
            acc += (v[0] + v[1]) + (v[2] + v[3]);
        }
    __syncthreads();
    red[threadIdx.x] = acc;
    __syncthreads();
    if ((int)threadIdx.x < nc4) {
        f32x4 s = red[threadIdx.x];
        for (int q = 1; q < groups; ++q) s += red[q * nc4 + threadIdx.x];
        const int c = 4 * threadIdx.x;
        float* o = outs[c / C];
        if (o) st4(o + c % C, s);
    }
}
// a workgroup's own partial: thread (row lane rl, column group cg) holds acc; lanes are added in order
__device__ __forceinline__ void workgroup_partial(f32x4 acc, int nc4, int lanes, float* ws_row, f32x4* red) {
    red[threadIdx.x] = acc;
    __syncthreads();
    if ((int)threadIdx.x < nc4) {
        f32x4 s = red[threadIdx.x];
        for (int q = 1; q < lanes; ++q) s += red[q * nc4 + threadIdx.x];
        st4(ws_row + 4 * threadIdx.x, s);
    }
}

__global__ __launch_bounds__(kTrainThreads) void k_train_ln_bwd(const float* __restrict__ dout, const float* __restrict__ dout2, const float* __restrict__ zhat, const float* __restrict__ rstd_in,
                                                      const float* __restrict__ gamma, uint32_t thr, float scale, uint64_t seed, uint64_t offset, long rows,
                                                      float* __restrict__ dx, float* __restrict__ dy, float* ws) {
    __shared__ f32x4 red[kTrainThreads];
    const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;   // 32 row lanes
    const f32x4 ga = ld4(gamma + 4 * cg);
    f32x4 sg = splat4(0.f), sb = splat4(0.f), sy = splat4(0.f);
    // two rows per thread and iteration: both rows' loads are in flight before the first is used (the kernel is latency-bound otherwise)
    const long G = (long)gridDim.x * 32;
    for (long r0 = (long)blockIdx.x * 32 + rl; r0 < rows; r0 += 2 * G) {
        const long rr[2] = {r0, r0 + G};
        f32x4 go[2], zh[2];
        float rstd[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const bool ok = rr[u] < rows;   // (uniform over the half wave that shares the row)
            const size_t e = (size_t)(ok ? rr[u] : r0) * 128 + 4 * cg;
            go[u] = ok ? (dout2 ? ld4(dout + e) + ld4(dout2 + e) : ld4(dout + e)) : splat4(0.f);
            zh[u] = ld4(zhat + e);
            rstd[u] = rstd_in[ok ? rr[u] : r0];
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const f32x4 gy = go[u] * ga;
            const float m1 = half_wave_sum((gy[0] + gy[1]) + (gy[2] + gy[3])) * (1.0f / 128.0f);
            const float m2 = half_wave_sum((gy[0] * zh[u][0] + gy[1] * zh[u][1]) + (gy[2] * zh[u][2] + gy[3] * zh[u][3])) * (1.0f / 128.0f);
            const f32x4 dz = (gy - splat4(m1) - zh[u] * m2) * rstd[u];
            if (rr[u] < rows) {
                const size_t e = (size_t)rr[u] * 128 + 4 * cg;
                const f32x4 dyv = dz * drop_scale4(seed, offset, e >> 2, thr, scale);
                if (dx) st4(dx + e, dz);
                st4(dy + e, dyv);
                sg += go[u] * zh[u];
                sb += go[u];
                sy += dyv;
            }
        }
    }
    float* wrow = ws + (size_t)blockIdx.x * 384;
    workgroup_partial(sg, 32, 32, wrow, red);
    __syncthreads();
    workgroup_partial(sb, 32, 32, wrow + 128, red);
    __syncthreads();
    workgroup_partial(sy, 32, 32, wrow + 256, red);
}
__global__ __launch_bounds__(kTrainThreads) void k_train_finalize(const float* ws, int n, float* o0, float* o1, float* o2, int ncol, int C) {
    __shared__ f32x4 red[kTrainThreads];
    float* outs[3] = {o0, o1, o2};
    sum_partials(ws, n, outs, ncol, C, red);
}

// ---- FFN activation: a = dropout(gelu(h + b)), exact erf (F.gelu's default)
__device__ __forceinline__ float gelu_exact(float u) { return 0.5f * u * (1.0f + erff(u * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad(float u) {
    return 0.5f * (1.0f + erff(u * 0.70710678118654752440f)) + u * 0.39894228040143267794f * expf(-0.5f * u * u);
}
__global__ __launch_bounds__(256) void k_train_bgd_fwd(const float* __restrict__ h, const float* __restrict__ b, uint32_t thr, float scale, uint64_t seed,
                                                       uint64_t offset, size_t n4, int F4, float* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const f32x4 u = ld4(h + 4 * i) + ld4(b + 4 * (i % F4));
        const f32x4 g = f32x4{gelu_exact(u[0]), gelu_exact(u[1]), gelu_exact(u[2]), gelu_exact(u[3])};
        st4(out + 4 * i, g * drop_scale4(seed, offset, i, thr, scale));
    }
}
// column sums ride along: a thread owns one 4-column group (F4 <= 256 groups) for all its rows - 1,024 / F4 row lanes per workgroup
__global__ __launch_bounds__(kTrainThreads) void k_train_bgd_bwd(const float* __restrict__ da, const float* __restrict__ h, const float* __restrict__ b, uint32_t thr,
                                                       float scale, uint64_t seed, uint64_t offset, long rows, int F4, float* __restrict__ dh, float* ws) {
    __shared__ f32x4 red[kTrainThreads];
    const int lanes = kTrainThreads / F4;
    const int cg = threadIdx.x % F4, rl = threadIdx.x / F4;
    f32x4 acc = splat4(0.f);
    if (rl < lanes) {
        const f32x4 bi = ld4(b + 4 * cg);
        const long G = (long)gridDim.x * lanes;
        for (long r0 = (long)blockIdx.x * lanes + rl; r0 < rows; r0 += 4 * G) {
            f32x4 hv[4], dv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long r = r0 + u * G;
                const size_t i = (size_t)(r < rows ? r : r0) * F4 + cg;
                hv[u] = ld4(h + 4 * i);
                dv[u] = ld4(da + 4 * i);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long r = r0 + u * G;
                if (r >= rows) continue;
                const size_t i = (size_t)r * F4 + cg;
                const f32x4 uu = hv[u] + bi;
                const f32x4 g = f32x4{gelu_grad(uu[0]), gelu_grad(uu[1]), gelu_grad(uu[2]), gelu_grad(uu[3])};
                const f32x4 d = dv[u] * drop_scale4(seed, offset, i, thr, scale) * g;
                st4(dh + 4 * i, d);
                acc += d;
            }
        }
    }
    workgroup_partial(acc, F4, lanes, ws + (size_t)blockIdx.x * 4 * F4, red);
}
__global__ __launch_bounds__(kTrainThreads) void k_train_colsum(const float* __restrict__ x, long rows, int C4, float* ws) {
    __shared__ f32x4 red[kTrainThreads];
    const int lanes = kTrainThreads / C4;
    const int cg = threadIdx.x % C4, rl = threadIdx.x / C4;
    f32x4 acc = splat4(0.f);
    if (rl < lanes)
    {
        const long G = (long)gridDim.x * lanes;
        for (long r0 = (long)blockIdx.x * lanes + rl; r0 < rows; r0 += 4 * G) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long r = r0 + u * G;
                v[u] = r < rows ? ld4(x + ((size_t)r * C4 + cg) * 4) : splat4(0.f);
            }
            acc += (v[0] + v[1]) + (v[2] + v[3]);
        }
    }
    workgroup_partial(acc, C4, lanes, ws + (size_t)blockIdx.x * 4 * C4, red);
}

int drop_args(float p, uint32_t* thr, float* scale) {
    if (!(p >= 0.f) || p >= 1.f) return fail(AMUSE_EINVAL, "dropout probability %g outside [0, 1)", (double)p);
    *thr = (uint32_t)(p * 16777216.0f);
    *scale = 1.0f / (1.0f - p);
    return 0;
}
int grid_for(long rows, int rows_per_wg) {
    const long g = (rows + rows_per_wg - 1) / rows_per_wg;
    return (int)(g < 1 ? 1 : g > kTrainWgs ? kTrainWgs : g);
}

}  // namespace
}  // namespace amuse

using namespace amuse;

extern "C" {

size_t amuse_train_ws_floats(void) { return (size_t)kTrainWgs * 1024; }   // [workgroups][up to 1,024 columns]

int amuse_train_ln_fwd(const float* x, const float* y, const float* bias, const float* gamma, const float* beta, float p, uint64_t seed, uint64_t offset,
                       long rows, float* out, float* zhat, float* rstd, void* stream) {
    if (!y || !gamma || !beta || !out) return fail(AMUSE_EINVAL, "amuse_train_ln_fwd: y, gamma, beta, out must be given");
    if (rows < 1) return fail(AMUSE_EINVAL, "rows must be >= 1, got %ld", rows);
    uint32_t thr; float scale;
    if (int e = drop_args(p, &thr, &scale)) return e;
    const long g = (rows + 7) / 8;
    hipLaunchKernelGGL(k_train_ln_fwd, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, (hipStream_t)stream, x, y, bias, gamma, beta, thr, scale, seed, offset, rows,
                       out, zhat, rstd);
    HIP_TRY(hipGetLastError());
    return 0;
}

int amuse_train_ln_bwd(const float* dout, const float* dout2, const float* zhat, const float* rstd, const float* gamma, float p, uint64_t seed, uint64_t offset, long rows,
                       float* dx, float* dy, float* dgamma, float* dbeta, float* dbias, float* ws, void* stream) {
    if (!dout || !zhat || !rstd || !gamma || !dy || !ws) return fail(AMUSE_EINVAL, "amuse_train_ln_bwd: NULL argument");
    if (rows < 1) return fail(AMUSE_EINVAL, "rows must be >= 1, got %ld", rows);
    uint32_t thr; float scale;
    if (int e = drop_args(p, &thr, &scale)) return e;
    const int g = grid_for(rows, 64);
    hipLaunchKernelGGL(k_train_ln_bwd, dim3(g), dim3(kTrainThreads), 0, (hipStream_t)stream, dout, dout2, zhat, rstd, gamma, thr, scale, seed, offset, rows, dx, dy, ws);
    hipLaunchKernelGGL(k_train_finalize, dim3(1), dim3(kTrainThreads), 0, (hipStream_t)stream, ws, g, dgamma, dbeta, dbias, 384, 128);
    HIP_TRY(hipGetLastError());
    return 0;
}

int amuse_train_bias_gelu_drop_fwd(const float* h, const float* b, float p, uint64_t seed, uint64_t offset, long rows, int F, float* out, void* stream) {
    if (!h || !b || !out) return fail(AMUSE_EINVAL, "amuse_train_bias_gelu_drop_fwd: NULL argument");
    if (rows < 1 || F < 4 || F > 1024 || (F & 3)) return fail(AMUSE_EINVAL, "rows %ld / F %d: F must be a multiple of 4 up to 1024", rows, F);
    uint32_t thr; float scale;
    if (int e = drop_args(p, &thr, &scale)) return e;
    const size_t n4 = (size_t)rows * (F / 4);
    const size_t g = (n4 + 255) / 256;
    hipLaunchKernelGGL(k_train_bgd_fwd, dim3((unsigned)(g > 8192 ? 8192 : g)), dim3(256), 0, (hipStream_t)stream, h, b, thr, scale, seed, offset, n4, F / 4, out);
    HIP_TRY(hipGetLastError());
    return 0;
}

int amuse_train_bias_gelu_drop_bwd(const float* da, const float* h, const float* b, float p, uint64_t seed, uint64_t offset, long rows, int F, float* dh,
                                   float* db, float* ws, void* stream) {
    if (!da || !h || !b || !dh || !db || !ws) return fail(AMUSE_EINVAL, "amuse_train_bias_gelu_drop_bwd: NULL argument");
    if (rows < 1 || F < 4 || F > 1024 || (F & 3)) return fail(AMUSE_EINVAL, "rows %ld / F %d: F must be a multiple of 4 up to 1024", rows, F);
    uint32_t thr; float scale;
    if (int e = drop_args(p, &thr, &scale)) return e;
    const int g = grid_for(rows, 2 * (kTrainThreads / (F / 4)));
    hipLaunchKernelGGL(k_train_bgd_bwd, dim3(g), dim3(kTrainThreads), 0, (hipStream_t)stream, da, h, b, thr, scale, seed, offset, rows, F / 4, dh, ws);
    hipLaunchKernelGGL(k_train_finalize, dim3(1), dim3(kTrainThreads), 0, (hipStream_t)stream, ws, g, db, (float*)nullptr, (float*)nullptr, F, F);
    HIP_TRY(hipGetLastError());
    return 0;
}

int amuse_train_colsum(const float* x, long rows, int C, float* out, float* ws, void* stream) {
    if (!x || !out || !ws) return fail(AMUSE_EINVAL, "amuse_train_colsum: NULL argument");
    if (rows < 1 || C < 4 || C > 1024 || (C & 3)) return fail(AMUSE_EINVAL, "rows %ld / C %d: C must be a multiple of 4 up to 1024", rows, C);
    const int g = grid_for(rows, 2 * (kTrainThreads / (C / 4)));
    hipLaunchKernelGGL(k_train_colsum, dim3(g), dim3(kTrainThreads), 0, (hipStream_t)stream, x, rows, C / 4, ws);
    hipLaunchKernelGGL(k_train_finalize, dim3(1), dim3(kTrainThreads), 0, (hipStream_t)stream, ws, g, out, (float*)nullptr, (float*)nullptr, C, C);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // extern "C"
