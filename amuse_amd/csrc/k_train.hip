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

#include <algorithm>
#include <mutex>
#include <type_traits>

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
                                                      uint64_t seed, uint64_t offset, const uint32_t* __restrict__ epoch, long rows, float* __restrict__ out, float* __restrict__ zhat,
                                                      float* __restrict__ rstd_out) {
    offset += (uint64_t)*epoch << 32;   // the device-side dropout epoch (train_epoch_ptr): fresh masks per replay of a captured step
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

__global__ __launch_bounds__(kTrainThreads) void k_train_ln_bwd(const float* dout, const float* dout2, const float* __restrict__ zhat, const float* __restrict__ rstd_in,
                                                      const float* __restrict__ gamma, uint32_t thr, float scale, uint64_t seed, uint64_t offset, const uint32_t* __restrict__ epoch, long rows,
                                                      float* dx, float* dy, float* ws) {
    // (dx may be dout, dy may be dout2: same thread, same elements)
    offset += (uint64_t)*epoch << 32;   // the device-side dropout epoch (train_epoch_ptr): fresh masks per replay of a captured step
    __shared__ f32x4 red[3 * kTrainThreads];
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
    // the workgroup's three partial rows at once: threads 0..31 add up sg's 32 row lanes, 32..63 sb's, 64..95 sy's - each in row-lane order, as workgroup_partial does
    // (one barrier and one pass of 32 dependent LDS reads instead of three of each)
    red[threadIdx.x] = sg;
    red[kTrainThreads + threadIdx.x] = sb;
    red[2 * kTrainThreads + threadIdx.x] = sy;
    __syncthreads();
    if (threadIdx.x < 96) {
        const int which = threadIdx.x >> 5, c = threadIdx.x & 31;
        const f32x4* src = red + which * kTrainThreads;
        f32x4 sum = src[c];
        for (int q = 1; q < 32; ++q) sum += src[q * 32 + c];
        st4(ws + (size_t)blockIdx.x * 384 + which * 128 + 4 * c, sum);
    }
}
__global__ __launch_bounds__(kTrainThreads) void k_train_finalize(const float* ws, int n, float* o0, float* o1, float* o2, int ncol, int C) {
    __shared__ f32x4 red[kTrainThreads];
    float* outs[3] = {o0, o1, o2};
    sum_partials(ws, n, outs, ncol, C, red);
}
// several reductions' partial sums (each in its own region of the workspace) in ONE launch behind a layer's backward pass: workgroup = job
struct FinJob {
    const float* ws;
    float* o[3];
    int n, ncol, C;
};
// a weight gradient's chunk partials (k_train_wgrad below) waiting to be added up: workgroups blk0 .. of the layer's one summing launch
struct WsumJob {
    const float* part;
    float* out;
    int chunks;
    unsigned n4, blk0;
};
// Every reduction a layer's backward pass leaves behind - the column sums of its norms / biases AND the chunk sums of its weight gradients - added up by ONE
// launch at the layer's end (k_train_layer_sums: workgroup = a column-sum job, or 256 float4 columns of a weight gradient) instead of 6-8 launches of ~5 us
// spread through the layer.  The arithmetic of each sum is what the stand-alone launches do (k_train_finalize, k_train_wgrad_sum): same order, same bits.
struct LayerSums {
    FinJob j[8];
    WsumJob w[6];
    int nj, nw;
    unsigned blocks;   // of the weight-gradient jobs so far
};
__device__ __forceinline__ void wgrad_sum_256(const float* __restrict__ part, int chunks, size_t n4, size_t i, int c, int q, f32x4* red, float* __restrict__ out);
__global__ __launch_bounds__(kTrainThreads) void k_train_layer_sums(LayerSums s) {
    __shared__ f32x4 red[kTrainThreads];
    if ((int)blockIdx.x < s.nj) {
        const FinJob& J = s.j[blockIdx.x];
        float* outs[3] = {J.o[0], J.o[1], J.o[2]};
        sum_partials(J.ws, J.n, outs, J.ncol, J.C, red);
        return;
    }
    const unsigned b = blockIdx.x - s.nj;
    int k = 0;
    for (int i = 1; i < s.nw; ++i)
        if (b >= s.w[i].blk0) k = i;
    const WsumJob W = s.w[k];
    // four groups of 256 threads, each what one workgroup of k_train_wgrad_sum is: 64 float4 columns
    const int sub = threadIdx.x >> 8, t = threadIdx.x & 255;
    wgrad_sum_256(W.part, W.chunks, W.n4, ((size_t)(b - W.blk0) * 4 + sub) * 64 + (t & 63), t & 63, t >> 6, red + sub * 192, W.out);
}
// the layer whose backward pass is being issued by this thread (amuse_train_layer_bwd): its reductions are parked here instead of being launched one by one
thread_local LayerSums* g_layer_sums = nullptr;
thread_local size_t g_layer_ws_used = 0;   // floats of the weight-gradient workspace holding parked partials

// ---- AdamW over a contiguous range of the trainer's flat parameter / gradient buffers (torch.optim.AdamW, amsgrad off: decoupled weight decay, then
// p -= lr / (1 - b1^t) . m / (sqrt(v) / sqrt(1 - b2^t) + eps)); bc1 = 1 - b1^t and rbc2 = 1 / sqrt(1 - b2^t) come from the host
__global__ __launch_bounds__(256) void k_train_adamw(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t n,
                                                     float step_size, float decay, float omb1, float b2, float omb2, float eps, float rbc2) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float gi = g[i];
        const float mi = m[i] + (gi - m[i]) * omb1;
        const float vi = b2 * v[i] + omb2 * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] = p[i] * decay - step_size * (mi / (sqrtf(vi) * rbc2 + eps));
    }
}

// The same update with the step count on the DEVICE (a training step captured as a HIP graph: the host's count is frozen at capture): k_train_adamw_scalars
// advances *step and leaves {lr / (1 - b1^t), 1 / sqrt(1 - b2^t)} - computed in double like the host path - in scal[0..1]; the update kernel reads them.
__global__ void k_train_adamw_scalars(long* step, double lr, double b1, double b2, float* scal) {
    const long t = ++*step;
    scal[0] = (float)(lr / (1.0 - pow(b1, (double)t)));
    scal[1] = (float)(1.0 / sqrt(1.0 - pow(b2, (double)t)));
}
__global__ __launch_bounds__(256) void k_train_adamw_dev(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t n,
                                                         const float* __restrict__ scal, float decay, float omb1, float b2, float omb2, float eps) {
    const float step_size = scal[0], rbc2 = scal[1];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float gi = g[i];
        const float mi = m[i] + (gi - m[i]) * omb1;
        const float vi = b2 * v[i] + omb2 * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] = p[i] * decay - step_size * (mi / (sqrtf(vi) * rbc2 + eps));
    }
}

// ---- FFN activation: a = dropout(gelu(h + b)), exact erf (F.gelu's default)
__device__ __forceinline__ float gelu_exact(float u) { return 0.5f * u * (1.0f + erff(u * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad(float u) {
    return 0.5f * (1.0f + erff(u * 0.70710678118654752440f)) + u * 0.39894228040143267794f * expf(-0.5f * u * u);
}
__global__ __launch_bounds__(256) void k_train_bgd_fwd(const float* __restrict__ h, const float* __restrict__ b, uint32_t thr, float scale, uint64_t seed,
                                                       uint64_t offset, const uint32_t* __restrict__ epoch, size_t n4, int F4, float* __restrict__ out) {
    offset += (uint64_t)*epoch << 32;   // the device-side dropout epoch (train_epoch_ptr): fresh masks per replay of a captured step
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const f32x4 u = ld4(h + 4 * i) + ld4(b + 4 * (i % F4));
        const f32x4 g = f32x4{gelu_exact(u[0]), gelu_exact(u[1]), gelu_exact(u[2]), gelu_exact(u[3])};
        st4(out + 4 * i, g * drop_scale4(seed, offset, i, thr, scale));
    }
}
// column sums ride along: a thread owns one 4-column group (F4 <= 256 groups) for all its rows - 1,024 / F4 row lanes per workgroup
__global__ __launch_bounds__(kTrainThreads) void k_train_bgd_bwd(const float* __restrict__ da, const float* __restrict__ h, const float* __restrict__ b, uint32_t thr,
                                                       float scale, uint64_t seed, uint64_t offset, const uint32_t* __restrict__ epoch, long rows, int F4, float* __restrict__ dh, float* ws) {
    offset += (uint64_t)*epoch << 32;   // the device-side dropout epoch (train_epoch_ptr): fresh masks per replay of a captured step
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

// ---- small pieces of the layer-level entry points
// out[r][:] = bias (the C operand of a GEMM with beta = 1: a Linear's bias)
__global__ __launch_bounds__(256) void k_train_bias_rows(const float* __restrict__ bias, size_t n4, int C4, float* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) st4(out + 4 * i, ld4(bias + 4 * (i % C4)));
}
// The decoder's cross-attention onto its ONE memory token (cross_attention.py:331-337; nn_modules.mha_one_key): the attention weight of every
// (clip, query, head) is 1, hit by the attention dropout: vk[b][s][:] = c[b][:] . keep(b, s, head) / (1 - p), c = the memory's value projection.
// keep(b, s, h) = draw h % 4 of Philox(seed, counter ((b S + s) H + h) / 4, offset).  D = 128, heads of 32 features: one thread = 4 features.
__device__ __forceinline__ float head_keep(uint64_t seed, uint64_t offset, size_t row, int H, int h, uint32_t thr, float scale) {
    if (thr == 0) return 1.0f;
    const size_t e = row * H + h;
    const uint4 b = drop_bits(seed, offset, e >> 2);
    const uint32_t d = (e & 3) == 0 ? b.x : (e & 3) == 1 ? b.y : (e & 3) == 2 ? b.z : b.w;
    return (d >> 8) >= thr ? scale : 0.f;
}
__global__ __launch_bounds__(256) void k_train_vk(const float* __restrict__ c, uint32_t thr, float scale, uint64_t seed, uint64_t offset, const uint32_t* __restrict__ epoch, long rows, int S, int H,
                                                  float* __restrict__ vk) {
    offset += (uint64_t)*epoch << 32;   // the device-side dropout epoch (train_epoch_ptr): fresh masks per replay of a captured step
    const int dh4 = 128 / H / 4;   // float4 groups per head
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)rows * 32; i += (size_t)gridDim.x * 256) {
        const size_t row = i >> 5;
        const int cg = (int)(i & 31);
        st4(vk + 4 * i, ld4(c + (row / S) * 128 + 4 * cg) * head_keep(seed, offset, row, H, cg / dh4, thr, scale));
    }
}
// dc[b][:] = sum_s dvk[b][s][:] . keep(b, s, head) / (1 - p): one workgroup per clip, 32 column groups x 8 row lanes
__global__ __launch_bounds__(256) void k_train_dc(const float* __restrict__ dvk, uint32_t thr, float scale, uint64_t seed, uint64_t offset, const uint32_t* __restrict__ epoch, int S, int H,
                                                  float* __restrict__ dc) {
    offset += (uint64_t)*epoch << 32;   // the device-side dropout epoch (train_epoch_ptr): fresh masks per replay of a captured step
    __shared__ f32x4 red[256];
    const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5, dh4 = 128 / H / 4;
    const size_t row0 = (size_t)blockIdx.x * S;
    f32x4 acc = splat4(0.f);
    for (int s0 = rl; s0 < S; s0 += 8) acc += ld4(dvk + (row0 + s0) * 128 + 4 * cg) * head_keep(seed, offset, row0 + s0, H, cg / dh4, thr, scale);
    red[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < 32) {
        f32x4 t = red[threadIdx.x];
        for (int q = 1; q < 8; ++q) t += red[q * 32 + threadIdx.x];
        st4(dc + (size_t)blockIdx.x * 128 + 4 * threadIdx.x, t);
    }
}

int drop_args(float p, uint32_t* thr, float* scale) {
    if (!(p >= 0.f) || p >= 1.f) return fail(AMUSE_EINVAL, "dropout probability %g outside [0, 1)", (double)p);
    *thr = (uint32_t)(p * 16777216.0f);
    *scale = 1.0f / (1.0f - p);
    return 0;
}
// ---- the dropout epoch: ONE device word per GPU that every mask-drawing kernel adds into the high half of its Philox counter's offset.  It stays 0 in eager
// training (the host hands every call a fresh offset); a training step captured as a HIP graph replays with FROZEN host offsets, so the graph ends with
// k_train_epoch_add and every replay draws fresh masks (amuse_train_epoch_advance; forward and backward of one step see the same value).
uint32_t* g_train_epoch[64] = {};
__global__ void k_train_epoch_set(uint32_t* e, uint32_t add, uint32_t set, int do_set) { *e = do_set ? set : *e + add; }
}  // namespace
uint32_t* train_epoch_ptr() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    dev &= 63;
    if (!g_train_epoch[dev]) {
        if (hipMalloc((void**)&g_train_epoch[dev], 256) != hipSuccess) return nullptr;
        if (hipMemset(g_train_epoch[dev], 0, 256) != hipSuccess) return nullptr;
    }
    return g_train_epoch[dev];
}
namespace {
int grid_for(long rows, int rows_per_wg) {
    const long g = (rows + rows_per_wg - 1) / rows_per_wg;
    return (int)(g < 1 ? 1 : g > kTrainWgs ? kTrainWgs : g);
}

// ---- the layer's plain GEMMs run on the library's own fp32-MFMA kernels (k_train_gemm.hip: the tall projections; the generic kernel for every other shape) and on the
// chunked weight-gradient kernel below: no vendor BLAS anywhere in the library (rounds 2-5 sent the 333-wide, 32-row and 160-row shapes to a dlopen'ed rocBLAS).
// blas_handle keeps its name from then: it records the stream of the call for rm_gemm.
hipStream_t g_blas_stream[64] = {};
int blas_handle(hipStream_t st, void** h) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    g_blas_stream[dev & 63] = st;
    *h = nullptr;
    return 0;
}
// ---- weight gradients: dW[M][N] = dy^T x over `rows` (dy [rows][M], x [rows][N], row-major).  The outputs are small (128 x 128 .. 512 x 128) and the reduction long
// (9,600 rows): rocBLAS runs them on 9-27 workgroups with a 3-way global split - ~25 us each, 91 of them per iteration (profiles/r04_train_step_torch_profile.txt).
// Here the ROWS are cut into chunks so that ~1,000 waves share the work: a wave owns a 32 x 32 block of dW for one chunk, v_mfma_f32_16x16x4_f32 with both operands
// read straight from the row-major arrays - lane (g, r) of a k-step of 4 rows loads dy[k + g][m0 + 2 r .. + 1] and x[k + g][n0 + 2 r .. + 1] (two 8-byte loads, the 16
// lanes of a row group contiguous), i.e. MFMA tile t holds the block's rows m0 + 2 i + t (columns likewise): no transpose, no LDS.  The chunks' partial blocks go to
// a workspace and a second kernel adds them IN ORDER (deterministic, like the step's other reductions).
constexpr int kWgradMaxChunks = 64;
constexpr int kWgradRows = 192;                       // rows per chunk (48 k-steps of the fp32 MFMA; 304-row chunks - one wave per SIMD at 9,600 rows - measured slower: 22.9 against 21.9 us;
                                                      // so did a workgroup's four waves taking four consecutive chunks of ONE block and adding them up through LDS - a quarter of the partials, but no
                                                      // operand shared between the waves any more: 26.8 us)
constexpr size_t kWgradWsFloats = (size_t)16 << 20;   // 64 MB of partial blocks per device: every weight gradient of one layer (10.7 M floats at 9,600 rows) until the layer's summing launch
// STREAM = false: up to ~1,000 waves (one per SIMD); true: more - half of the chunk's loads in flight (~130 registers: three waves per SIMD), every k-step's registers
// reloaded with the k-step 24 further on as soon as its MFMAs are issued.  MT = MFMA tiles of the wave's block along m: 2 (32 x 32 per wave, 64 x 64 per workgroup) or
// 4 (64 x 32 per wave, 128 x 64 per workgroup: half the waves and half the x loads per MFMA for the 512-wide gradients; the m side is then read 16 bytes per lane)
template <bool STREAM, int MT>
__global__ __launch_bounds__(256) void k_train_wgrad(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ part, int rows, int M, int N) {
    typedef float avec __attribute__((ext_vector_type(MT)));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, r = lane & 15;
    const int nbn = N >> 6, bm = blockIdx.x / nbn, bn = blockIdx.x - bm * nbn;
    const int m0 = bm * (32 * MT) + (wave >> 1) * (16 * MT), n0 = (bn << 6) + ((wave & 1) << 5);
    int k0 = blockIdx.y * kWgradRows, k1 = min(rows, k0 + kWgradRows);   // (rows is a multiple of 4); the workgroup's chunks blockIdx.y, blockIdx.y + gridDim.y, ... (one up to 12,288 rows)
    f32x4 acc[MT][2];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t][0] = acc[t][1] = splat4(0.f);
    // A wave is alone on its SIMD (~800 waves per launch), so its loads are its only latency hiding: ALL 96 of the chunk's loads go out before the first
    // MFMA (192 registers) and the MFMAs follow the data in as it arrives - one memory round trip per launch.  Rows past the chunk's end load a valid row
    // and count as zero.
    constexpr int NK = kWgradRows / 4, NL = STREAM ? NK / 2 : NK;
    // fp32-input MFMAs hide no VALU work (profiles/r06_mfma_valu_samewave_probe.txt: every vector instruction beside them adds its ~6 cycles), and per-lane 64-bit
    // addresses + the row clamp + the dead-row selects were ~9 vector instructions per k-step beside its 4-8 MFMAs (+30-40 % of the launch).  A FULL chunk (all
    // 192 rows inside: every chunk at 9,600 rows) therefore reads through a wave-uniform row pointer + ONE per-lane offset (scalar address arithmetic, the loads'
    // saddr form) and multiplies unconditionally; only a ragged last chunk takes the clamped / selected path.
    const unsigned la = 4u * (unsigned)(g * M + m0 + MT * r), lb = 4u * (unsigned)(g * N + n0 + 2 * r);   // bytes; unsigned 32-bit: the loads' scalar-base + vector-offset form
    __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dy + (size_t)k0 * M), 0, 0x7fffffff, 0x00020000);
    __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + (size_t)k0 * N), 0, 0x7fffffff, 0x00020000);
    auto run = [&](auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        avec a[NL];
        float2 b[NL];
        auto fetch = [&](int slot, int u) {
            if constexpr (FULL) {
                // buffer loads: descriptor = the chunk's first row (wave-uniform), the k-step in the SCALAR offset, the lane's part in the one vector offset
                if constexpr (MT == 4) a[slot] = __builtin_bit_cast(avec, __builtin_amdgcn_raw_buffer_load_b128(ra, (int)la, (int)((unsigned)(16 * u) * (unsigned)M), 0));
                else a[slot] = __builtin_bit_cast(avec, __builtin_amdgcn_raw_buffer_load_b64(ra, (int)la, (int)((unsigned)(16 * u) * (unsigned)M), 0));
                b[slot] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rb, (int)lb, (int)((unsigned)(16 * u) * (unsigned)N), 0));
            } else {
                const int rc = min(k0 + 4 * u + g, rows - 1);
                a[slot] = *reinterpret_cast<const avec*>(dy + (size_t)rc * M + m0 + MT * r);
                b[slot] = *reinterpret_cast<const float2*>(x + (size_t)rc * N + n0 + 2 * r);
            }
        };
        auto multiply = [&](int slot, int u) {
            const bool dead = !FULL && k0 + 4 * u + g >= k1;
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const float at = dead ? 0.f : a[slot][t];
                acc[t][0] = mfma_f32(at, b[slot].x, acc[t][0]);
                acc[t][1] = mfma_f32(at, b[slot].y, acc[t][1]);
            }
        };
#pragma unroll
        for (int u = 0; u < NL; ++u) fetch(u, u);
        __builtin_amdgcn_sched_barrier(0);   // (hipcc otherwise interleaves loads and MFMAs to save registers: 42 instead of ~230, one round trip per k-step)
        if constexpr (STREAM) {
#pragma unroll
            for (int u = 0; u < NL; ++u) {
                multiply(u, u);
                fetch(u, u + NL);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int u = 0; u < NL; ++u) multiply(u, u + NL);
        } else {
#pragma unroll
            for (int u = 0; u < NK; ++u) multiply(u, u);
        }
    };
    for (;;) {
        if (k0 + kWgradRows <= rows) run(std::true_type{});
        else run(std::false_type{});
        k0 += gridDim.y * kWgradRows;      // more than 64 chunks (batches beyond 12,288 rows): the workgroup goes on with chunk + 64, ... into the same accumulators
        if (k0 >= rows) break;
        k1 = min(rows, k0 + kWgradRows);
        ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dy + (size_t)k0 * M), 0, 0x7fffffff, 0x00020000);
        rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + (size_t)k0 * N), 0, 0x7fffffff, 0x00020000);
    }
    // C fragment: lane (g, r), element v = tile row 4 g + v, tile column r  ->  dW row m0 + MT (4 g + v) + t, columns n0 + 2 r + {0, 1}
    float* o = part + (size_t)blockIdx.y * M * N;
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int v = 0; v < 4; ++v)
            *reinterpret_cast<float2*>(o + (size_t)(m0 + MT * (4 * g + v) + t) * N + n0 + 2 * r) = float2{acc[t][0][v], acc[t][1][v]};
}
// out = the chunks' partial blocks added up in a FIXED order: thread (column c of 64 float4 columns, group q of 4) adds chunks q, q + 4, ... (their loads in
// flight together), the four groups' sums are added q = 0..3 through LDS (red: 3 x 64 float4).  i = the thread's float4 column (whole groups of 64 are in or out)
__device__ __forceinline__ void wgrad_sum_256(const float* __restrict__ part, int chunks, size_t n4, size_t i, int c, int q, f32x4* red, float* __restrict__ out) {
    const bool in = i < n4;
    f32x4 v[kWgradMaxChunks / 4];
#pragma unroll
    for (int j = 0; j < kWgradMaxChunks / 4; ++j) {
        const int ch = q + 4 * j;
        v[j] = (in && ch < chunks) ? ld4(part + ((size_t)ch * n4 + i) * 4) : splat4(0.f);
    }
    f32x4 s = v[0];
#pragma unroll
    for (int j = 1; j < kWgradMaxChunks / 4; ++j) s += v[j];
    if (q > 0) red[(q - 1) * 64 + c] = s;
    __syncthreads();
    if (q == 0 && in) st4(out + 4 * i, ((s + red[c]) + red[64 + c]) + red[128 + c]);
}
__global__ __launch_bounds__(256) void k_train_wgrad_sum(const float* __restrict__ part, int chunks, size_t n4, float* __restrict__ out) {
    __shared__ f32x4 red[3 * 64];
    const int c = threadIdx.x & 63;
    wgrad_sum_256(part, chunks, n4, (size_t)blockIdx.x * 64 + c, c, threadIdx.x >> 6, red, out);   // (n4 is a multiple of 64)
}
// Two chains of layers may be in flight on two streams of a device (the trainer runs the Denoiser's forward / backward pass beside the prior's): every scratch buffer
// of this file and of k_train_gemm.hip exists once per LANE, a small integer the calling thread sets in front of its calls (amuse_train_set_lane; 0 unless told otherwise)
constexpr int kTrainLanes = 2;
thread_local int g_train_lane = 0;
float* g_wgrad_ws[64][kTrainLanes] = {};
// a stream of the library's own per device and lane for the decoder layers' memory-token branch (amuse_train_layer_bwd), with the one fork / join event pair it needs;
// created on the first (eager) call, so that a later graph capture of the step finds them
struct MemSide {
    hipStream_t st;
    hipEvent_t fork, join;
    bool ok;
};
MemSide g_mem_side[64][kTrainLanes] = {};
std::mutex g_mem_side_mu;
int mem_side(MemSide** out) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_mem_side_mu);
    MemSide& m = g_mem_side[dev & 63][g_train_lane];
    if (!m.ok) {
        HIP_TRY(hipStreamCreateWithFlags(&m.st, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&m.fork, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&m.join, hipEventDisableTiming));
        m.ok = true;
    }
    *out = &m;
    return 0;
}
// gradients of at least this many elements go to the generic kernel (k_train_gemm.hip)
constexpr long wgrad_max_elems() { return 131072L; }
int wgrad_launch(const float* dy, const float* x, float* out, long rows, long M, long N, hipStream_t st) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    dev &= 63;
    const int lane = g_train_lane;
    if (!g_wgrad_ws[dev][lane]) HIP_TRY(hipMalloc((void**)&g_wgrad_ws[dev][lane], kWgradWsFloats * sizeof(float)));
    const long chunks = std::min<long>((rows + kWgradRows - 1) / kWgradRows, kWgradMaxChunks);   // partial blocks (a workgroup takes every 64th chunk beyond that)
    const bool wide = M * N >= 65536 && !(M & 127);
    const long blocks = wide ? (M >> 7) * (N >> 6) : (M >> 6) * (N >> 6);
    // partials: behind the ones a layer has parked (g_layer_ws_used; 0 outside amuse_train_layer_bwd)
    const size_t need = (size_t)chunks * M * N;
    if (chunks > 1 && g_layer_ws_used + need > kWgradWsFloats) return fail(AMUSE_EINVAL, "weight-gradient workspace: %zu + %zu floats of %zu", g_layer_ws_used, need, kWgradWsFloats);
    float* part = g_wgrad_ws[dev][lane] + g_layer_ws_used;
    float* dst = chunks == 1 ? out : part;
    const dim3 grid((unsigned)blocks, (unsigned)chunks);
    if (wide) hipLaunchKernelGGL((k_train_wgrad<true, 4>), grid, dim3(256), 0, st, dy, x, dst, (int)rows, (int)M, (int)N);
    else if (blocks * chunks > 256) hipLaunchKernelGGL((k_train_wgrad<true, 2>), grid, dim3(256), 0, st, dy, x, dst, (int)rows, (int)M, (int)N);
    else hipLaunchKernelGGL((k_train_wgrad<false, 2>), grid, dim3(256), 0, st, dy, x, dst, (int)rows, (int)M, (int)N);
    if (chunks > 1) {
        const size_t n4 = (size_t)M * N / 4;
        LayerSums* S = g_layer_sums;
        if (S && S->nw < 6) {   // inside a layer's backward pass: added up by the layer's one summing launch
            S->w[S->nw++] = WsumJob{part, out, (int)chunks, (unsigned)n4, S->blocks};
            S->blocks += (unsigned)((n4 + 255) / 256);
            g_layer_ws_used += need;
        } else {
            hipLaunchKernelGGL(k_train_wgrad_sum, dim3((unsigned)(n4 / 64)), dim3(256), 0, st, part, (int)chunks, n4, out);
        }
    }
    return 0;
}
// out[M][N] (+)= op(a) . op(b); a is [M][K] (ta: [K][M]), b is [K][N] (tb: [N][K]), all row-major and dense
int bias_rows_launch(const float* bias, long rows, int C, float* out, hipStream_t st);
// bias (nullable; [N]): added to every row of the product - by the own kernel's epilogue, or as out's initial contents in front of the library call
int rm_gemm(void* h, bool ta, bool tb, long M, long N, long K, const float* a, const float* b, float* out, bool accumulate, const float* bias = nullptr) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    // (the own kernel copies 16 bytes per lane with LDS-DMA and stores pairs: operands on 16-byte, out / bias on 8-byte boundaries - the Denoiser's weights, views into the
    // trainer's flat parameter buffer behind the prior's 333-element bias, sit on 4-byte ones)
    const bool aligned = !(((uintptr_t)a | (uintptr_t)b) & 15) && !(((uintptr_t)out | (uintptr_t)bias) & 7);
    if (!ta && aligned && train_gemm_tall_takes(M, N, K, tb, bias != nullptr)) {
        HIP_TRY(launch_train_gemm_tall(a, b, bias, out, M, N, K, tb, accumulate, g_blas_stream[dev & 63]));
        return 0;
    }
    if (bias && accumulate) return fail(AMUSE_EINVAL, "rm_gemm: bias and accumulate together");
    // (measured, profiles/r04_train_wgrad_kernel_ab.txt: 12 us against 25 for a 128 x 128 gradient over 9,664 rows, 21 against 26 for 384 x 128; the FFN's 512 x 128 - bound by
    // the fp32 MFMA rate and their 13 MB of partial blocks - 25 against 27.5 on the 64 x 32-per-wave instantiation)
    if (ta && !tb && !accumulate && K >= 1024 && !(K & 3) && !(M & 63) && !(N & 63) && M * N < wgrad_max_elems() &&
        (size_t)(M * N) * std::min<long>((K + kWgradRows - 1) / kWgradRows, kWgradMaxChunks) <= kWgradWsFloats && !bias) {   // a weight gradient with a long reduction
        return wgrad_launch(a, b, out, K, M, N, g_blas_stream[dev & 63]);
    }
    HIP_TRY(launch_train_gemm_any(a, b, bias, out, M, N, K, ta, tb, accumulate, g_blas_stream[dev & 63]));   // every other shape: 333-wide, 32 / 160 rows, short or wide gradients
    return 0;
}
#define TRY(expr) do { if (int e_ = (expr)) return e_; } while (0)

int ln_fwd_launch(const float* x, const float* y, const float* bias, const float* gamma, const float* beta, uint32_t thr, float scale, uint64_t seed, uint64_t off,
                  long rows, float* out, float* zhat, float* rstd, hipStream_t st) {
    const long g = (rows + 7) / 8;
    hipLaunchKernelGGL(k_train_ln_fwd, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, st, x, y, bias, gamma, beta, thr, scale, seed, off, train_epoch_ptr(), rows, out, zhat, rstd);
    return 0;
}
// `job` given: the partial sums stay in `ws` (a region of their own) and the job is added up later by ONE k_train_finalize_jobs launch; else right away
int ln_bwd_launch(const float* dout, const float* dout2, const float* zhat, const float* rstd, const float* gamma, uint32_t thr, float scale, uint64_t seed, uint64_t off,
                  long rows, float* dx, float* dy, float* dgamma, float* dbeta, float* dbias, float* ws, hipStream_t st, FinJob* job = nullptr) {
    const int g = grid_for(rows, 64);
    hipLaunchKernelGGL(k_train_ln_bwd, dim3(g), dim3(kTrainThreads), 0, st, dout, dout2, zhat, rstd, gamma, thr, scale, seed, off, train_epoch_ptr(), rows, dx, dy, ws);
    if (job) *job = FinJob{ws, {dgamma, dbeta, dbias}, g, 384, 128};
    else hipLaunchKernelGGL(k_train_finalize, dim3(1), dim3(kTrainThreads), 0, st, ws, g, dgamma, dbeta, dbias, 384, 128);
    return 0;
}
int colsum_launch(const float* x, long rows, int C, float* out, float* ws, hipStream_t st, FinJob* job = nullptr) {
    const int g = grid_for(rows, 2 * (kTrainThreads / (C / 4)));
    hipLaunchKernelGGL(k_train_colsum, dim3(g), dim3(kTrainThreads), 0, st, x, rows, C / 4, ws);
    if (!job && g_layer_sums && g_layer_sums->nj < 8) job = &g_layer_sums->j[g_layer_sums->nj++];   // (a layer's backward pass: the caller gave a region of its own)
    if (job) *job = FinJob{ws, {out, nullptr, nullptr}, g, C, C};
    else hipLaunchKernelGGL(k_train_finalize, dim3(1), dim3(kTrainThreads), 0, st, ws, g, out, (float*)nullptr, (float*)nullptr, C, C);
    return 0;
}
// column sums for ANY width (the 333-wide output layer's bias gradient): block (column block of 64, row chunk) -> ws[chunk][C], then the chunks in order
__global__ __launch_bounds__(256) void k_train_colsum_any(const float* __restrict__ x, long rows, int C, int chunks, float* __restrict__ ws) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
    const long per = (rows + chunks - 1) / chunks, r0 = (long)blockIdx.y * per, r1 = r0 + per < rows ? r0 + per : rows;
    float s = 0.f;
    if (c < C)
        for (long r = r0 + q; r < r1; r += 4) s += x[(size_t)r * C + c];
    red[q][threadIdx.x & 63] = s;
    __syncthreads();
    if (q == 0 && c < C) ws[(size_t)blockIdx.y * C + c] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}
__global__ __launch_bounds__(256) void k_train_colsum_any_fin(const float* __restrict__ ws, int chunks, int C, float* __restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float s = ws[c];
    for (int k = 1; k < chunks; ++k) s += ws[(size_t)k * C + c];
    out[c] = s;
}
int colsum_any_launch(const float* x, long rows, int C, float* out, float* ws, hipStream_t st) {
    const int chunks = (int)(rows < 64 ? 1 : rows / 64 > kTrainWgs ? kTrainWgs : rows / 64);
    hipLaunchKernelGGL(k_train_colsum_any, dim3((unsigned)((C + 63) / 64), (unsigned)chunks), dim3(256), 0, st, x, rows, C, chunks, ws);
    hipLaunchKernelGGL(k_train_colsum_any_fin, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, st, ws, chunks, C, out);
    return 0;
}
constexpr size_t kWsRegion = (size_t)kTrainWgs * 1024;   // floats per reduction's partial sums
int bias_rows_launch(const float* bias, long rows, int C, float* out, hipStream_t st) {
    const size_t n4 = (size_t)rows * (C / 4), g = (n4 + 255) / 256;
    hipLaunchKernelGGL(k_train_bias_rows, dim3((unsigned)(g > 8192 ? 8192 : g)), dim3(256), 0, st, bias, n4, C / 4, out);
    return 0;
}

}  // namespace
int train_lane() { return g_train_lane; }   // (k_train_gemm.hip's split-k workspace)
}  // namespace amuse

using namespace amuse;

extern "C" {

int amuse_train_set_lane(int lane) {
    if (lane < 0 || lane >= kTrainLanes) return fail(AMUSE_EINVAL, "lane %d (0 .. %d)", lane, kTrainLanes - 1);
    g_train_lane = lane;
    return 0;
}

size_t amuse_train_ws_floats(void) { return 8 * kWsRegion; }   // 8 regions of [workgroups][up to 1,024 columns]: one per reduction of a layer's backward pass

int amuse_train_ln_fwd(const float* x, const float* y, const float* bias, const float* gamma, const float* beta, float p, uint64_t seed, uint64_t offset,
                       long rows, float* out, float* zhat, float* rstd, void* stream) {
    if (!y || !gamma || !beta || !out) return fail(AMUSE_EINVAL, "amuse_train_ln_fwd: y, gamma, beta, out must be given");
    if (rows < 1) return fail(AMUSE_EINVAL, "rows must be >= 1, got %ld", rows);
    uint32_t thr; float scale;
    if (int e = drop_args(p, &thr, &scale)) return e;
    const long g = (rows + 7) / 8;
    hipLaunchKernelGGL(k_train_ln_fwd, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, (hipStream_t)stream, x, y, bias, gamma, beta, thr, scale, seed, offset, train_epoch_ptr(), rows,
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
    hipLaunchKernelGGL(k_train_ln_bwd, dim3(g), dim3(kTrainThreads), 0, (hipStream_t)stream, dout, dout2, zhat, rstd, gamma, thr, scale, seed, offset, train_epoch_ptr(), rows, dx, dy, ws);
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
    hipLaunchKernelGGL(k_train_bgd_fwd, dim3((unsigned)(g > 8192 ? 8192 : g)), dim3(256), 0, (hipStream_t)stream, h, b, thr, scale, seed, offset, train_epoch_ptr(), n4, F / 4, out);
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
    hipLaunchKernelGGL(k_train_bgd_bwd, dim3(g), dim3(kTrainThreads), 0, (hipStream_t)stream, da, h, b, thr, scale, seed, offset, train_epoch_ptr(), rows, F / 4, dh, ws);
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

int amuse_train_adamw(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, double lr, double beta1, double beta2, double eps,
                      double weight_decay, long step, void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq) return fail(AMUSE_EINVAL, "amuse_train_adamw: NULL argument");
    if (n < 1 || step < 1) return fail(AMUSE_EINVAL, "n %zu, step %ld (1-based)", n, step);
    // (the scalars in double on the host, as torch's Python optimizer computes them: 1 - 0.999 in fp32 is 4.7e-5 off)
    const double bc1 = 1.0 - pow(beta1, (double)step), rbc2 = 1.0 / sqrt(1.0 - pow(beta2, (double)step));
    const size_t g = (n + 255) / 256;
    hipLaunchKernelGGL(k_train_adamw, dim3((unsigned)(g > 16384 ? 16384 : g)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n, (float)(lr / bc1),
                       (float)(1.0 - lr * weight_decay), (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, (float)rbc2);
    HIP_TRY(hipGetLastError());
    return 0;
}

int amuse_train_adamw_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, double lr, double beta1, double beta2, double eps,
                          double weight_decay, long* step_dev, float* scal_dev, int advance, void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || !step_dev || !scal_dev) return fail(AMUSE_EINVAL, "amuse_train_adamw_dev: NULL argument");
    if (n < 1) return fail(AMUSE_EINVAL, "n %zu", n);
    if (advance) hipLaunchKernelGGL(k_train_adamw_scalars, dim3(1), dim3(1), 0, (hipStream_t)stream, step_dev, lr, beta1, beta2, scal_dev);
    const size_t g = (n + 255) / 256;
    hipLaunchKernelGGL(k_train_adamw_dev, dim3((unsigned)(g > 16384 ? 16384 : g)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n, scal_dev,
                       (float)(1.0 - lr * weight_decay), (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps);
    HIP_TRY(hipGetLastError());
    return 0;
}

int amuse_train_epoch_advance(unsigned add, void* stream) {
    uint32_t* e = train_epoch_ptr();
    if (!e) return fail(AMUSE_EHIP, "no device word for the dropout epoch");
    hipLaunchKernelGGL(k_train_epoch_set, dim3(1), dim3(1), 0, (hipStream_t)stream, e, (uint32_t)add, 0u, 0);
    HIP_TRY(hipGetLastError());
    return 0;
}
int amuse_train_epoch_set(unsigned value, void* stream) {
    uint32_t* e = train_epoch_ptr();
    if (!e) return fail(AMUSE_EHIP, "no device word for the dropout epoch");
    hipLaunchKernelGGL(k_train_epoch_set, dim3(1), dim3(1), 0, (hipStream_t)stream, e, 0u, (uint32_t)value, 1);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---- layer-level entry points: everything of a transformer layer but its self-attention core in ONE call each way (amuse_hip.h amuse_train_layer)
int amuse_train_linear_fwd(const float* x, const float* W, const float* b, long rows, int K, int N, float* out, void* stream) {
    if (!x || !W || !out) return fail(AMUSE_EINVAL, "amuse_train_linear_fwd: NULL argument");
    if (rows < 1 || K < 1 || N < 1) return fail(AMUSE_EINVAL, "rows %ld, K %d, N %d", rows, K, N);
    hipStream_t st = (hipStream_t)stream;
    void* h;
    TRY(blas_handle(st, &h));
    TRY(rm_gemm(h, false, true, rows, N, K, x, W, out, false, b));
    HIP_TRY(hipGetLastError());
    return 0;
}

int amuse_train_linear_bwd(const float* dy, const float* x, const float* W, long rows, int K, int N, float* dW, float* db, float* dx, int accumulate_dx, float* ws,
                           void* stream) {
    if (!dy || !x || !W) return fail(AMUSE_EINVAL, "amuse_train_linear_bwd: NULL argument");
    if (rows < 1 || K < 1 || N < 1 || N > 1024) return fail(AMUSE_EINVAL, "rows %ld, K %d, N %d (up to 1024)", rows, K, N);
    if (db && !ws) return fail(AMUSE_EINVAL, "the bias gradient needs the workspace");
    hipStream_t st = (hipStream_t)stream;
    void* h;
    TRY(blas_handle(st, &h));
    if (dW) TRY(rm_gemm(h, true, false, N, K, rows, dy, x, dW, false));          // dW[N][K] = dy^T x
    if (db) { if (N & 3) colsum_any_launch(dy, rows, N, db, ws, st); else colsum_launch(dy, rows, N, db, ws, st); }
    if (dx) TRY(rm_gemm(h, false, false, rows, K, N, dy, W, dx, accumulate_dx != 0));   // dx[rows][K] (+)= dy W
    HIP_TRY(hipGetLastError());
    return 0;
}

static int layer_check(const amuse_train_layer* L, bool bwd) {
    if (!L) return fail(AMUSE_EINVAL, "layer is NULL");
    if (L->rows < 1 || L->B < 1 || L->S < 1 || (long)L->B * L->S != L->rows) return fail(AMUSE_EINVAL, "rows %ld != B %d x S %d", L->rows, L->B, L->S);
    if (L->H < 1 || 32 % L->H) return fail(AMUSE_EINVAL, "heads %d must divide 32 (128 features, whole float4 groups per head)", L->H);
    if (L->ff < 4 || L->ff > 1024 || (L->ff & 3)) return fail(AMUSE_EINVAL, "ff %d must be a multiple of 4 up to 1024", L->ff);
    if (!L->Wo || !L->g1 || !L->be1 || !L->W1 || !L->b1 || !L->W2 || !L->g3 || !L->be3 || !L->x || !L->o2 || !L->x1 || !L->h || !L->a || !L->out || !L->tmp)
        return fail(AMUSE_EINVAL, "amuse_train_layer: a required pointer is NULL");
    if (L->mem && (!L->Wv || !L->Wc || !L->g2 || !L->be2 || !L->c || !L->vk || !L->xm)) return fail(AMUSE_EINVAL, "amuse_train_layer: decoder pointers missing");
    if (bwd) {
        if (!L->dout || !L->dx || !L->do2 || !L->zh1 || !L->r1 || !L->zh3 || !L->r3 || !L->s128a || !L->s128b || !L->s512a || !L->s512b || !L->ws || !L->dWo || !L->dg1 ||
            !L->dbe1 || !L->dW1 || !L->db1 || !L->dW2 || !L->dg3 || !L->dbe3)
            return fail(AMUSE_EINVAL, "amuse_train_layer (backward): a required pointer is NULL");
        if (L->mem && (!L->zh2 || !L->r2 || !L->sdc || !L->dWv || !L->dbv || !L->dWc || !L->dg2 || !L->dbe2 || !L->dmem))
            return fail(AMUSE_EINVAL, "amuse_train_layer (backward): decoder pointers missing");
    }
    return 0;
}

int amuse_train_layer_fwd(const amuse_train_layer* L, void* stream) {
    TRY(layer_check(L, false));
    uint32_t thr, thr_a; float scale, scale_a;
    TRY(drop_args(L->p, &thr, &scale));
    TRY(drop_args(L->p_attn, &thr_a, &scale_a));
    hipStream_t st = (hipStream_t)stream;
    void* h;
    TRY(blas_handle(st, &h));
    const long rows = L->rows;
    if (L->Win) {   // the self-attention in the same call: packed in-projection, then the attention core into o2
        if (!L->qkv || !L->lse || L->H != 4) return fail(AMUSE_EINVAL, "amuse_train_layer: self-attention asked for without qkv / lse buffers, or heads != 4");
        TRY(amuse_train_linear_fwd(L->x, L->Win, L->bin, rows, 128, 384, L->qkv, stream));
        TRY(amuse_train_attn_fwd(L->qkv, L->B, L->S, L->p_attn, L->seed, L->off_self, const_cast<float*>(L->o2), L->lse, nullptr, stream));
    }
    // x1 = norm1(x + dropout1(o2 Wo^T + bo))
    TRY(rm_gemm(h, false, true, rows, 128, 128, L->o2, L->Wo, L->tmp, false));
    ln_fwd_launch(L->x, L->tmp, L->bo, L->g1, L->be1, thr, scale, L->seed, L->off[0], rows, L->x1, L->zh1, L->r1, st);
    const float* src = L->x1;
    if (L->mem) {   // xm = norm2(x1 + dropout2(vk Wc^T + bc)), vk = the memory token's value projection under the attention dropout
        TRY(rm_gemm(h, false, true, L->B, 128, 128, L->mem, L->Wv, L->c, false, L->bv));   // (the bias in the generic kernel's epilogue: one launch; product + bias = bias + product, the same bits as pre-filling c)
        const size_t n = (size_t)rows * 32, g = (n + 255) / 256;
        hipLaunchKernelGGL(k_train_vk, dim3((unsigned)(g > 8192 ? 8192 : g)), dim3(256), 0, st, L->c, thr_a, scale_a, L->seed, L->off[4], train_epoch_ptr(), rows, L->S, L->H, L->vk);
        TRY(rm_gemm(h, false, true, rows, 128, 128, L->vk, L->Wc, L->tmp, false));
        ln_fwd_launch(L->x1, L->tmp, L->bc, L->g2, L->be2, thr, scale, L->seed, L->off[1], rows, L->xm, L->zh2, L->r2, st);
        src = L->xm;
    }
    // out = norm3(src + dropout3(dropout(gelu(src W1^T + b1)) W2^T + b2))
    TRY(rm_gemm(h, false, true, rows, L->ff, 128, src, L->W1, L->h, false));
    {
        const size_t n4 = (size_t)rows * (L->ff / 4), g = (n4 + 255) / 256;
        hipLaunchKernelGGL(k_train_bgd_fwd, dim3((unsigned)(g > 8192 ? 8192 : g)), dim3(256), 0, st, L->h, L->b1, thr, scale, L->seed, L->off[2], train_epoch_ptr(), n4, L->ff / 4, L->a);
    }
    TRY(rm_gemm(h, false, true, rows, 128, L->ff, L->a, L->W2, L->tmp, false));
    ln_fwd_launch(src, L->tmp, L->b2, L->g3, L->be3, thr, scale, L->seed, L->off[3], rows, L->out, L->zh3, L->r3, st);
    HIP_TRY(hipGetLastError());
    return 0;
}

int amuse_train_layer_bwd(const amuse_train_layer* L, void* stream) {
    TRY(layer_check(L, true));
    uint32_t thr, thr_a; float scale, scale_a;
    TRY(drop_args(L->p, &thr, &scale));
    TRY(drop_args(L->p_attn, &thr_a, &scale_a));
    hipStream_t st = (hipStream_t)stream;
    void* h;
    TRY(blas_handle(st, &h));
    const long rows = L->rows;
    const int ff = L->ff;
    const float* src = L->mem ? L->xm : L->x1;
    // the layer's reductions (column sums of norms / biases, chunk sums of weight gradients) are parked in `sums` by the launches below and added up by ONE launch
    // at the end; nothing inside the layer reads a parameter gradient
    LayerSums sums{};
    struct Park {
        explicit Park(LayerSums* s) { g_layer_sums = s; g_layer_ws_used = 0; }
        ~Park() { g_layer_sums = nullptr; g_layer_ws_used = 0; }
    } park(&sums);
    int& nj = sums.nj;
    MemSide* side = nullptr;
    // FFN + last norm: s128a = d(src) through the norm, s128b = d(linear2 output)
    ln_bwd_launch(L->dout, nullptr, L->zh3, L->r3, L->g3, thr, scale, L->seed, L->off[3], rows, L->s128a, L->s128b, L->dg3, L->dbe3, L->db2, L->ws + nj * kWsRegion, st, &sums.j[nj]);
    ++nj;
    TRY(rm_gemm(h, true, false, 128, ff, rows, L->s128b, L->a, L->dW2, false));
    TRY(rm_gemm(h, false, false, rows, ff, 128, L->s128b, L->W2, L->s512a, false));
    {
        const int g = grid_for(rows, 2 * (kTrainThreads / (ff / 4)));
        float* wsj = L->ws + nj * kWsRegion;
        hipLaunchKernelGGL(k_train_bgd_bwd, dim3(g), dim3(kTrainThreads), 0, st, L->s512a, L->h, L->b1, thr, scale, L->seed, L->off[2], train_epoch_ptr(), rows, ff / 4, L->s512b, wsj);
        sums.j[nj++] = FinJob{wsj, {L->db1, nullptr, nullptr}, g, ff, ff};
    }
    TRY(rm_gemm(h, true, false, ff, 128, rows, L->s512b, src, L->dW1, false));
    TRY(rm_gemm(h, false, false, rows, 128, ff, L->s512b, L->W1, L->s128b, false));      // the FFN branch's gradient of src
    if (L->mem) {
        // cross-attention + norm2: s128a <- d(x1), s128b <- d(out_proj output)
        ln_bwd_launch(L->s128a, L->s128b, L->zh2, L->r2, L->g2, thr, scale, L->seed, L->off[1], rows, L->s128a, L->s128b, L->dg2, L->dbe2, L->dbc, L->ws + nj * kWsRegion, st,
                      &sums.j[nj]);
        ++nj;
        TRY(rm_gemm(h, true, false, 128, 128, rows, L->s128b, L->vk, L->dWc, false));
        // (The same fork in the FORWARD pass - vk produced on the side stream under the self-attention half - measured 10 % SLOWER: a cross-queue edge of a replayed graph
        // takes on the order of 100 us to propagate, harmless here where the join has a whole layer of slack, fatal there where it has 40 us.)
        // The memory token's branch - d(c) per clip, dWv, dbv, d(mem): four launches over 32 rows, 5-13 us each of mostly latency - feeds nothing inside the layer.  In
        // the tall layers it runs on a stream of the library's own beside the rest of the layer (forked here, joined in front of the layer's summing launch); d(vk) then
        // sits in the forward pass's scratch `tmp`, because `do2` is written again further down.
        if (rows >= 1024) TRY(mem_side(&side));
        float* dvk = side ? L->tmp : L->do2;
        TRY(rm_gemm(h, false, false, rows, 128, 128, L->s128b, L->Wc, dvk, false));     // d(vk)
        hipStream_t ms = st;
        if (side) {
            ms = side->st;
            HIP_TRY(hipEventRecord(side->fork, st));
            HIP_TRY(hipStreamWaitEvent(ms, side->fork, 0));
        }
        hipLaunchKernelGGL(k_train_dc, dim3(L->B), dim3(256), 0, ms, dvk, thr_a, scale_a, L->seed, L->off[4], train_epoch_ptr(), L->S, L->H, L->sdc);
        if (side) HIP_TRY(launch_train_gemm_any(L->sdc, L->mem, nullptr, L->dWv, 128, 128, L->B, true, false, false, ms));
        else TRY(rm_gemm(h, true, false, 128, 128, L->B, L->sdc, L->mem, L->dWv, false));
        colsum_launch(L->sdc, L->B, 128, L->dbv, L->ws + nj * kWsRegion, ms, &sums.j[nj]);
        ++nj;
        if (side) HIP_TRY(launch_train_gemm_any(L->sdc, L->Wv, nullptr, L->dmem, L->B, 128, 128, false, false, false, ms));
        else TRY(rm_gemm(h, false, false, L->B, 128, 128, L->sdc, L->Wv, L->dmem, false));
        ln_bwd_launch(L->s128a, nullptr, L->zh1, L->r1, L->g1, thr, scale, L->seed, L->off[0], rows, L->dx, L->s128b, L->dg1, L->dbe1, L->dbo, L->ws + nj * kWsRegion, st,
                      &sums.j[nj]);
    } else {
        ln_bwd_launch(L->s128a, L->s128b, L->zh1, L->r1, L->g1, thr, scale, L->seed, L->off[0], rows, L->dx, L->s128b, L->dg1, L->dbe1, L->dbo, L->ws + nj * kWsRegion, st,
                      &sums.j[nj]);
    }
    ++nj;
    // self-attention's out_proj
    TRY(rm_gemm(h, true, false, 128, 128, rows, L->s128b, L->o2, L->dWo, false));
    TRY(rm_gemm(h, false, false, rows, 128, 128, L->s128b, L->Wo, L->do2, false));
    if (L->Win) {   // the attention's backward pass and the in-projection's in the same call
        if (!L->qkv || !L->lse || !L->dqkv || !L->dWin) return fail(AMUSE_EINVAL, "amuse_train_layer (backward): self-attention buffers missing");
        TRY(amuse_train_attn_bwd(L->qkv, L->o2, L->lse, L->do2, L->B, L->S, L->p_attn, L->seed, L->off_self, L->dqkv, stream));
        TRY(amuse_train_linear_bwd(L->dqkv, L->x, L->Win, rows, 128, 384, L->dWin, L->dbin, L->dx, 1, L->ws + nj * kWsRegion, stream));
    }
    if (side) {   // join: the summing launch reads the branch's column sums, the caller d(mem)
        HIP_TRY(hipEventRecord(side->join, side->st));
        HIP_TRY(hipStreamWaitEvent(st, side->join, 0));
    }
    hipLaunchKernelGGL(k_train_layer_sums, dim3((unsigned)nj + sums.blocks), dim3(kTrainThreads), 0, st, sums);   // every column sum and weight gradient of the layer
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // extern "C"
