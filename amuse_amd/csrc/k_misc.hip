// One-off prologue kernels: step-invariant work hoisted out of the reference's per-step
// Denoiser.forward (the reference recomputes all of this every step - SURVEY.md 8a rows A3-A5).
#include "amuse_dev.hpp"
#include "amuse_kernels.hpp"

namespace amuse {
namespace {

// Timesteps (sinusoid, flip_sin_to_cos) + TimestepEmbedding MLP + pe[1]   (embeddings.py:245-322)
// grid = T, block = 128.  Weights are passed transposed ([in][out]) so lane reads coalesce.
__global__ __launch_bounds__(128) void k_time_tokens(const int* __restrict__ ts, const float* __restrict__ freqs,
                                                     const float* __restrict__ w1t, const float* __restrict__ b1,
                                                     const float* __restrict__ w2t, const float* __restrict__ b2,
                                                     const float* __restrict__ pe1, float* __restrict__ out) {
    __shared__ float in[kCond];
    __shared__ float hid[kD];
    const int j = threadIdx.x;
    const float ang = (float)ts[blockIdx.x] * freqs[j];
    in[j] = cosf(ang);        // [cos | sin]: flip_sin_to_cos = true
    in[kD + j] = sinf(ang);
    __syncthreads();
    float acc = 0.f;
    for (int k = 0; k < kCond; ++k) acc = fmaf(in[k], w1t[k * kD + j], acc);
    acc += b1[j];
    hid[j] = acc / (1.0f + expf(-acc));  // SiLU
    __syncthreads();
    float o = 0.f;
    for (int k = 0; k < kD; ++k) o = fmaf(hid[k], w2t[k * kD + j], o);
    out[(size_t)blockIdx.x * kD + j] = (o + b2[j]) + pe1[j];
}

// emb_proj_{con,emo,sty}: Linear(ReLU(z)) + pe[2 + n]   (denoiser.py:74-79,153-181).  grid = B, block = 128
__global__ __launch_bounds__(128) void k_cond_tokens(CondArgs a) {
    __shared__ float z[kCond];
    const int j = threadIdx.x, b = blockIdx.x;
    for (int n = 0; n < a.ncond; ++n) {
        __syncthreads();
        z[j] = fmaxf(a.z[n][(size_t)b * kCond + j], 0.f);
        z[j + kD] = fmaxf(a.z[n][(size_t)b * kCond + kD + j], 0.f);
        __syncthreads();
        float acc = 0.f;
        const float* wt = a.wt[n];
        for (int k = 0; k < kCond; ++k) acc = fmaf(z[k], wt[k * kD + j], acc);
        a.out[((size_t)b * a.ncond + n) * kD + j] = (acc + a.bias[n][j]) + a.pe[(size_t)(a.pe_base + n) * kD + j];
    }
}

__global__ __launch_bounds__(64) void k_counter_normal(uint64_t seed, uint64_t clip0, int B, int step, int rng_stream,
                                                       float* out, int nq) {
    const size_t idx = (size_t)blockIdx.x * 64 + threadIdx.x;  // one Philox call per 4 features
    if (idx >= (size_t)B * nq) return;
    const int b = (int)(idx / nq), q = (int)(idx - (size_t)b * nq);
    st4(out + ((size_t)b * nq + q) * 4, counter_normal4(seed, clip0 + (uint64_t)b, (uint32_t)step, (uint32_t)q, (uint32_t)rng_stream));
}

// K / V of the trans_dec variants' memory tokens for all nine layers (cross_attention.py:331-336: key = value = memory; the
// in_proj rows 128..255 / 256..383 of multihead_attn).  grid = (N, 18), block = 128; weights transposed [l][k|v][in][out].
__global__ __launch_bounds__(128) void k_mem_kv(const float* __restrict__ tok, const float* __restrict__ wkv_t,
                                                const float* __restrict__ bkv, float* __restrict__ kv) {
    __shared__ float ts[kD];
    const int j = threadIdx.x, n = blockIdx.x, lk = blockIdx.y;
    ts[j] = tok[(size_t)n * kD + j];
    __syncthreads();
    const float* w = wkv_t + (size_t)lk * kD * kD;
    float acc = 0.f;
    for (int k = 0; k < kD; ++k) acc = fmaf(ts[k], w[k * kD + j], acc);
    kv[((size_t)n * 2 * kLayers + lk) * kD + j] = acc + bkv[lk * kD + j];
}

// infer_ldm.py:168-173 on a feature sequence: one thread per (row, joint) + one per row for the translation
__global__ __launch_bounds__(256) void k_feats_to_smplx(const float* __restrict__ feats, size_t nrows, int quat_mode,
                                                        float* __restrict__ poses, float* __restrict__ trans) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nrows * (kJoints + 1)) return;
    const size_t row = i / (kJoints + 1);
    const int jn = (int)(i - row * (kJoints + 1));
    const float* src = feats + row * kFeats;
    if (jn == kJoints) {
        if (trans) { trans[row * 3] = src[330]; trans[row * 3 + 1] = src[331]; trans[row * 3 + 2] = src[332]; }
        return;
    }
    if (!poses) return;
    float d6[6], aa[3];
    for (int e = 0; e < 6; ++e) d6[e] = src[6 * jn + e];
    rot6d_to_axis_angle(d6, quat_mode, aa);
    float* dst = poses + (row * kJoints + jn) * 3;
    dst[0] = aa[0]; dst[1] = aa[1]; dst[2] = aa[2];
}

// cross-attention onto a one-token memory: softmax over a single key == 1, so the layer adds
// out_proj(v_proj(z)) to every row (cross_attention.py:331-336).  grid = (9, B), block = 128.
__global__ __launch_bounds__(128) void k_vae_ca(const float* __restrict__ z, const float* __restrict__ wv_t,
                                                const float* __restrict__ bv, const float* __restrict__ wo_t,
                                                const float* __restrict__ bo, float* __restrict__ ca) {
    __shared__ float zs[kD];
    __shared__ float vs[kD];
    const int j = threadIdx.x, blk = blockIdx.x, b = blockIdx.y;
    zs[j] = z[(size_t)b * kD + j];
    __syncthreads();
    float acc = 0.f;
    const float* wv = wv_t + (size_t)blk * kD * kD;
    for (int k = 0; k < kD; ++k) acc = fmaf(zs[k], wv[k * kD + j], acc);
    vs[j] = acc + bv[blk * kD + j];
    __syncthreads();
    float o = 0.f;
    const float* wo = wo_t + (size_t)blk * kD * kD;
    for (int k = 0; k < kD; ++k) o = fmaf(vs[k], wo[k * kD + j], o);
    ca[((size_t)b * kLayers + blk) * kD + j] = o + bo[blk * kD + j];
}

// SMPL-X axis-angle + translation -> the prior's 333 motion features (infer_ldm.py:459-464):
// axis_angle_to_matrix (= axis_angle_to_quaternion -> quaternion_to_matrix, rotation_conversions.py:425-478, 41-71)
// then matrix_to_rotation_6d = the first two matrix rows (rotation_conversions.py:536-551).  One thread per (row, joint).
__global__ __launch_bounds__(256) void k_smplx_to_feats(const float* __restrict__ poses, const float* __restrict__ trans,
                                                        size_t nrows, float* __restrict__ feats) {
#pragma clang fp contract(off)
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nrows * (kJoints + 1)) return;
    const size_t row = i / (kJoints + 1);
    const int jn = (int)(i - row * (kJoints + 1));
    float* dst = feats + row * kFeats;
    if (jn == kJoints) {
        const float* t = trans + row * 3;
        dst[330] = t[0]; dst[331] = t[1]; dst[332] = t[2];
        return;
    }
    const float* aa = poses + (row * kJoints + jn) * 3;
    const float ax = aa[0], ay = aa[1], az = aa[2];
    const float ang = sqrtf(ax * ax + ay * ay + az * az);
    const float half = 0.5f * ang;
    const float s = (fabsf(ang) < 1e-6f) ? (0.5f - (ang * ang) / 48.0f) : (sinf(half) / ang);
    const float r = cosf(half), qi = ax * s, qj = ay * s, qk = az * s;
    const float two_s = 2.0f / (r * r + qi * qi + qj * qj + qk * qk);
    dst += 6 * jn;
    dst[0] = 1.0f - two_s * (qj * qj + qk * qk);
    dst[1] = two_s * (qi * qj - qk * r);
    dst[2] = two_s * (qi * qk + qj * r);
    dst[3] = two_s * (qi * qj + qk * r);
    dst[4] = 1.0f - two_s * (qi * qi + qk * qk);
    dst[5] = two_s * (qj * qk - qi * r);
}

// DDPMScheduler.add_noise (diffusers 0.17.1; call site ldm.py:84): per-clip coefficients from the host table
__global__ __launch_bounds__(128) void k_add_noise(const float* __restrict__ z0, const float* __restrict__ noise,
                                                   const float* __restrict__ sa, const float* __restrict__ sb, float* out, int nfeat) {
#pragma clang fp contract(off)
    for (int f = threadIdx.x; f < nfeat; f += 128) {
        const size_t i = (size_t)blockIdx.x * nfeat + f;
        out[i] = sa[blockIdx.x] * z0[i] + sb[blockIdx.x] * noise[i];
    }
}

// MotionPrior.encode tail (vae.py:203-213): mu = dist[0], logvar = dist[1]; std = logvar.exp().pow(0.5);
// latent = Normal(mu, std).rsample() = mu + std * eps with eps supplied by the caller (or latent = mu when absent)
__global__ __launch_bounds__(128) void k_vae_latent(const float* __restrict__ stats, const float* __restrict__ eps,
                                                    float* mu, float* sd, float* latent) {
#pragma clang fp contract(off)  // mu + std * eps as two rounded operations, like the reference's tensor ops
    const int b = blockIdx.x, j = threadIdx.x;
    const float m = stats[((size_t)b * 2 + 0) * kD + j];
    const float s = sqrtf(expf(stats[((size_t)b * 2 + 1) * kD + j]));
    if (mu) mu[(size_t)b * kD + j] = m;
    if (sd) sd[(size_t)b * kD + j] = s;
    if (latent) latent[(size_t)b * kD + j] = eps ? m + s * eps[(size_t)b * kD + j] : m;
}

}  // namespace

hipError_t launch_time_tokens(const int* timesteps_dev, int T, const float* freqs, const float* w1t, const float* b1,
                              const float* w2t, const float* b2, const float* pe1, float* out, hipStream_t stream) {
    hipLaunchKernelGGL(k_time_tokens, dim3(T), dim3(128), 0, stream, timesteps_dev, freqs, w1t, b1, w2t, b2, pe1, out);
    return hipGetLastError();
}

hipError_t launch_cond_tokens(const CondArgs& a, hipStream_t stream) {
    hipLaunchKernelGGL(k_cond_tokens, dim3(a.B), dim3(128), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_counter_normal(uint64_t seed, uint64_t clip0, int B, int step, int rng_stream, float* out,
                                 hipStream_t stream, int nfeat) {
    const int nq = nfeat / 4;
    hipLaunchKernelGGL(k_counter_normal, dim3((unsigned)(((size_t)B * nq + 63) / 64)), dim3(64), 0, stream, seed, clip0, B, step,
                       rng_stream, out, nq);
    return hipGetLastError();
}

hipError_t launch_mem_kv(const float* tok, int N, const float* wkv_t, const float* bkv, float* kv, hipStream_t stream) {
    hipLaunchKernelGGL(k_mem_kv, dim3(N, 2 * kLayers), dim3(128), 0, stream, tok, wkv_t, bkv, kv);
    return hipGetLastError();
}

hipError_t launch_feats_to_smplx(const float* feats, size_t nrows, int quat_mode, float* poses, float* trans, hipStream_t stream) {
    const size_t n = nrows * (kJoints + 1);
    hipLaunchKernelGGL(k_feats_to_smplx, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, feats, nrows, quat_mode, poses, trans);
    return hipGetLastError();
}

hipError_t launch_vae_ca(const float* z, const float* wv_t, const float* bv, const float* wo_t, const float* bo,
                         float* ca, int B, hipStream_t stream) {
    hipLaunchKernelGGL(k_vae_ca, dim3(kLayers, B), dim3(128), 0, stream, z, wv_t, bv, wo_t, bo, ca);
    return hipGetLastError();
}

hipError_t launch_smplx_to_feats(const float* poses, const float* trans, size_t nrows, float* feats, hipStream_t stream) {
    const size_t n = nrows * (kJoints + 1);
    hipLaunchKernelGGL(k_smplx_to_feats, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, poses, trans, nrows, feats);
    return hipGetLastError();
}

// amuse_update_weights_device: a packed image (weight stream, parameter vector, transposed matrix) is a GATHER of the model's
// parameters - map[j] = 1 + index of the parameter that image element j holds, 0 = padding
__global__ __launch_bounds__(256) void k_repack(const float* __restrict__ params, const int* __restrict__ map, void* __restrict__ dst, size_t n, int kind) {
    for (size_t j = (size_t)blockIdx.x * 256 + threadIdx.x; j < n; j += (size_t)gridDim.x * 256) {
        const int m = map[j];
        const float v = m ? params[m - 1] : 0.f;
        if (kind == 1) {
            typedef __bf16 bf;
            reinterpret_cast<unsigned short*>(dst)[j] = __builtin_bit_cast(unsigned short, (bf)v);   // round-to-nearest-even, as the host packer
        } else if (kind == 3) {   // fp16 image (AMUSE_PREC_F16)
            reinterpret_cast<unsigned short*>(dst)[j] = __builtin_bit_cast(unsigned short, (_Float16)v);
        } else if (kind == 2) {
            // split-fp16 image: 1 KiB units (512 elements) alternate hi / lo pieces of the same weights (amuse_api.hip pack_gemm)
            const _Float16 hi = (_Float16)v;
            const _Float16 out = ((j >> 9) & 1) ? (_Float16)(v - (float)hi) : hi;
            reinterpret_cast<unsigned short*>(dst)[j] = __builtin_bit_cast(unsigned short, out);
        } else {
            reinterpret_cast<float*>(dst)[j] = v;
        }
    }
}
hipError_t launch_repack(const float* params, const int* map, void* dst, size_t n, int kind, hipStream_t stream) {
    const size_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(k_repack, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, stream, params, map, dst, n, kind);
    return hipGetLastError();
}

hipError_t launch_add_noise(const float* z0, const float* noise, const float* sa, const float* sb, float* out, int B,
                            hipStream_t stream, int nfeat) {
    hipLaunchKernelGGL(k_add_noise, dim3(B), dim3(128), 0, stream, z0, noise, sa, sb, out, nfeat);
    return hipGetLastError();
}

hipError_t launch_vae_latent(const float* stats, const float* eps, float* mu, float* std, float* latent, int B,
                             hipStream_t stream) {
    hipLaunchKernelGGL(k_vae_latent, dim3(B), dim3(128), 0, stream, stats, eps, mu, std, latent);
    return hipGetLastError();
}

}  // namespace amuse
