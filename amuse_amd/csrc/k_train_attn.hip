// Self-attention of the train_gesture step's transformer layers, forward and backward, in fp32 on the matrix cores (v_mfma_f32_16x16x4_f32): the
// nn.MultiheadAttention core of utils/cross_attention.py:259-272,323-345 as the reference trains it - softmax(q k^T / sqrt(32)) with dropout on the
// probabilities, times v - for B x 4 heads of S <= 304 tokens (MotionPrior's 300 / 302, the Denoiser's 3..5), head width 32.
//
// Why its own kernels: the vendor's fused fp32 attention takes 39 us forward and 121 us backward per call at B = 32, S = 300 - 3.75 of the iteration's
// 17.8 ms of device time - and two host-side operator calls per layer and direction.  Here a (clip, head) is a workgroup's job and everything is a
// 16 x 16 tile product with operands already in MFMA layout:
//   * the "row image" R(X) of a [S][32] matrix keeps, per 16-row tile and 16-column half, lane (g, r) -> X[16 tile + r][16 half + 4 g + m] (m = 0..3: one
//     f32x4 = the four k-slots of four MFMAs) - the layout a global row-major ld4 delivers, and the library's row-lane layout (amuse_dev.hpp);
//   * the "transposed image" T(X) keeps lane (g, r) -> X[16 tile + 4 g + m][16 half + r]: the A operand of a product that contracts over ROWS.
//   Scores come out with the contraction-friendly layout for the next product (a tile's accumulator registers m = 0..3 are the k-slots of the following
//   MFMAs), so nothing is shuffled between the two GEMMs of the forward pass or the five of the backward pass.
//   * forward: K as R, V as T in LDS; a wave owns query tiles, walks the key tiles with an online softmax, keeps O^T in registers; writes O and the
//     row's log-sum-exp (log2 units, scale folded).
//   * backward, two roles in ONE launch (workgroups [0, B H): dQ; [B H, 2 B H): dK and dV - they run side by side on the chip):
//       dQ role: R(K), R(V), T(K) in LDS, a wave owns query tiles: S^T, P, dPd^T = V dO^T, dS, dQ^T += K^T dS^T;
//       dK/dV role: R(Q), T(Q), R(dO), T(dO) in LDS, a wave owns key tiles: S, P, dPd, dS; dV^T += dO^T Pd, dK^T += Q^T dS.
//     D_i = sum_d dO O (the softmax backward's row term) is recomputed by every workgroup in its prologue.
//   * dropout: keep(b, h, i, j) = hash32(seed, offset, element index) >= p 2^32 - one integer hash per element, the same in every layout (a counter-based
//     Philox draw serves four CONSECUTIVE elements, which the two backward roles index along different axes); O = (P . keep / (1 - p)) V with the softmax
//     denominator from the undropped P, as torch does.
// Numerics: true fp32 products and accumulation (the reference's arithmetic up to summation order); tests/test_gpu_train_ops.py holds outputs and all three
// gradients against torch's math attention (<= 2e-5 / 2e-4 . max) and the dropout path through a mask read back from the kernel.
#include <cstdlib>

#include "amuse_dev.hpp"
#include "amuse_host.hpp"

namespace amuse {
namespace {

constexpr int kAMaxTiles = 19;     // S <= 304
constexpr int kAImg = kAMaxTiles * 2 * 64;   // f32x4 per image (38,912 B)

__device__ __forceinline__ uint32_t hash32(uint32_t x) {   // lowbias32 (Chris Wellons): full avalanche in two multiplies
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// mask . 1 / (1 - p) of element (row i, key j) of (clip, head) bh; thr = p 2^32 (0: no dropout)
__device__ __forceinline__ float keep_scale(uint32_t key, uint32_t bh, int i, int j, uint32_t thr, float scale) {
    // (thr == 0, no dropout: every hash passes and scale is 1 - no special case, so that the callers' key loops stay free of branches)
    const uint32_t idx = (bh * 512u + (uint32_t)i) * 512u + (uint32_t)j;   // S <= 304 < 512
    return hash32(idx ^ key) >= thr ? scale : 0.f;
}

// stage rows [0, S) x 32 columns at `src` (row stride `ld` floats) as R / T images; rows >= S are zero.  All 512 threads of the workgroup; a thread's
// (up to five) loads are all in flight before its first LDS write.
__device__ __forceinline__ void stage_images(const float* src, int ld, int S, int ntiles, f32x4* R, f32x4* T) {
    constexpr int kMax = (kAMaxTiles * 16 * 8 + 511) / 512;
    const int n = ntiles * 16 * 8;
    f32x4 v[kMax];
#pragma unroll
    for (int k = 0; k < kMax; ++k) {
        const int i = threadIdx.x + 512 * k, row = i >> 3, c4 = i & 7;   // 4 consecutive columns 4 c4 ..
        v[k] = (i < n && row < S) ? ld4(src + (size_t)row * ld + 4 * c4) : splat4(0.f);
    }
#pragma unroll
    for (int k = 0; k < kMax; ++k) {
        const int i = threadIdx.x + 512 * k, row = i >> 3, c4 = i & 7;
        if (i >= n) break;
        const int tile = row >> 4, r = row & 15, half = c4 >> 2, g = c4 & 3;
        if (R) R[(tile * 2 + half) * 64 + 16 * g + r] = v[k];
        if (T) {   // element (row, col = 4 c4 + e) -> lane (g' = r / 4, r' = col % 16), slot m = r % 4
            float* t = reinterpret_cast<float*>(T + (tile * 2 + half) * 64 + 16 * (r >> 2)) + (r & 3);
#pragma unroll
            for (int e = 0; e < 4; ++e) t[(4 * g + e) * 4] = v[k][e];
        }
    }
}
// acc += A_image_tile . B (contraction over the 32 columns): A = R image tile `at`, B = row-lane registers b[2]
__device__ __forceinline__ f32x4 dot32(const f32x4* A, int at, const f32x4 (&b)[2], f32x4 acc, int lane) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const f32x4 a = A[(at * 2 + h) * 64 + lane];
#pragma unroll
        for (int m = 0; m < 4; ++m) acc = mfma_f32(a[m], b[h][m], acc);
    }
    return acc;
}
// out[half] += T_image_tile(half) . p (contraction over the tile's 16 rows, k-slots = p's registers)
__device__ __forceinline__ void dotrows(const f32x4* T, int tt, const f32x4& p, f32x4 (&out)[2], int lane) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const f32x4 a = T[(tt * 2 + h) * 64 + lane];
#pragma unroll
        for (int m = 0; m < 4; ++m) out[h] = mfma_f32(a[m], p[m], out[h]);
    }
}

struct AttnArgs {
    const float* qkv;   // [B S][384]
    float* o;           // [B S][128]
    float* lse;         // [B][4][S]  (log2 units)
    const float* dout;  // [B S][128]
    float* dqkv;        // [B S][384]
    float* mask_out;    // debugging: [B][4][S][S] keep . 1/(1-p), or null
    int B, S;
    uint32_t thr, key;
    float drop_scale;
    const uint32_t* epoch;   // the device-side dropout epoch (k_train.hip train_epoch_ptr): mixed into the key, fresh masks per replay of a captured step
};
constexpr float kScaleLog2 = 0.17677669529663687f * 1.44269504088896340736f;   // 1 / sqrt(32) . log2 e

// ---------------------------------------------------------------- forward: grid (B 4, query parts), eight waves
// MASK_OUT: the tests' instantiation that also writes the keep mask (a store behind a branch per element inside the key loop - kept out of the product's loop, whose
// MFMAs and softmax arithmetic then sit in ONE basic block for the scheduler)
template <bool MASK_OUT>
__global__ __launch_bounds__(512) void k_attn_fwd(AttnArgs a) {
    a.key ^= hash32(*a.epoch * 0x9E3779B9u + 0x85EBCA6Bu) * (*a.epoch != 0u);   // (epoch 0: the key as the host made it)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* RK = reinterpret_cast<f32x4*>(smem);
    f32x4* TV = RK + kAImg;
    const int S = a.S, nt = (S + 15) >> 4;
    const int bh = blockIdx.x, b = bh >> 2, h = bh & 3;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
    const float* base = a.qkv + (size_t)b * S * 384 + 32 * h;
    stage_images(base + 128, 384, S, nt, RK, nullptr);
    stage_images(base + 256, 384, S, nt, nullptr, TV);
    __syncthreads();
    const int nw = blockDim.x >> 6, nwq = nw * gridDim.y;           // query tiles are dealt round-robin over (part, wave)
    for (int it = blockIdx.y * nw + wave; it < nt; it += nwq) {
        const int qi = 16 * it + c;
        f32x4 q[2];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) q[hh] = qi < S ? ld4(base + (size_t)qi * 384 + 16 * hh + 4 * g) : splat4(0.f);
        float m_run = -INFINITY, l_run = 0.f;
        f32x4 o[2] = {splat4(0.f), splat4(0.f)};
        // the scores of key tile jt + 1 are issued (8 dependent MFMAs, ~290 cycles of the matrix pipe) BEFORE the softmax arithmetic of tile jt, which they run under;
        // the last iteration's look-ahead recomputes its own tile (no branch in the loop body)
        f32x4 s_next = dot32(RK, 0, q, splat4(0.f), lane);
        for (int jt = 0; jt < nt; ++jt) {
            f32x4 s = s_next * kScaleLog2;                                // lane (g, c): S[query c][keys 16 jt + 4 g + m], log2 units
            s_next = dot32(RK, min(jt + 1, nt - 1), q, splat4(0.f), lane);
            float mx = -INFINITY;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                if (16 * jt + 4 * g + m >= S) s[m] = -INFINITY;
                mx = fmaxf(mx, s[m]);
            }
            mx = allreduce_g_max(mx);
            const float m_new = fmaxf(m_run, mx);
            const float alpha = m_run == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(m_run - m_new);
            f32x4 p;
            float ps = 0.f;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                p[m] = __builtin_amdgcn_exp2f(s[m] - m_new);
                ps += p[m];
                const float ks = keep_scale(a.key, bh, qi, 16 * jt + 4 * g + m, a.thr, a.drop_scale);
                p[m] *= ks;
                if constexpr (MASK_OUT)
                    if (qi < S && 16 * jt + 4 * g + m < S) a.mask_out[((size_t)bh * S + qi) * S + 16 * jt + 4 * g + m] = ks;   // (tests)
            }
            ps = allreduce_g_sum(ps);
            l_run = l_run * alpha + ps;
            m_run = m_new;
            o[0] *= alpha;
            o[1] *= alpha;
            dotrows(TV, jt, p, o, lane);                             // O^T[d][query c] += sum_keys V[key][d] Pd[key]
        }
        if (qi < S) {
            const float inv = 1.0f / l_run;
            float* dst = a.o + ((size_t)b * S + qi) * 128 + 32 * h + 4 * g;
            st4(dst, o[0] * inv);
            st4(dst + 16, o[1] * inv);
            if (g == 0) a.lse[(size_t)bh * S + qi] = m_run + __builtin_amdgcn_logf(l_run);   // (v_log_f32 = log2)
        }
    }
}

// D[i] = sum_d dO[i][d] O[i][d] and the row's lse into LDS arrays (all rows of the (clip, head))
__device__ __forceinline__ void stage_rowterms(const AttnArgs& a, int b, int h, int bh, int S, float* Dl, float* Ll) {
    for (int i = threadIdx.x; i < 304; i += blockDim.x) {
        if (i >= S) {
            Dl[i] = 0.f;
            Ll[i] = 0.f;
            continue;
        }
        const float* po = a.o + ((size_t)b * S + i) * 128 + 32 * h;
        const float* pd = a.dout + ((size_t)b * S + i) * 128 + 32 * h;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const f32x4 x = ld4(po + 4 * k), y = ld4(pd + 4 * k);
            s += (x[0] * y[0] + x[1] * y[1]) + (x[2] * y[2] + x[3] * y[3]);
        }
        Dl[i] = s;
        Ll[i] = a.lse[(size_t)bh * S + i];
    }
}

// ---------------------------------------------------------------- backward: grid 2 B 4 workgroups of 512 threads
__global__ __launch_bounds__(512) void k_attn_bwd(AttnArgs a) {
    a.key ^= hash32(*a.epoch * 0x9E3779B9u + 0x85EBCA6Bu) * (*a.epoch != 0u);   // (epoch 0: the key as the host made it)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int S = a.S, nt = (S + 15) >> 4;
    const int nbh = a.B * 4;
    const bool role_q = (int)blockIdx.x < nbh;
    const int bh = role_q ? blockIdx.x : blockIdx.x - nbh, b = bh >> 2, h = bh & 3;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
    const float* base = a.qkv + (size_t)b * S * 384 + 32 * h;
    const float* dbase = a.dout + (size_t)b * S * 128 + 32 * h;
    f32x4* I0 = reinterpret_cast<f32x4*>(smem);
    f32x4* I1 = I0 + kAImg;
    f32x4* I2 = I1 + kAImg;
    f32x4* I3 = I2 + kAImg;
    float* Dl = reinterpret_cast<float*>(I3 + kAImg);
    float* Ll = Dl + 304;
    stage_rowterms(a, b, h, bh, S, Dl, Ll);
    if (role_q) {
        // ---- dQ: a wave owns query tiles; keys walk by.  I0 = R(K), I1 = R(V), I2 = T(K)
        stage_images(base + 128, 384, S, nt, I0, I2);
        stage_images(base + 256, 384, S, nt, I1, nullptr);
        __syncthreads();
        for (int it = wave; it < nt; it += 8) {
            const int qi = 16 * it + c;
            const bool qv = qi < S;
            f32x4 q[2], d_o[2];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                q[hh] = qv ? ld4(base + (size_t)qi * 384 + 16 * hh + 4 * g) : splat4(0.f);
                d_o[hh] = qv ? ld4(dbase + (size_t)qi * 128 + 16 * hh + 4 * g) : splat4(0.f);
            }
            const float lse = qv ? Ll[qi] : 0.f, Di = qv ? Dl[qi] : 0.f;
            f32x4 dq[2] = {splat4(0.f), splat4(0.f)};
            for (int jt = 0; jt < nt; ++jt) {
                const f32x4 s = dot32(I0, jt, q, splat4(0.f), lane) * kScaleLog2;
                const f32x4 dpd = dot32(I1, jt, d_o, splat4(0.f), lane);      // dPd^T[key][query c] = sum_d V[key][d] dO[c][d]
                f32x4 ds;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int kj = 16 * jt + 4 * g + m;
                    const float p = (kj < S && qv) ? __builtin_amdgcn_exp2f(s[m] - lse) : 0.f;
                    const float dp = dpd[m] * keep_scale(a.key, bh, qi, kj, a.thr, a.drop_scale);
                    ds[m] = p * (dp - Di) * 0.17677669529663687f;
                }
                dotrows(I2, jt, ds, dq, lane);                                 // dQ^T[d][c] += sum_keys K[key][d] dS[key]
            }
            if (qv) {
                float* dst = a.dqkv + ((size_t)b * S + qi) * 384 + 32 * h + 4 * g;
                st4(dst, dq[0]);
                st4(dst + 16, dq[1]);
            }
        }
    } else {
        // ---- dK, dV: a wave owns key tiles; queries walk by.  I0 = R(Q), I1 = T(Q), I2 = R(dO), I3 = T(dO)
        stage_images(base, 384, S, nt, I0, I1);
        stage_images(dbase, 128, S, nt, I2, I3);
        __syncthreads();
        for (int jt = wave; jt < nt; jt += 8) {
            const int kj = 16 * jt + c;
            const bool kv = kj < S;
            f32x4 k[2], v[2];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                k[hh] = kv ? ld4(base + 128 + (size_t)kj * 384 + 16 * hh + 4 * g) : splat4(0.f);
                v[hh] = kv ? ld4(base + 256 + (size_t)kj * 384 + 16 * hh + 4 * g) : splat4(0.f);
            }
            f32x4 dk[2] = {splat4(0.f), splat4(0.f)}, dv[2] = {splat4(0.f), splat4(0.f)};
            for (int it = 0; it < nt; ++it) {
                const f32x4 s = dot32(I0, it, k, splat4(0.f), lane) * kScaleLog2;   // lane (g, c): S[query 16 it + 4 g + m][key c]
                const f32x4 dpd = dot32(I2, it, v, splat4(0.f), lane);              // dPd[query][key c] = sum_d dO[query][d] V[c][d]
                const int q0 = 16 * it + 4 * g;
                const f32x4 lse = *reinterpret_cast<const f32x4*>(Ll + q0), Di = *reinterpret_cast<const f32x4*>(Dl + q0);
                f32x4 pd, ds;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const bool ok = kv && q0 + m < S;
                    const float p = ok ? __builtin_amdgcn_exp2f(s[m] - lse[m]) : 0.f;
                    const float ks = keep_scale(a.key, bh, q0 + m, kj, a.thr, a.drop_scale);
                    pd[m] = p * ks;
                    ds[m] = ok ? p * (dpd[m] * ks - Di[m]) * 0.17677669529663687f : 0.f;
                }
                dotrows(I3, it, pd, dv, lane);                                      // dV^T[d][key c] += sum_queries dO[query][d] Pd[query]
                dotrows(I1, it, ds, dk, lane);                                      // dK^T[d][key c] += sum_queries Q[query][d] dS[query]
            }
            if (kv) {
                float* dst = a.dqkv + ((size_t)b * S + kj) * 384 + 32 * h + 4 * g;
                st4(dst + 128, dk[0]);
                st4(dst + 128 + 16, dk[1]);
                st4(dst + 256, dv[0]);
                st4(dst + 256 + 16, dv[1]);
            }
        }
    }
}

constexpr int kFwdLds = 2 * kAImg * 16;
constexpr int kBwdLds = 4 * kAImg * 16 + 2 * 304 * 4;

int attn_args(AttnArgs* a, const float* qkv, float* o, float* lse, const float* dout, float* dqkv, int B, int S, float p, uint64_t seed, uint64_t offset) {
    if (!qkv || !o || !lse) return fail(AMUSE_EINVAL, "attention: NULL argument");
    if (B < 1 || S < 1 || S > 304) return fail(AMUSE_EINVAL, "attention: B %d, S %d (1..304)", B, S);
    if (!(p >= 0.f) || p >= 1.f) return fail(AMUSE_EINVAL, "dropout probability %g outside [0, 1)", (double)p);
    *a = AttnArgs{};
    a->qkv = qkv; a->o = o; a->lse = lse; a->dout = dout; a->dqkv = dqkv; a->B = B; a->S = S;
    a->thr = p > 0.f ? (uint32_t)((double)p * 4294967296.0) : 0u;
    a->drop_scale = 1.0f / (1.0f - p);
    uint64_t k = seed * 0x9E3779B97F4A7C15ull + offset * 0xD1B54A32D192ED03ull;
    k ^= k >> 29;
    a->key = (uint32_t)(k ^ (k >> 32));
    a->epoch = train_epoch_ptr();
    if (!a->epoch) return fail(AMUSE_EHIP, "attention: no device word for the dropout epoch");
    return 0;
}

}  // namespace
}  // namespace amuse

using namespace amuse;

extern "C" {

int amuse_train_attn_fwd(const float* qkv, int B, int S, float p, uint64_t seed, uint64_t offset, float* o, float* lse, float* mask_debug, void* stream) {
    AttnArgs a;
    if (int e = attn_args(&a, qkv, o, lse, nullptr, nullptr, B, S, p, seed, offset)) return e;
    a.mask_out = mask_debug;
    static DeviceOnce once;
    int dev_;
    if (!once.done(&dev_)) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attn_fwd<false>), hipFuncAttributeMaxDynamicSharedMemorySize, kFwdLds));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attn_bwd), hipFuncAttributeMaxDynamicSharedMemorySize, kBwdLds));
        once.set(dev_);
    }
    // two workgroups per (clip, head) while that fills the chip; eight waves each (two per SIMD: one's softmax under the other's MFMAs)
    const int parts = (S > 64 && B * 4 < 512) ? 2 : 1;
    if (mask_debug) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attn_fwd<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kFwdLds));
        hipLaunchKernelGGL(k_attn_fwd<true>, dim3(B * 4, parts), dim3(512), kFwdLds, (hipStream_t)stream, a);
    } else {
        hipLaunchKernelGGL(k_attn_fwd<false>, dim3(B * 4, parts), dim3(512), kFwdLds, (hipStream_t)stream, a);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

int amuse_train_attn_bwd(const float* qkv, const float* o, const float* lse, const float* dout, int B, int S, float p, uint64_t seed, uint64_t offset,
                         float* dqkv, void* stream) {
    AttnArgs a;
    if (!dout || !dqkv) return fail(AMUSE_EINVAL, "amuse_train_attn_bwd: NULL argument");
    if (int e = attn_args(&a, qkv, const_cast<float*>(o), const_cast<float*>(lse), dout, dqkv, B, S, p, seed, offset)) return e;
    static DeviceOnce once;
    int dev_;
    if (!once.done(&dev_)) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attn_fwd<false>), hipFuncAttributeMaxDynamicSharedMemorySize, kFwdLds));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attn_bwd), hipFuncAttributeMaxDynamicSharedMemorySize, kBwdLds));
        once.set(dev_);
    }
    hipLaunchKernelGGL(k_attn_bwd, dim3(2 * B * 4), dim3(512), kBwdLds, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // extern "C"
