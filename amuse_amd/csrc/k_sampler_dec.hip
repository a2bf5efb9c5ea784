// Persistent sampling kernel of the Denoiser's arch = "trans_dec" variant with a latent sample
// (reference models/latent_diffusion/denoiser.py:116-131,190-199; TransformerDecoder / TransformerDecoderLayer.forward_post,
// utils/cross_attention.py:195-234,323-345): the whole T-step loop of infer_ldm.py:137-161 in one launch, like k_sampler.hip does
// for the shipped trans_enc configuration.
//
// Per step and clip:  tgt = x_t + query_pos.pe[0] (ONE token);  memory = [time(t), con, (emo), (sty)] + mem_pos.pe[0..3];
// nine layers of
//     x = norm1(x + out_proj(v_proj(x)))                      self-attention over a single key: softmax == 1, the value path only
//     x = norm2(x + out_proj2(softmax(q K^T / sqrt 32) V))    cross-attention of the token's four heads onto the 2..4 memory keys
//     x = norm3(x + linear2(gelu(linear1(x))))
// then decoder.norm and the scheduler update.  The memory tokens do not change through the layers and their K / V projections do not
// depend on x: K_l, V_l of the condition tokens are computed once per job, those of the time token once per schedule (MemKV,
// amuse_kernels.hpp; k_mem_kv in k_misc.hip) - the per-step network is the target token's path alone.
//
// Work decomposition (gfx950): a clip is ONE row, so a 16-row MFMA tile holds 16 clips (the trans_enc kernel: 3).  Workgroup = 4 waves,
// all holding the tile's [16 x 128] residual stream in the row-lane layout (amuse_dev.hpp).  Wave w owns head w: its 32 features
// of v (self-attention) and of q (cross-attention) are two output tiles of a full-K GEMM, and exactly the k-slice it contributes to
// the following out_proj - three split-K combines per layer (reduce-scatter + LayerNorm + all-gather, amuse_dev.hpp combine_rs), the
// third one the FFN's (hidden quarter w).  The 2..4-key attention runs on the VALU: a lane holds 8 of its row's 32 head features,
// scores are 8-term dot products + a four-lane butterfly.  Each wave streams its quarter of the weights (192 fp32 / 96 16-bit units
// per layer) through the 32-slot register ring, one sequential pass per step.
#include "amuse_dev.hpp"
#include "amuse_kernels.hpp"

namespace amuse {
namespace {

template <int PREC>
__device__ __forceinline__ void decoder_layer(f32x4 (&x)[kTiles], WRing<kRing>& rg, const float* pv, const MemKV& mem,
                                              const float* tkv, long clip_ld, int layer, char* comb, int wave, int lane) {
    const int g = lane >> 4;
    constexpr int U_P = gemm_units(PREC, 2, kTiles);      // a head's 2 output tiles, full K
    constexpr int U_O = gemm_units(PREC, kTiles, 2);      // out_proj k-slice of a head
    constexpr int U_F = gemm_units(PREC, kTiles, kTiles);
    constexpr int P_V = 0, P_O1 = (P_V + U_P) % kRing, P_Q = (P_O1 + U_O) % kRing, P_O2 = (P_Q + U_P) % kRing;
    constexpr int P_F1 = (P_O2 + U_O) % kRing, P_F2 = (P_F1 + U_F) % kRing;
    static_assert((P_F2 + U_F) % kRing == 0, "a layer must leave the ring at phase 0");
    constexpr bool FAST = is_op16(PREC);
    // K / V of this clip's memory keys for head `wave`, features 32 w + 16 o + 4 g + m: issued first, consumed after two GEMMs
    f32x4 mk[4][2], mv[4][2];
    const int nmem = 1 + mem.ncond;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (j < nmem) {
            const float* base = j == 0 ? tkv + (size_t)clip_ld * mem.tkv_clip_stride + (size_t)layer * 2 * kD
                                       : mem.ckv + (((size_t)clip_ld * mem.ncond + (j - 1)) * kLayers + layer) * 2 * kD;
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                mk[j][o] = ld4(base + 32 * wave + 16 * o + 4 * g);
                mv[j][o] = ld4(base + kD + 32 * wave + 16 * o + 4 * g);
            }
        } else {
            mk[j][0] = mk[j][1] = mv[j][0] = mv[j][1] = splat4(0.f);
        }
    }
    f32x4 part[kTiles];
    // ---- self-attention over the single target token (cross_attention.py:333-336 with one key): v_proj, out_proj, norm1
    {
        f32x4 v[2];
#pragma unroll
        for (int o = 0; o < 2; ++o) v[o] = splat4(0.f);
        gemm_ring<PREC, 2, kTiles, false, kRing, P_V>(v, x, rg);
#pragma unroll
        for (int o = 0; o < 2; ++o) v[o] += ld4(pv + PV_IN_B + 2 * kD + 16 * (2 * wave + o) + 4 * g);
#pragma unroll
        for (int t = 0; t < kTiles; ++t) part[t] = splat4(0.f);
        gemm_ring<PREC, kTiles, 2, false, kRing, P_O1>(part, v, rg);
        combine_rs<true, FAST>(part, x, true, pv + PV_OUT_B, pv + PV_LN1_W, pv + PV_LN1_B, comb, wave, lane);
    }
    // ---- cross-attention onto the memory tokens (cross_attention.py:337-343): q of head `wave`, 2..4 keys, out_proj, norm2
    {
        f32x4 q[2];
#pragma unroll
        for (int o = 0; o < 2; ++o) q[o] = splat4(0.f);
        gemm_ring<PREC, 2, kTiles, false, kRing, P_Q>(q, x, rg);
        const float scaling = 0.17677669529663687f;   // q * sqrt(1 / 32) (F.multi_head_attention_forward)
#pragma unroll
        for (int o = 0; o < 2; ++o) q[o] = (q[o] + ld4(pv + PVX_CQ_B + 16 * (2 * wave + o) + 4 * g)) * scaling;
        float s[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float d = 0.f;
#pragma unroll
            for (int o = 0; o < 2; ++o)
#pragma unroll
                for (int m = 0; m < 4; ++m) d = fmaf(q[o][m], mk[j][o][m], d);
            s[j] = allreduce_g_sum(d);
        }
        float mx = s[0];
#pragma unroll
        for (int j = 1; j < 4; ++j) mx = j < nmem ? fmaxf(mx, s[j]) : mx;
        float p[4], sum = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            p[j] = j < nmem ? (FAST ? __builtin_amdgcn_exp2f(1.44269504088896340736f * (s[j] - mx)) : expf(s[j] - mx)) : 0.f;
            sum += p[j];
        }
        f32x4 a[2] = {splat4(0.f), splat4(0.f)};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float pj = p[j] / sum;
#pragma unroll
            for (int o = 0; o < 2; ++o) a[o] += mv[j][o] * pj;
        }
#pragma unroll
        for (int t = 0; t < kTiles; ++t) part[t] = splat4(0.f);
        gemm_ring<PREC, kTiles, 2, false, kRing, P_O2>(part, a, rg);
        combine_rs<true, FAST>(part, x, true, pv + PVX_CO_B, pv + PV_LN2_W, pv + PV_LN2_B, comb, wave, lane);
    }
    // ---- FFN: hidden quarter `wave`, linear2 split-K over the quarters, norm3
    {
        f32x4 hid[kTiles];
#pragma unroll
        for (int t = 0; t < kTiles; ++t) hid[t] = ld4(pv + PV_L1_B + 16 * (kTiles * wave + t) + 4 * g);
        gemm_ring<PREC, kTiles, kTiles, false, kRing, P_F1>(hid, x, rg);
#pragma unroll
        for (int t = 0; t < kTiles; ++t) {
            if constexpr (is_op16(PREC)) {
                hid[t] = gelu_poly16<PREC>(hid[t]);
            } else {
#pragma unroll
                for (int m = 0; m < 4; ++m) hid[t][m] = gelu_erf(hid[t][m]);
            }
        }
#pragma unroll
        for (int t = 0; t < kTiles; ++t) part[t] = splat4(0.f);
        gemm_ring<PREC, kTiles, kTiles, false, kRing, P_F2>(part, hid, rg);
        combine_rs<true, FAST>(part, x, true, pv + PV_L2_B, pv + PV_LN3_W, pv + PV_LN3_B, comb, wave, lane);
    }
}

__device__ __forceinline__ void store_tap16(float* tap, int slot, const f32x4 (&x)[kTiles], int g, int r) {
#pragma unroll
    for (int t = 0; t < kTiles; ++t) st4(tap + ((size_t)slot * 16 + r) * kD + 16 * t + 4 * g, x[t]);
}

template <int PREC>
__global__ __launch_bounds__(256, 1) void k_sample_dec(SampleDecArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* comb = smem;                                                     // split-K combine buffers (kCombBytes)
    float* pvl = reinterpret_cast<float*>(smem + kSampleCombBytes);       // all small parameters (PVX_TOTAL floats)
    for (int i = threadIdx.x; i < PVX_TOTAL / 4; i += 256) st4(pvl + 4 * i, ld4(a.pvec + 4 * i));
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, r = lane & 15;
    const long clip = (long)blockIdx.x * 16 + r;
    const bool valid = clip < (long)a.B;
    const long clip_ld = valid ? clip : (long)a.B - 1;   // padding rows read a real clip's tables; they are never stored
    f32x4 lat[kTiles], pe[kTiles];
#pragma unroll
    for (int t = 0; t < kTiles; ++t) {
        pe[t] = ld4(a.pe0 + 16 * t + 4 * g);
        lat[t] = splat4(0.f);
        if (valid)
            lat[t] = a.x_init ? ld4(a.x_init + (size_t)clip * kD + 16 * t + 4 * g)
                              : counter_normal4(a.seed, a.clip0 + (uint64_t)clip, 0u, (uint32_t)(4 * t + g), 0u);
    }
    const bool tap = a.tap_out != nullptr && blockIdx.x == 0 && wave == 0;
    const uint4* wbase = a.wstream + (size_t)wave * (a.wave_units + kRing) * 64 + lane;
    WRing<kRing> rg;
    ring_fill(rg, wbase);
#pragma unroll 1
    for (int step = 0; step < a.T; ++step) {
        f32x4 x[kTiles];
#pragma unroll
        for (int t = 0; t < kTiles; ++t) x[t] = lat[t] + pe[t];   // query_pos(sample) (denoiser.py:193)
        if (tap && step == 0) store_tap16(a.tap_out, 0, x, g, r);
        rg.next = wbase + kRing * 64;  // the ring already holds units 0..R-1 of this step (stream tail = its head)
        const float* tkv = a.mem.tkv + (size_t)step * a.tkv_step_stride;
#pragma unroll 1
        for (int layer = 0; layer < kLayers; ++layer) {
            decoder_layer<PREC>(x, rg, pvl + layer * PVX_BLOCK, a.mem, tkv, clip_ld, layer, comb, wave, lane);
            if (tap && step == 0) store_tap16(a.tap_out, 1 + layer, x, g, r);
        }
        layer_norm_rows<is_op16(PREC)>(x, pvl + PVX_FINAL_W, pvl + PVX_FINAL_B, g);
        if (tap && step == 0) store_tap16(a.tap_out, 10, x, g, r);
        if (a.eps_out && valid && wave == 0 && step == a.T - 1) {
#pragma unroll
            for (int t = 0; t < kTiles; ++t) st4(a.eps_out + (size_t)clip * kD + 16 * t + 4 * g, x[t]);
        }
        // ---- scheduler.step (diffusers 0.17.1 DDIM / DDPM; amuse_hip.h amuse_schedule) - k_sampler.hip's update, one row per clip
        if (!a.no_update) {
            const float* cf = a.coef + (size_t)step * 8;
            const float sb = cf[0], sa = cf[1], c0 = cf[2], cx = cf[3], ce = cf[4], sg = cf[5], clipv = cf[6];
            f32x4 zt[kTiles];
#pragma unroll
            for (int t = 0; t < kTiles; ++t) {
                zt[t] = splat4(0.f);
                if (sg != 0.f && valid)
                    zt[t] = a.step_noise ? ld4(a.step_noise + ((size_t)step * a.B + clip) * kD + 16 * t + 4 * g)
                                         : counter_normal4(a.seed, a.clip0 + (uint64_t)clip, (uint32_t)step, (uint32_t)(4 * t + g), 1u);
            }
            constexpr bool FASTU = is_op16(PREC);  // 16-bit modes: reciprocal multiply instead of IEEE division
            const float inv_sa = 1.0f / sa;
            {
#pragma clang fp contract(off)
#pragma unroll
                for (int t = 0; t < kTiles; ++t) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        const float e = x[t][m], xl = lat[t][m];
                        const float num = __fsub_rn(xl, __fmul_rn(sb, e));
                        float x0 = FASTU ? num * inv_sa : __fdiv_rn(num, sa);
                        if (clipv > 0.f) x0 = fminf(fmaxf(x0, -clipv), clipv);
                        float nx = __fmul_rn(c0, x0);
                        if (cx != 0.f) nx = __fadd_rn(nx, __fmul_rn(cx, xl));
                        if (ce != 0.f) nx = __fadd_rn(nx, __fmul_rn(ce, e));
                        if (sg != 0.f) nx = __fadd_rn(nx, __fmul_rn(sg, zt[t][m]));
                        lat[t][m] = nx;
                    }
                }
            }
            if (a.traj_out && valid && wave == 0) {
#pragma unroll
                for (int t = 0; t < kTiles; ++t) st4(a.traj_out + ((size_t)step * a.B + clip) * kD + 16 * t + 4 * g, lat[t]);
            }
        }
    }
    if (a.latents_out && valid && wave == 0) {
#pragma unroll
        for (int t = 0; t < kTiles; ++t) st4(a.latents_out + (size_t)clip * kD + 16 * t + 4 * g, lat[t]);
    }
}

}  // namespace

hipError_t launch_sample_dec(const SampleDecArgs& a, int precision, hipStream_t stream) {
    const dim3 grid((a.B + 15) / 16), block(256);
    static DeviceOnce once;
    int dev_;
    if (!once.done(&dev_)) {
        const void* ks[4] = {reinterpret_cast<const void*>(&k_sample_dec<PREC_F32>), reinterpret_cast<const void*>(&k_sample_dec<PREC_BF16>),
                             reinterpret_cast<const void*>(&k_sample_dec<PREC_F16X2>), reinterpret_cast<const void*>(&k_sample_dec<PREC_F16>)};
        for (const void* k : ks) {
            hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, kSampleDecLdsBytes);
            if (e != hipSuccess) return e;
        }
        once.set(dev_);
    }
    if (precision == PREC_F32) hipLaunchKernelGGL((k_sample_dec<PREC_F32>), grid, block, kSampleDecLdsBytes, stream, a);
    else if (precision == PREC_F16X2) hipLaunchKernelGGL((k_sample_dec<PREC_F16X2>), grid, block, kSampleDecLdsBytes, stream, a);
    else if (precision == PREC_F16) hipLaunchKernelGGL((k_sample_dec<PREC_F16>), grid, block, kSampleDecLdsBytes, stream, a);
    else hipLaunchKernelGGL((k_sample_dec<PREC_BF16>), grid, block, kSampleDecLdsBytes, stream, a);
    return hipGetLastError();
}

}  // namespace amuse
