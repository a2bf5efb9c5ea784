// VAE decode: MotionPrior.decode (reference models/latent_diffusion/vae.py:216-278) =
// zeros(300,B,128) + learned PE -> SkipTransformerDecoder (cross_attention.py:89-125) of 9
// TransformerDecoderLayer.forward_post blocks (cross_attention.py:323-345) -> final_layer
// Linear(128 -> 333), followed by the 6D -> matrix -> axis-angle conversion of infer_ldm.py:168-173.
//
// Two kernels alternate (stream-ordered, activations stay in L2/MALL between them):
//   k_vae_rows  - everything that is independent per frame row, on 16-row tiles with the same
//                 4-wave split as the sampling kernel (heads / hidden-feature quarters, split-K
//                 combines through LDS):  [out_proj + LN1 + cross-attn constant + LN2 + FFN + LN3
//                 (+ skip push / skip linear)] of block i fused with the q,k,v projection of block
//                 i+1; the last stage fuses decoder.norm + final_layer + the rotation epilogue.
//   k_vae_attn  - the only S = 300 attention in the system: one workgroup per (clip, head), K_h and
//                 V_h staged once in LDS, flash-style online softmax, S^T = K.Q^T and O^T = V^T.P^T on
//                 MFMA so that the softmax runs along registers and O lands in row-lane layout.
// The one-token cross-attention collapses to a per-clip constant (k_vae_ca in k_misc.hip).
#include <cstdlib>
#include <type_traits>

#include "amuse_dev.hpp"
#include "amuse_kernels.hpp"

namespace amuse {
namespace {

constexpr int kRowTiles = 19;       // ceil(300 / 16)
constexpr int kFeatStride = 388;    // 384 padded features + 4: LDS row stride of the staged feats tile
// LDS: the split-K combine buffers (amuse_dev.hpp combine_rs, 41,472 B); the staged feature tile of the last stage
// (16 x 388 floats) reuses them - every combine is over by then.  Two workgroups fit a CU.
constexpr int kRowsLdsBytes = kCombBytes;
static_assert(16 * kFeatStride * 4 <= kCombBytes, "staged feature tile must fit the combine buffers");
constexpr int kVR = kVaeRing;       // weight-stream ring depth of k_vae_rows

// MODE M_DEC: MotionPrior.decode rows (S = 300).  M_ENC: MotionPrior.encode rows (vae.py:154-214): S = 302 =
// [2 distribution tokens | 300 embedded frames], TransformerEncoderLayer blocks (no cross-attention, two norms),
// stage 0 = skel_embedding + token concat + PE, last stage = encoder.norm of the two distribution rows only.
// M_DEN_E / M_DEN_D: ONE STEP of the Denoiser's diffusion_only variants (denoiser.py:64-66,174-204) on the same stages -
//   M_DEN_E (arch "trans_enc"): S = npre + 300 rows = [time, con, (emo), (sty) | pose_embd(x_t frames)] + query_pos, the M_ENC
//     stack (skip encoder, two norms per block) WITHOUT a key mask (denoiser.py:182 passes none);
//   M_DEN_D (arch "trans_dec"): S = 300 rows = pose_embd(x_t) + query_pos, nine plain TransformerDecoderLayer.forward_post blocks
//     (no skips): self-attention, cross-attention of every row onto the clip's 2..4 memory tokens (K / V hoisted: MemKV), FFN;
//   last stage = final norm + pose_proj (128 -> 333) + `sample[~mask.T] = 0` + (optionally) the scheduler update of x_t.
constexpr int M_DEC = 0, M_ENC = 1, M_DEN_E = 2, M_DEN_D = 3;
constexpr bool mode_den(int m) { return m == M_DEN_E || m == M_DEN_D; }
constexpr bool mode_enc_layers(int m) { return m == M_ENC || m == M_DEN_E; }   // two norms per block, no cross-attention

template <int PREC, int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_vae_rows(VaeRowsArgs a) {
    constexpr bool ENC = MODE == M_ENC;
    constexpr bool DEN = mode_den(MODE);
    constexpr bool EMB = ENC || DEN;               // stage 0 embeds 333 input features
    constexpr bool ENCL = mode_enc_layers(MODE);
    const int S = MODE == M_DEN_E ? a.S : (ENC ? kFrames + 2 : kFrames);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* comb = smem;
    float* fst = reinterpret_cast<float*>(smem);  // [16][kFeatStride] staged feats (last stage)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, r = lane & 15;
    const int b = blockIdx.x / a.tiles, rt = blockIdx.x - b * a.tiles;
    const int frame = rt * 16 + r;
    const bool rvalid = frame < S;
    const size_t row = (size_t)b * S + (rvalid ? frame : 0);
    const size_t nrows = (size_t)a.B * S;
    // this wave's slice of the stage's weight stream, pulled through a register ring (see amuse_dev.hpp: left
    // to itself hipcc serialises load -> wait -> mfma, one L2 round trip per KiB).  The ring runs up to kVR units
    // past the slice (into the neighbouring slice; the buffer is padded at its end) - those units are never used.
    const uint4* w = a.wstream + ((size_t)a.stage_base[a.stage] + (size_t)wave * a.stage_units[a.stage]) * 64 + lane;
    WRing<kVR> rg;
    ring_fill(rg, w);
    constexpr int U_O = gemm_units(PREC, kTiles, 2), U_F = gemm_units(PREC, kTiles, kTiles);
    constexpr int U_S = gemm_units(PREC, kTiles, 4);
    constexpr int U_P = gemm_units(PREC, 2, kTiles);   // M_DEN_D: the cross-attention's q of one head
    constexpr int P_O = 0, P_CQ = U_O % kVR, P_CO = (P_CQ + U_P) % kVR;
    constexpr int P_F1 = MODE == M_DEN_D ? (P_CO + U_O) % kVR : U_O % kVR, P_F2 = (P_F1 + U_F) % kVR, P_S = (P_F2 + U_F) % kVR;
    constexpr int P_Q0 = P_S, P_Q1 = (P_S + U_S) % kVR;   // in_proj / final phase without / with a skip linear before
    bool skipped = false;
    constexpr bool FAST = is_op16(PREC);
    f32x4 x[kTiles];

    constexpr int kEmbK = 22;  // 333 input features padded to 22 k-tiles (zero weights / zero operands beyond 333)
    constexpr int P_E = EMB ? gemm_units(PREC, 2, kEmbK) % kVR : 0;  // in_proj phase of stage 0
    const int npre = ENC ? 2 : (MODE == M_DEN_E ? a.npre : 0);   // rows in front of the 300 frames
    if (a.stage == 0) {
        if constexpr (EMB) {
            // xseq = cat(global_motion_token, skel_embedding(features)) + query_pos_encoder.pe[:302]  (vae.py:171-188)
            // Denoiser: cat(emb_latent, pose_embd(sample)) + query_pos (denoiser.py:178-181) / pose_embd(sample) + query_pos (:192-193)
            const int fi = frame - npre;
            const bool fvalid = rvalid && fi >= 0;
            const float* src = a.enc_feats + ((size_t)b * kFrames + (fvalid ? fi : 0)) * kFeats;
            f32x4 xin[kEmbK];
#pragma unroll
            for (int t = 0; t < kEmbK; ++t)
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int c = 16 * t + 4 * g + m;
                    xin[t][m] = (fvalid && c < kFeats) ? src[c] : 0.f;
                }
            f32x4 acc[2] = {splat4(0.f), splat4(0.f)};  // this wave's output tiles 2 wave, 2 wave + 1
            gemm_ring<PREC, 2, kEmbK, false, kVR, 0>(acc, xin, rg);
            f32x4 part[kTiles];
#pragma unroll
            for (int t = 0; t < kTiles; ++t) part[t] = (t >> 1) == wave ? acc[t & 1] : splat4(0.f);
            // all-gather of the four waves' tile pairs (+ skel_embedding.bias)
            combine_rs<false, FAST>(part, x, false, a.emb_bias, nullptr, nullptr, comb, wave, lane);
            if constexpr (DEN) {   // the condition tokens arrive with their positions added (k_time_tokens / k_cond_tokens)
                const float* pt = frame == 0 ? a.pre_tok_t + (size_t)b * a.pre_tok_t_stride
                                             : a.pre_tok_c + ((size_t)b * (npre - 1) + (frame - 1)) * kD;
#pragma unroll
                for (int t = 0; t < kTiles; ++t) {
                    const int c = 16 * t + 4 * g;
                    x[t] = fvalid ? x[t] + ld4(a.pe + (size_t)frame * kD + c) : (rvalid ? ld4(pt + c) : splat4(0.f));
                }
            } else {
#pragma unroll
            for (int t = 0; t < kTiles; ++t) {
                const int c = 16 * t + 4 * g;
                const f32x4 e = fvalid ? x[t] : ld4(a.tok + (frame & 1) * kD + c);
                x[t] = rvalid ? e + ld4(a.pe + (size_t)frame * kD + c) : splat4(0.f);
            }
            }
        } else {  // queries = zeros + query_pos_decoder.pe[:300]  (vae.py:220,258)
#pragma unroll
            for (int t = 0; t < kTiles; ++t)
                x[t] = rvalid ? ld4(a.pe + (size_t)frame * kD + 16 * t + 4 * g) : splat4(0.f);
        }
    } else {
        const int blk = a.stage - 1;
        const float* pv = a.pvec + blk * (MODE == M_DEN_D ? PVX_BLOCK : PV_BLOCK);
        f32x4 o[2];
#pragma unroll
        for (int t = 0; t < kTiles; ++t) x[t] = rvalid ? ld4(a.x + row * kD + 16 * t + 4 * g) : splat4(0.f);
#pragma unroll
        for (int td = 0; td < 2; ++td) {
            if constexpr (is_op16(PREC)) {
                // bf16 / fp16 modes: q, k, v and the attention output travel between the kernels in the operand format - the values
                // the MFMAs consume anyway (rounded at the same point as before, so results are unchanged), half the bytes
                const uint2 u = rvalid ? *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(a.attn_o) +
                                                                         row * kD + 32 * wave + 16 * td + 4 * g)
                                       : uint2{0u, 0u};
                o[td] = x16x4_to_f32<PREC>(u);
            } else {
                o[td] = rvalid ? ld4(a.attn_o + row * kD + 32 * wave + 16 * td + 4 * g) : splat4(0.f);
            }
        }
        f32x4 part[kTiles];
        // self-attention out_proj (split-K over heads) + residual + norm1
#pragma unroll
        for (int t = 0; t < kTiles; ++t) part[t] = splat4(0.f);
        gemm_ring<PREC, kTiles, 2, false, kVR, P_O>(part, o, rg);
        combine_rs<true, FAST>(part, x, true, pv + PV_OUT_B, pv + PV_LN1_W, pv + PV_LN1_B, comb, wave, lane);
        if constexpr (MODE == M_DEN_D) {
            // cross-attention of the tile's rows onto the clip's 2..4 memory tokens (cross_attention.py:337-343): q of head
            // `wave` (two output tiles, full K), the scores on the VALU (a lane holds 8 of its row's 32 head features), out_proj
            // split-K over the heads, residual + norm2 - the scheme of k_sampler_dec.hip, K / V from the same hoisted tables
            const int nmem = 1 + a.mem.ncond;
            f32x4 mk[4][2], mv[4][2];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j < nmem) {
                    const float* base = j == 0 ? a.mem.tkv + (size_t)b * a.mem.tkv_clip_stride + (size_t)blk * 2 * kD
                                               : a.mem.ckv + (((size_t)b * a.mem.ncond + (j - 1)) * kLayers + blk) * 2 * kD;
#pragma unroll
                    for (int oo = 0; oo < 2; ++oo) {
                        mk[j][oo] = ld4(base + 32 * wave + 16 * oo + 4 * g);
                        mv[j][oo] = ld4(base + kD + 32 * wave + 16 * oo + 4 * g);
                    }
                } else {
                    mk[j][0] = mk[j][1] = mv[j][0] = mv[j][1] = splat4(0.f);
                }
            }
            f32x4 q2[2] = {splat4(0.f), splat4(0.f)};
            gemm_ring<PREC, 2, kTiles, false, kVR, P_CQ>(q2, x, rg);
#pragma unroll
            for (int oo = 0; oo < 2; ++oo) q2[oo] = (q2[oo] + ld4(pv + PVX_CQ_B + 16 * (2 * wave + oo) + 4 * g)) * 0.17677669529663687f;
            float sc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float d = 0.f;
#pragma unroll
                for (int oo = 0; oo < 2; ++oo)
#pragma unroll
                    for (int m = 0; m < 4; ++m) d = fmaf(q2[oo][m], mk[j][oo][m], d);
                sc[j] = allreduce_g_sum(d);
            }
            float mx = sc[0];
#pragma unroll
            for (int j = 1; j < 4; ++j) mx = j < nmem ? fmaxf(mx, sc[j]) : mx;
            float pj[4], psum = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                pj[j] = j < nmem ? (FAST ? __builtin_amdgcn_exp2f(1.44269504088896340736f * (sc[j] - mx)) : expf(sc[j] - mx)) : 0.f;
                psum += pj[j];
            }
            f32x4 a2[2] = {splat4(0.f), splat4(0.f)};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float w = pj[j] / psum;
#pragma unroll
                for (int oo = 0; oo < 2; ++oo) a2[oo] += mv[j][oo] * w;
            }
#pragma unroll
            for (int t = 0; t < kTiles; ++t) part[t] = splat4(0.f);
            gemm_ring<PREC, kTiles, 2, false, kVR, P_CO>(part, a2, rg);
            combine_rs<true, FAST>(part, x, true, pv + PVX_CO_B, pv + PV_LN2_W, pv + PV_LN2_B, comb, wave, lane);
        } else if constexpr (MODE == M_DEC) {
            // cross-attention onto the single latent token == per-clip constant; residual + norm2
            const float* ca = a.ca + ((size_t)b * kLayers + blk) * kD;
#pragma unroll
            for (int t = 0; t < kTiles; ++t) x[t] = x[t] + ld4(ca + 16 * t + 4 * g);
            layer_norm_rows<is_op16(PREC)>(x, pv + PV_LN2_W, pv + PV_LN2_B, g);
        }
        // FFN + residual + norm3 (decoder layer) / norm2 (encoder layer, cross_attention.py:259-272)
        f32x4 hid[kTiles];
#pragma unroll
        for (int t = 0; t < kTiles; ++t) hid[t] = ld4(pv + PV_L1_B + 16 * (kTiles * wave + t) + 4 * g);
        gemm_ring<PREC, kTiles, kTiles, false, kVR, P_F1>(hid, x, rg);
        // bf16 mode: the activations are MFMA operands (rounded to bf16 next), so the polynomial GELU of the sampling
        // kernel serves (amuse_dev.hpp gelu_poly4: |error| <= 1.9e-4, a third of gelu_erf_fast's issue slots)
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
        gemm_ring<PREC, kTiles, kTiles, false, kVR, P_F2>(part, hid, rg);
        combine_rs<true, FAST>(part, x, true, pv + PV_L2_B, pv + (ENCL ? PV_LN2_W : PV_LN3_W), pv + (ENCL ? PV_LN2_B : PV_LN3_B),
                               comb, wave, lane);
        if (MODE != M_DEN_D && blk < 4 && wave == 0 && rvalid) {  // xs.append(x)
            float* sk = a.skip + ((size_t)blk * nrows + row) * kD;
#pragma unroll
            for (int t = 0; t < kTiles; ++t) st4(sk + 16 * t + 4 * g, x[t]);
        }
        if (MODE != M_DEN_D && blk >= 4 && blk <= 7) {  // x = linear_blocks[blk-4](cat(x, xs.pop())) ahead of output block blk+1
            const float* sk = a.skip + ((size_t)(7 - blk) * nrows + row) * kD;
            f32x4 src[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (wave < 2) src[i] = (wave == 1) ? x[4 + i] : x[i];
                else src[i] = rvalid ? ld4(sk + 16 * (4 * (wave - 2) + i) + 4 * g) : splat4(0.f);
            }
#pragma unroll
            for (int t = 0; t < kTiles; ++t) part[t] = splat4(0.f);
            gemm_ring<PREC, kTiles, 4, false, kVR, P_S>(part, src, rg);
            skipped = true;
            combine_rs<false, FAST>(part, x, false, a.pvec + PV_SKIP_B + (blk - 4) * kD, nullptr, nullptr, comb, wave, lane);
        }
    }

    if (a.stage < kLayers) {
        // residual stream for the next stage + in_proj of block `stage` (this wave's head)
        if (wave == 0 && rvalid) {
#pragma unroll
            for (int t = 0; t < kTiles; ++t) st4(a.x + row * kD + 16 * t + 4 * g, x[t]);
        }
        const float* pv = a.pvec + a.stage * (MODE == M_DEN_D ? PVX_BLOCK : PV_BLOCK);
        f32x4 qkv[6];
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            qkv[o] = ld4(pv + PV_IN_B + 16 * (2 * wave + o) + 4 * g);
            qkv[2 + o] = ld4(pv + PV_IN_B + kD + 16 * (2 * wave + o) + 4 * g);
            qkv[4 + o] = ld4(pv + PV_IN_B + 2 * kD + 16 * (2 * wave + o) + 4 * g);
        }
        if (a.stage == 0) gemm_ring<PREC, 6, kTiles, false, kVR, P_E>(qkv, x, rg);
        else if (skipped) gemm_ring<PREC, 6, kTiles, false, kVR, P_Q1>(qkv, x, rg);
        else gemm_ring<PREC, 6, kTiles, false, kVR, P_Q0>(qkv, x, rg);
        if (rvalid) {
            const size_t hrow = (((size_t)b * kHeads + wave) * S + frame) * 32;
            const float scaling = 0.17677669529663687f;
#pragma unroll
            for (int td = 0; td < 2; ++td) {
                if constexpr (is_op16(PREC)) {
                    const size_t off = hrow + 16 * td + 4 * g;
                    *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(a.q) + off) = f32_to_x16x4<PREC>(qkv[td] * scaling);
                    *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(a.k) + off) = f32_to_x16x4<PREC>(qkv[2 + td]);
                    *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(a.v) + off) = f32_to_x16x4<PREC>(qkv[4 + td]);
                } else {
                    st4(a.q + hrow + 16 * td + 4 * g, qkv[td] * scaling);
                    st4(a.k + hrow + 16 * td + 4 * g, qkv[2 + td]);
                    st4(a.v + hrow + 16 * td + 4 * g, qkv[4 + td]);
                }
            }
        }
    } else if constexpr (ENC) {
        // encoder.norm; only the two distribution rows leave the stack: mu = row 0, logvar = row 1 (vae.py:203-207)
        layer_norm_rows<is_op16(PREC)>(x, a.pvec + PV_FINAL_W, a.pvec + PV_FINAL_B, g);
        if (wave == 0 && rt == 0 && frame < 2) {
#pragma unroll
            for (int t = 0; t < kTiles; ++t) st4(a.stats_out + ((size_t)b * 2 + frame) * kD + 16 * t + 4 * g, x[t]);
        }
    } else {
        // decoder.norm -> final_layer (333 outputs padded to 24 tiles, 6 per wave) -> rotation epilogue
        // (Denoiser: encoder.norm / decoder.norm -> pose_proj -> mask -> eps_hat / scheduler update)
        layer_norm_rows<is_op16(PREC)>(x, a.pvec + (MODE == M_DEN_D ? PVX_FINAL_W : PV_FINAL_W), a.pvec + (MODE == M_DEN_D ? PVX_FINAL_B : PV_FINAL_B), g);
        f32x4 f[6];
#pragma unroll
        for (int o = 0; o < 6; ++o) f[o] = ld4(a.final_bias + 16 * (6 * wave + o) + 4 * g);
        gemm_ring<PREC, 6, kTiles, false, kVR, P_Q0>(f, x, rg);
        const int len = a.lengths ? a.lengths[b] + npre : S;
        const bool keep = rvalid && frame < len;  // output[~mask.T] = 0 (vae.py:274; denoiser.py:187,199)
#pragma unroll
        for (int o = 0; o < 6; ++o)
            st4(fst + r * kFeatStride + 16 * (6 * wave + o) + 4 * g, keep ? f[o] : splat4(0.f));
        __syncthreads();
        const int tid = threadIdx.x;
        const int rows_here = min(16, S - rt * 16);
        const size_t row0 = (size_t)b * S + rt * 16;
        if constexpr (DEN) {
            // eps_hat of the frame rows; then scheduler.step on x_t (amuse_hip.h amuse_schedule; k_sampler.hip's update), element by
            // element, each read and written by the same thread (x_out may alias x_t)
            const float* cf = a.coef;
            const float sb = cf ? cf[0] : 0.f, sa = cf ? cf[1] : 1.f, c0 = cf ? cf[2] : 0.f, cx = cf ? cf[3] : 0.f, ce = cf ? cf[4] : 0.f,
                        sg = cf ? cf[5] : 0.f, clipv = cf ? cf[6] : 0.f;
            constexpr bool FASTU = is_op16(PREC);
            const float inv_sa = 1.0f / sa;
            for (int i = tid; i < rows_here * kFeats; i += 256) {
                const int rr = i / kFeats, c = i - rr * kFeats;
                const int fr = rt * 16 + rr - npre;
                if (fr < 0) continue;
                const float e = fst[rr * kFeatStride + c];
                const size_t idx = ((size_t)b * kFrames + fr) * kFeats + c;
                if (a.feats_out) a.feats_out[idx] = e;
                if (a.x_out) {
#pragma clang fp contract(off)
                    const float xl = a.enc_feats[idx];
                    const float num = __fsub_rn(xl, __fmul_rn(sb, e));
                    float x0 = FASTU ? num * inv_sa : __fdiv_rn(num, sa);
                    if (clipv > 0.f) x0 = fminf(fmaxf(x0, -clipv), clipv);
                    float nx = __fmul_rn(c0, x0);
                    if (cx != 0.f) nx = __fadd_rn(nx, __fmul_rn(cx, xl));
                    if (ce != 0.f) nx = __fadd_rn(nx, __fmul_rn(ce, e));
                    if (sg != 0.f) {
                        const size_t el = (size_t)fr * kFeats + c;   // element of the clip's state: draw el % 4 of counter el / 4
                        const float z = a.step_noise ? a.step_noise[idx]
                                                     : counter_normal4(a.seed, a.clip0 + (uint64_t)b, (uint32_t)a.step, (uint32_t)(el >> 2), 1u)[el & 3];
                        nx = __fadd_rn(nx, __fmul_rn(sg, z));
                    }
                    a.x_out[idx] = nx;
                }
            }
            return;
        }
        if (a.feats_out) {
            for (int i = tid; i < rows_here * kFeats; i += 256) {
                const int rr = i / kFeats, c = i - rr * kFeats;
                a.feats_out[(row0 + rr) * kFeats + c] = fst[rr * kFeatStride + c];
            }
        }
        if (a.poses_out) {
            for (int i = tid; i < rows_here * kJoints; i += 256) {
                const int rr = i / kJoints, jn = i - rr * kJoints;
                float aa[3];
                rot6d_to_axis_angle(fst + rr * kFeatStride + 6 * jn, a.quat_mode, aa);
                float* dst = a.poses_out + ((row0 + rr) * kJoints + jn) * 3;
                dst[0] = aa[0]; dst[1] = aa[1]; dst[2] = aa[2];
            }
        }
        if (a.trans_out) {
            for (int i = tid; i < rows_here * 3; i += 256) {
                const int rr = i / 3, c = i - rr * 3;
                a.trans_out[(row0 + rr) * 3 + c] = fst[rr * kFeatStride + 330 + c];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
constexpr int kKS = 36;              // padded LDS row stride (floats) of the K_h / V_h images
constexpr int kKeyRows = 320;        // 300 keys padded to 20 tiles (zero rows, masked)
constexpr int kAttnLdsBytes = 2 * kKeyRows * kKS * 4;

template <int PREC, int MODE>
__global__ __launch_bounds__(256) void k_vae_attn(VaeAttnArgs a) {
    constexpr bool ENC = MODE == M_ENC;
    const int S = MODE == M_DEN_E ? a.S : (ENC ? kFrames + 2 : kFrames);  // encode: keys 0,1 = distribution tokens, always valid (vae.py:176-181)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Ks = reinterpret_cast<float*>(smem);
    float* Vs = Ks + kKeyRows * kKS;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, r = lane & 15;
    const int bh = blockIdx.x, b = bh / kHeads, h = bh - b * kHeads;
    const int len = (a.lengths && !mode_den(MODE)) ? a.lengths[b] + (ENC ? 2 : 0) : S;   // (the Denoiser variants pass no key mask)
    const float* qg = a.q + (size_t)bh * S * 32;
    const float* kg = a.k + (size_t)bh * S * 32;
    const float* vg = a.v + (size_t)bh * S * 32;
    for (int i = threadIdx.x; i < kKeyRows * 8; i += 256) {
        const int rowi = i >> 3, c4 = (i & 7) * 4;
        const bool ok = rowi < S;
        st4(Ks + rowi * kKS + c4, ok ? ld4(kg + rowi * 32 + c4) : splat4(0.f));
        st4(Vs + rowi * kKS + c4, ok ? ld4(vg + rowi * 32 + c4) : splat4(0.f));
    }
    __syncthreads();
    for (int qt = wave; qt < a.q_tiles; qt += 4) {
        const int fq = qt * 16 + r;
        const bool qvalid = fq < S;
        f32x4 q[2];
#pragma unroll
        for (int td = 0; td < 2; ++td) q[td] = qvalid ? ld4(qg + fq * 32 + 16 * td + 4 * g) : splat4(0.f);
        float m_run = -INFINITY, l_run = 0.f;
        f32x4 o[2] = {splat4(0.f), splat4(0.f)};
        [[maybe_unused]] const F16Pair qs = split_f16(q[0], q[1]);
#pragma unroll 1
        for (int jp = 0; jp < kKeyRows / 32; ++jp) {
            f32x4 st[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const float* kr = Ks + (32 * jp + 16 * u + r) * kKS + 4 * g;
                const f32x4 k0 = ld4(kr), k1 = ld4(kr + 16);
                st[u] = splat4(0.f);
                if constexpr (PREC == PREC_F32) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) st[u] = mfma_f32(k0[m], q[0][m], st[u]);
#pragma unroll
                    for (int m = 0; m < 4; ++m) st[u] = mfma_f32(k1[m], q[1][m], st[u]);
                } else if constexpr (PREC == PREC_F16X2) {   // fp32x: split operands, three MFMAs (amuse_dev.hpp)
                    const F16Pair ks = split_f16(k0, k1);
                    st[u] = mfma_f16(ks.lo, qs.hi, st[u]);
                    st[u] = mfma_f16(ks.hi, qs.lo, st[u]);
                    st[u] = mfma_f16(ks.hi, qs.hi, st[u]);
                } else {
                    st[u] = mfma_bf16(pack_bf16(k0, k1), pack_bf16(q[0], q[1]), st[u]);
                }
            }
            // lane (g, i) holds S[i][key = 32 jp + 16 u + 4 g + m]
            bool ok[2][4];
            float mx = -INFINITY;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    ok[u][m] = (32 * jp + 16 * u + 4 * g + m) < len;
                    mx = ok[u][m] ? fmaxf(mx, st[u][m]) : mx;
                }
            mx = allreduce_g_max(mx);
            const float m_new = fmaxf(m_run, mx);
            const float alpha = (m_new == -INFINITY) ? 1.0f : expf(m_run - m_new);
            f32x4 p[2];
            float ps = 0.f;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    p[u][m] = ok[u][m] ? expf(st[u][m] - m_new) : 0.f;
                    ps += p[u][m];
                }
            ps = allreduce_g_sum(ps);
            l_run = l_run * alpha + ps;
            o[0] *= alpha;
            o[1] *= alpha;
            m_run = m_new;
            // O^T[d][i] += sum_key V[key][d] P[i][key]; A operand lane (g, d): V[32 jp + 16 u + 4 g + m][16 td + d]
            const float* vr = Vs + (32 * jp + 4 * g) * kKS + r;
            if constexpr (PREC == PREC_F32) {
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        const float* vv = vr + (16 * u + m) * kKS;
                        o[0] = mfma_f32(vv[0], p[u][m], o[0]);
                        o[1] = mfma_f32(vv[16], p[u][m], o[1]);
                    }
            } else if constexpr (PREC == PREC_F16X2) {
                const F16Pair ps = split_f16(p[0], p[1]);
#pragma unroll
                for (int td = 0; td < 2; ++td) {
                    f32x4 lo, hi;
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        lo[m] = vr[m * kKS + 16 * td];
                        hi[m] = vr[(16 + m) * kKS + 16 * td];
                    }
                    const F16Pair vs = split_f16(lo, hi);
                    o[td] = mfma_f16(vs.lo, ps.hi, o[td]);
                    o[td] = mfma_f16(vs.hi, ps.lo, o[td]);
                    o[td] = mfma_f16(vs.hi, ps.hi, o[td]);
                }
            } else {
                const bf16x8 pb = pack_bf16(p[0], p[1]);
#pragma unroll
                for (int td = 0; td < 2; ++td) {
                    f32x4 lo, hi;
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        lo[m] = vr[m * kKS + 16 * td];
                        hi[m] = vr[(16 + m) * kKS + 16 * td];
                    }
                    o[td] = mfma_bf16(pack_bf16(lo, hi), pb, o[td]);
                }
            }
        }
        if (qvalid) {
            float* dst = a.o + ((size_t)b * S + fq) * kD + 32 * h + 4 * g;
            st4(dst, o[0] / l_run);
            st4(dst + 16, o[1] / l_run);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// bf16 attention: K_h and V_h^T are converted ONCE per workgroup into LDS images that are already MFMA
// fragments - Kb[key][g] = the 8 bf16 of k-slot group g (d = 4g+e | 16+4g+e-4) of that key row, Vt[pair][td][d][g]
// = the 8 bf16 of V[key(g,e)][16 td + d] over the 32 keys of a key-tile pair - so the inner loop is
// one ds_read_b128 per MFMA operand, no packing.  Each wave walks TWO 16-query tiles at a time (shared K/V
// fragments, two independent MFMA/softmax chains).  Per (q-tile, 32 keys): 2 score MFMAs (K.Q^T, K = d = 32),
// online softmax on 8 register values + two permlane butterflies, 2 PV MFMAs (V^T.P^T, K = 32 keys).
constexpr int kPairs = kKeyRows / 32;                        // 10
constexpr int kAttnBf16LdsBytes = kKeyRows * 64 + kPairs * 2 * 16 * 64;  // 20 KiB + 20 KiB
constexpr int kAttnSplitMaxClips = 48;   // up to this many clips per launch a (clip, head) pair is five workgroups (below); measured (profiles/r03_attn_split_threshold.txt): pays up to ~48 clips in bf16 and fp32x, not at 63

template <int P16, int NQ>   // P16 = PREC_BF16 / PREC_F16: the operand format of q, k, v, p and the output
__device__ __forceinline__ void attn_qtiles_bf16(const uint4* Kb, const uint4* Vt, const unsigned short* qg,
                                                 unsigned short* og, int qt0, int len, int g, int r, const int S) {
    constexpr float kLog2e = 1.44269504088896340736f;
    typedef Op16<P16> Op;
    typedef typename Op::vec OPV;
    OPV qb[NQ];
    float m_run[NQ], l_run[NQ];
    f32x4 o[NQ][2];
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
        const int fq = (qt0 + 4 * n) * 16 + r;
        const bool qv = fq < S;
        const uint2 q0 = qv ? *reinterpret_cast<const uint2*>(qg + fq * 32 + 4 * g) : uint2{0u, 0u};
        const uint2 q1 = qv ? *reinterpret_cast<const uint2*>(qg + fq * 32 + 16 + 4 * g) : uint2{0u, 0u};
        qb[n] = __builtin_bit_cast(OPV, uint4{q0.x, q0.y, q1.x, q1.y});
        m_run[n] = -INFINITY;
        l_run[n] = 0.f;
        o[n][0] = o[n][1] = splat4(0.f);
    }
    // one pair of key tiles.  MASKED (wave-uniform) = the pair reaches past `len`: full pairs run without the compares and selects; a pair in
    // which no row's running maximum moves keeps alpha = 1 and skips the rescaling (x * 1.0f is x): neither shortcut changes a bit
    auto pair = [&](int jp, auto masked) {
        constexpr bool MASKED = decltype(masked)::value;
        const OPV k0 = __builtin_bit_cast(OPV, Kb[(32 * jp + r) * 4 + g]);
        const OPV k1 = __builtin_bit_cast(OPV, Kb[(32 * jp + 16 + r) * 4 + g]);
        const OPV v0 = __builtin_bit_cast(OPV, Vt[((jp * 2 + 0) * 16 + r) * 4 + g]);
        const OPV v1 = __builtin_bit_cast(OPV, Vt[((jp * 2 + 1) * 16 + r) * 4 + g]);
        bool ok[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int m = 0; m < 4; ++m) ok[u][m] = !MASKED || (32 * jp + 16 * u + 4 * g + m) < len;
#pragma unroll
        for (int n = 0; n < NQ; ++n) {
            f32x4 st[2];
            st[0] = Op::mfma(k0, qb[n], splat4(0.f));  // lane (g, i): S[i][32 jp + 4 g + m]
            st[1] = Op::mfma(k1, qb[n], splat4(0.f));  //              S[i][32 jp + 16 + 4 g + m]
            float mx = -INFINITY;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int m = 0; m < 4; ++m) mx = ok[u][m] ? fmaxf(mx, st[u][m] * kLog2e) : mx;
            mx = allreduce_g_max(mx);
            float m_new = m_run[n], alpha = 1.0f;
            if (__builtin_amdgcn_ballot_w64(mx > m_run[n]) != 0) {   // (uniform)
                m_new = fmaxf(m_run[n], mx);
                alpha = (m_new == -INFINITY) ? 1.0f : __builtin_amdgcn_exp2f(m_run[n] - m_new);
                o[n][0] = o[n][0] * alpha;
                o[n][1] = o[n][1] * alpha;
                m_run[n] = m_new;
            }
            f32x4 p[2];
            float ps = 0.f;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    // (the maximum is taken over the ROUNDED products, the exponent is one fused multiply-add: the contraction hipcc chose
                    // for the round-1 kernel, spelled out so that restructuring the loop cannot change the bits)
                    p[u][m] = ok[u][m] ? __builtin_amdgcn_exp2f(__builtin_fmaf(st[u][m], kLog2e, -m_new)) : 0.f;
                    ps += p[u][m];
                }
            ps = allreduce_g_sum(ps);
            l_run[n] = l_run[n] * alpha + ps;
            const OPV pb = Op::pack(p[0], p[1]);
            o[n][0] = Op::mfma(v0, pb, o[n][0]);  // O^T[d][i] += sum_key V[key][d] P[i][key]
            o[n][1] = Op::mfma(v1, pb, o[n][1]);
        }
    };
    const int full = min(len / 32, kPairs);   // pairs entirely below len
#pragma unroll 1
    for (int jp = 0; jp < full; ++jp) pair(jp, std::false_type{});
#pragma unroll 1
    for (int jp = full; jp < kPairs; ++jp) pair(jp, std::true_type{});
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
        const int fq = (qt0 + 4 * n) * 16 + r;
        if (fq < S) {
            unsigned short* dst = og + (size_t)fq * kD + 4 * g;
            *reinterpret_cast<uint2*>(dst) = f32_to_x16x4<P16>(o[n][0] / l_run[n]);
            *reinterpret_cast<uint2*>(dst + 16) = f32_to_x16x4<P16>(o[n][1] / l_run[n]);
        }
    }
}

template <int P16, int MODE>
__global__ __launch_bounds__(256) void k_vae_attn_bf16(VaeAttnArgs a) {
    constexpr bool ENC = MODE == M_ENC;
    const int S = MODE == M_DEN_E ? a.S : (ENC ? kFrames + 2 : kFrames);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint4* Kb = reinterpret_cast<uint4*>(smem);
    uint4* Vt = Kb + kKeyRows * 4;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, r = lane & 15;
    const int bh = blockIdx.x, b = bh / kHeads, h = bh - b * kHeads;
    const int len = (a.lengths && !mode_den(MODE)) ? a.lengths[b] + (ENC ? 2 : 0) : S;
    // q, k, v are 16-bit here ([B * heads][S][32], written by k_vae_rows<P16> in its operand format), as is the output
    const unsigned short* qg = reinterpret_cast<const unsigned short*>(a.q) + (size_t)bh * S * 32;
    const unsigned short* kg = reinterpret_cast<const unsigned short*>(a.k) + (size_t)bh * S * 32;
    const unsigned short* vg = reinterpret_cast<const unsigned short*>(a.v) + (size_t)bh * S * 32;
    for (int i = threadIdx.x; i < kKeyRows * 4; i += 256) {  // K fragments: item = (key row, slot group)
        const int row = i >> 2, gg = i & 3;
        const bool ok = row < S;
        const uint2 lo = ok ? *reinterpret_cast<const uint2*>(kg + row * 32 + 4 * gg) : uint2{0u, 0u};
        const uint2 hi = ok ? *reinterpret_cast<const uint2*>(kg + row * 32 + 16 + 4 * gg) : uint2{0u, 0u};
        Kb[i] = uint4{lo.x, lo.y, hi.x, hi.y};
    }
    for (int i = threadIdx.x; i < kPairs * 2 * 16 * 4; i += 256) {  // V^T fragments: item = ((pair, td), g, d)
        const int d = i & 15, gg = (i >> 4) & 3, pt = i >> 6;  // pt = jp * 2 + td
        const int jp = pt >> 1, td = pt & 1;
        unsigned lo[4], hi[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k0 = 32 * jp + 4 * gg + e, k1 = k0 + 16;
            lo[e] = k0 < S ? vg[k0 * 32 + 16 * td + d] : 0u;
            hi[e] = k1 < S ? vg[k1 * 32 + 16 * td + d] : 0u;
        }
        Vt[(pt * 16 + d) * 4 + gg] = uint4{lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16), hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16)};
    }
    __syncthreads();
    unsigned short* og = reinterpret_cast<unsigned short*>(a.o) + (size_t)b * S * kD + 32 * h;
    // 19 query tiles: wave w owns tiles w, w+4, w+8, w+12 (two pairs) and w+16 (waves 0..2)
    if (a.q_tiles == 1) {  // last encoder block: only the distribution rows (tile 0) are consumed downstream
        if (wave == 0 && blockIdx.y == 0) attn_qtiles_bf16<P16, 1>(Kb, Vt, qg, og, 0, len, g, r, S);
        return;
    }
    // Few clips (launch_vae_attn: gridDim.y = 5): a (clip, head) pair is FIVE workgroups, each with the whole K / V image and four
    // of the query tiles - one per wave - instead of one workgroup walking all nineteen: 24 -> 11 us per launch for a single clip,
    // where the nine attention launches are 60 % of the decode.  Which workgroup computes a query tile does not change its bits.
    if (gridDim.y > 1) {
        const int qt = 4 * blockIdx.y + wave;
        if (qt < kRowTiles) attn_qtiles_bf16<P16, 1>(Kb, Vt, qg, og, qt, len, g, r, S);
        return;
    }
    attn_qtiles_bf16<P16, 2>(Kb, Vt, qg, og, wave, len, g, r, S);
    attn_qtiles_bf16<P16, 2>(Kb, Vt, qg, og, wave + 8, len, g, r, S);
    if (wave + 16 < kRowTiles) attn_qtiles_bf16<P16, 1>(Kb, Vt, qg, og, wave + 16, len, g, r, S);
}

// ------------------------------------------------------------------------------------------------
// fp32x attention (AMUSE_PREC_F32X): the layout of the bf16 kernel above with split-fp16 operands.  K_h and V_h^T are
// split ONCE per workgroup into hi / lo fp16 fragment images in LDS (x = hi + lo, amuse_dev.hpp split_f16), so the inner
// loop reads ready MFMA operands - one ds_read_b128 each - and only q (once per tile) and p (once per 32 keys) are split in
// registers; scores and PV are three v_mfma_f32_16x16x32_f16 each (hi.hi + hi.lo + lo.hi, fp32 accumulate), the softmax
// runs in fp32 in exp2 units (q arrives scaled by 1 / sqrt(32) from k_vae_rows; log2 e is folded in before the split).
// q, k, v and the output are fp32 in HBM, as in the fp32 mode.
constexpr int kAttnXLdsBytes = 2 * (kKeyRows * 64 + kPairs * 2 * 16 * 64);   // hi + lo images of K and V^T: 80 KiB

template <int NQ>
__device__ __forceinline__ void attn_qtiles_x(const uint4* Kh, const uint4* Kl, const uint4* Vh, const uint4* Vl, const float* qg,
                                              float* og, int qt0, int len, int g, int r, const int S) {
    constexpr float kLog2e = 1.44269504088896340736f;
    F16Pair qs[NQ];
    float m_run[NQ], l_run[NQ];
    f32x4 o[NQ][2];
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
        const int fq = (qt0 + 4 * n) * 16 + r;
        const bool qv = fq < S;
        const f32x4 q0 = qv ? ld4(qg + fq * 32 + 4 * g) * kLog2e : splat4(0.f);
        const f32x4 q1 = qv ? ld4(qg + fq * 32 + 16 + 4 * g) * kLog2e : splat4(0.f);
        qs[n] = split_f16(q0, q1);
        m_run[n] = -INFINITY;
        l_run[n] = 0.f;
        o[n][0] = o[n][1] = splat4(0.f);
    }
    // one pair of key tiles (32 keys).  MASKED = the pair reaches past `len` (at most the last pair of a full-length clip; wave-uniform): the
    // full pairs run without the 8 compares and 16 selects.  A pair in which no row's running maximum moves (wave-uniform ballot) keeps
    // alpha = exp2(0) = 1 and skips the rescaling of o and l - x * 1.0f is x, so both shortcuts leave every bit as it was.
    auto pair = [&](int jp, auto masked) {
        constexpr bool MASKED = decltype(masked)::value;
        f16x8 kh[2], kl[2], vh[2], vl[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            kh[u] = __builtin_bit_cast(f16x8, Kh[(32 * jp + 16 * u + r) * 4 + g]);
            kl[u] = __builtin_bit_cast(f16x8, Kl[(32 * jp + 16 * u + r) * 4 + g]);
            vh[u] = __builtin_bit_cast(f16x8, Vh[((jp * 2 + u) * 16 + r) * 4 + g]);
            vl[u] = __builtin_bit_cast(f16x8, Vl[((jp * 2 + u) * 16 + r) * 4 + g]);
        }
        bool ok[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int m = 0; m < 4; ++m) ok[u][m] = !MASKED || (32 * jp + 16 * u + 4 * g + m) < len;
#pragma unroll
        for (int n = 0; n < NQ; ++n) {
            f32x4 st[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {   // lane (g, i): S[i][32 jp + 16 u + 4 g + m], log2 units
                st[u] = mfma_f16(kl[u], qs[n].hi, splat4(0.f));
                st[u] = mfma_f16(kh[u], qs[n].lo, st[u]);
                st[u] = mfma_f16(kh[u], qs[n].hi, st[u]);
            }
            float mx = -INFINITY;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int m = 0; m < 4; ++m) mx = ok[u][m] ? fmaxf(mx, st[u][m]) : mx;
            mx = allreduce_g_max(mx);
            const bool moved = __builtin_amdgcn_ballot_w64(mx > m_run[n]) != 0;   // (uniform)
            float m_new = m_run[n], alpha = 1.0f;
            if (moved) {
                m_new = fmaxf(m_run[n], mx);
                alpha = (m_new == -INFINITY) ? 1.0f : __builtin_amdgcn_exp2f(m_run[n] - m_new);
                o[n][0] = o[n][0] * alpha;
                o[n][1] = o[n][1] * alpha;
                m_run[n] = m_new;
            }
            f32x4 p[2];
            float ps = 0.f;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    p[u][m] = ok[u][m] ? __builtin_amdgcn_exp2f(st[u][m] - m_new) : 0.f;
                    ps += p[u][m];
                }
            ps = allreduce_g_sum(ps);
            l_run[n] = l_run[n] * alpha + ps;
            const F16Pair pp = split_f16(p[0], p[1]);
#pragma unroll
            for (int td = 0; td < 2; ++td) {   // O^T[d][i] += sum_key V[key][d] P[i][key]
                o[n][td] = mfma_f16(vl[td], pp.hi, o[n][td]);
                o[n][td] = mfma_f16(vh[td], pp.lo, o[n][td]);
                o[n][td] = mfma_f16(vh[td], pp.hi, o[n][td]);
            }
        }
    };
    const int full = min(len / 32, kPairs);   // pairs entirely below len
#pragma unroll 1
    for (int jp = 0; jp < full; ++jp) pair(jp, std::false_type{});
#pragma unroll 1
    for (int jp = full; jp < kPairs; ++jp) pair(jp, std::true_type{});
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
        const int fq = (qt0 + 4 * n) * 16 + r;
        if (fq < S) {
            float* dst = og + (size_t)fq * kD + 4 * g;
            st4(dst, o[n][0] / l_run[n]);
            st4(dst + 16, o[n][1] / l_run[n]);
        }
    }
}

// NW = 4 or 8 waves per workgroup.  The 80 KiB of fragment images leave room for one workgroup (two at best) per CU, so with four waves the
// CU runs the attention on 4-8 waves; eight waves share one image and halve a workgroup's time (launch_vae_attn picks NW by the grid).
template <int MODE, int NW>
__global__ __launch_bounds__(64 * NW) void k_vae_attn_x(VaeAttnArgs a) {
    constexpr bool ENC = MODE == M_ENC;
    const int S = MODE == M_DEN_E ? a.S : (ENC ? kFrames + 2 : kFrames);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint4* Kh = reinterpret_cast<uint4*>(smem);
    uint4* Kl = Kh + kKeyRows * 4;
    uint4* Vh = Kl + kKeyRows * 4;
    uint4* Vl = Vh + kPairs * 2 * 16 * 4;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, r = lane & 15;
    const int bh = blockIdx.x, b = bh / kHeads, h = bh - b * kHeads;
    const int len = (a.lengths && !mode_den(MODE)) ? a.lengths[b] + (ENC ? 2 : 0) : S;
    const float* qg = a.q + (size_t)bh * S * 32;
    const float* kg = a.k + (size_t)bh * S * 32;
    const float* vg = a.v + (size_t)bh * S * 32;
    for (int i = threadIdx.x; i < kKeyRows * 4; i += 64 * NW) {  // K fragments: item = (key row, slot group)
        const int row = i >> 2, gg = i & 3;
        const bool ok = row < S;
        const F16Pair ks = split_f16(ok ? ld4(kg + row * 32 + 4 * gg) : splat4(0.f), ok ? ld4(kg + row * 32 + 16 + 4 * gg) : splat4(0.f));
        Kh[i] = __builtin_bit_cast(uint4, ks.hi);
        Kl[i] = __builtin_bit_cast(uint4, ks.lo);
    }
    for (int i = threadIdx.x; i < kPairs * 2 * 16 * 4; i += 64 * NW) {  // V^T fragments: item = ((pair, td), g, d)
        const int d = i & 15, gg = (i >> 4) & 3, pt = i >> 6;  // pt = jp * 2 + td
        const int jp = pt >> 1, td = pt & 1;
        f32x4 lo, hi;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k0 = 32 * jp + 4 * gg + e, k1 = k0 + 16;
            lo[e] = k0 < S ? vg[k0 * 32 + 16 * td + d] : 0.f;
            hi[e] = k1 < S ? vg[k1 * 32 + 16 * td + d] : 0.f;
        }
        const F16Pair vs = split_f16(lo, hi);
        Vh[(pt * 16 + d) * 4 + gg] = __builtin_bit_cast(uint4, vs.hi);
        Vl[(pt * 16 + d) * 4 + gg] = __builtin_bit_cast(uint4, vs.lo);
    }
    __syncthreads();
    float* og = a.o + (size_t)b * S * kD + 32 * h;
    if (a.q_tiles == 1) {  // last encoder block: only the distribution rows (tile 0) are consumed downstream
        if (wave == 0 && blockIdx.y == 0) attn_qtiles_x<1>(Kh, Kl, Vh, Vl, qg, og, 0, len, g, r, S);
        return;
    }
    if (gridDim.y > 1) {   // few clips: five workgroups per (clip, head), one query tile per wave (as in k_vae_attn_bf16)
        const int qt = 4 * blockIdx.y + wave;
        if (wave < 4 && qt < kRowTiles) attn_qtiles_x<1>(Kh, Kl, Vh, Vl, qg, og, qt, len, g, r, S);
        return;
    }
    if constexpr (NW == 8) {   // waves 0..3: tile pairs (w, w + 4); waves 4..7: (w + 4, w + 8); tiles 16..18: waves 0..2
        attn_qtiles_x<2>(Kh, Kl, Vh, Vl, qg, og, wave < 4 ? wave : wave + 4, len, g, r, S);
        if (wave + 16 < kRowTiles) attn_qtiles_x<1>(Kh, Kl, Vh, Vl, qg, og, wave + 16, len, g, r, S);
    } else {
        attn_qtiles_x<2>(Kh, Kl, Vh, Vl, qg, og, wave, len, g, r, S);
        attn_qtiles_x<2>(Kh, Kl, Vh, Vl, qg, og, wave + 8, len, g, r, S);
        if (wave + 16 < kRowTiles) attn_qtiles_x<1>(Kh, Kl, Vh, Vl, qg, og, wave + 16, len, g, r, S);
    }
}

template <typename K>
hipError_t set_lds(K kern, int bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

}  // namespace

namespace {
template <int PREC, int MODE>
hipError_t rows_one(const VaeRowsArgs& a, hipStream_t stream, bool setup) {
    if (setup) return set_lds(&k_vae_rows<PREC, MODE>, kRowsLdsBytes);
    hipLaunchKernelGGL((k_vae_rows<PREC, MODE>), dim3(a.B * a.tiles), dim3(256), kRowsLdsBytes, stream, a);
    return hipSuccess;
}
template <int MODE>
hipError_t rows_prec(const VaeRowsArgs& a, int precision, hipStream_t stream, bool setup) {
    if (setup) {
        hipError_t e = rows_one<PREC_F32, MODE>(a, stream, true);
        if (e == hipSuccess) e = rows_one<PREC_BF16, MODE>(a, stream, true);
        if (e == hipSuccess) e = rows_one<PREC_F16X2, MODE>(a, stream, true);
        if (e == hipSuccess) e = rows_one<PREC_F16, MODE>(a, stream, true);
        return e;
    }
    if (precision == PREC_F32) return rows_one<PREC_F32, MODE>(a, stream, false);
    if (precision == PREC_F16X2) return rows_one<PREC_F16X2, MODE>(a, stream, false);
    if (precision == PREC_F16) return rows_one<PREC_F16, MODE>(a, stream, false);
    return rows_one<PREC_BF16, MODE>(a, stream, false);
}
hipError_t rows_mode(const VaeRowsArgs& a, int precision, int mode, hipStream_t stream, bool setup) {
    if (mode == M_ENC) return rows_prec<M_ENC>(a, precision, stream, setup);
    if (mode == M_DEN_E) return rows_prec<M_DEN_E>(a, precision, stream, setup);
    if (mode == M_DEN_D) return rows_prec<M_DEN_D>(a, precision, stream, setup);
    return rows_prec<M_DEC>(a, precision, stream, setup);
}

// the three attention kernels (generic fp32 / f16x2, 16-bit fragment images, fp32x fragment images) of one MODE
template <int MODE>
hipError_t attn_mode(const VaeAttnArgs& a, int precision, hipStream_t stream, bool setup) {
    if (setup) {
        hipError_t e = set_lds(&k_vae_attn<PREC_F32, MODE>, kAttnLdsBytes);
        if (e == hipSuccess) e = set_lds(&k_vae_attn<PREC_F16X2, MODE>, kAttnLdsBytes);
        if (e == hipSuccess) e = set_lds(&k_vae_attn_x<MODE, 4>, kAttnXLdsBytes);
        if (e == hipSuccess) e = set_lds(&k_vae_attn_x<MODE, 8>, kAttnXLdsBytes);
        if (e == hipSuccess) e = set_lds(&k_vae_attn_bf16<PREC_BF16, MODE>, kAttnBf16LdsBytes);
        if (e == hipSuccess) e = set_lds(&k_vae_attn_bf16<PREC_F16, MODE>, kAttnBf16LdsBytes);
        return e;
    }
    const dim3 grid(a.B * kHeads), block(256);
    const dim3 grid16(a.B * kHeads, a.B <= kAttnSplitMaxClips ? 5 : 1);   // the fragment-image kernels (16-bit, fp32x): query tiles over 5 workgroups for small batches
    if (precision == PREC_F32) {
        hipLaunchKernelGGL((k_vae_attn<PREC_F32, MODE>), grid, block, kAttnLdsBytes, stream, a);
    } else if (precision == PREC_F16X2) {
        // the fragment-image kernel: eight waves per workgroup for unsplit launches (measured: decode 2.42 -> 2.30 ms at 256 clips, 0.93 -> 0.86 at 64); the split
        // launches of small batches keep four: one query tile per wave
        if (grid16.y == 1) hipLaunchKernelGGL((k_vae_attn_x<MODE, 8>), grid16, dim3(512), kAttnXLdsBytes, stream, a);
        else hipLaunchKernelGGL((k_vae_attn_x<MODE, 4>), grid16, block, kAttnXLdsBytes, stream, a);
    } else if (precision == PREC_F16) {
        hipLaunchKernelGGL((k_vae_attn_bf16<PREC_F16, MODE>), grid16, block, kAttnBf16LdsBytes, stream, a);
    } else {
        hipLaunchKernelGGL((k_vae_attn_bf16<PREC_BF16, MODE>), grid16, block, kAttnBf16LdsBytes, stream, a);
    }
    return hipSuccess;
}
}  // namespace

// mode: VAE_MODE_* (amuse_kernels.hpp) = M_DEC / M_ENC / M_DEN_E / M_DEN_D above
hipError_t launch_vae_rows(const VaeRowsArgs& a, int precision, int mode, hipStream_t stream) {
    static DeviceOnce once;
    int dev_;
    if (!once.done(&dev_)) {
        for (int m = 0; m < 4; ++m) {
            hipError_t e = rows_mode(a, precision, m, stream, true);
            if (e != hipSuccess) return e;
        }
        once.set(dev_);
    }
    hipError_t e = rows_mode(a, precision, mode, stream, false);
    return e != hipSuccess ? e : hipGetLastError();
}

hipError_t launch_vae_attn(const VaeAttnArgs& a, int precision, int mode, hipStream_t stream) {
    static DeviceOnce once;
    int dev_;
    if (!once.done(&dev_)) {
        hipError_t e = attn_mode<M_DEC>(a, precision, stream, true);
        if (e == hipSuccess) e = attn_mode<M_ENC>(a, precision, stream, true);
        if (e == hipSuccess) e = attn_mode<M_DEN_E>(a, precision, stream, true);
        if (e != hipSuccess) return e;
        once.set(dev_);
    }
    // (M_DEN_D's self-attention is M_DEC's kernel: S = 300, and the Denoiser passes lengths = NULL - no key mask)
    hipError_t e = mode == M_ENC ? attn_mode<M_ENC>(a, precision, stream, false)
                 : mode == M_DEN_E ? attn_mode<M_DEN_E>(a, precision, stream, false)
                                   : attn_mode<M_DEC>(a, precision, stream, false);
    return e != hipSuccess ? e : hipGetLastError();
}

}  // namespace amuse
