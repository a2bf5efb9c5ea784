// MotionPrior.decode as ONE persistent kernel per clip (bf16 throughput mode): reference models/latent_diffusion/vae.py:216-278
// = zeros(300,B,128) + learned PE -> SkipTransformerDecoder (cross_attention.py:66-125) of 9 TransformerDecoderLayer.forward_post
// blocks (cross_attention.py:297-345) -> final_layer Linear(128 -> 333) -> 6D -> matrix -> axis-angle (infer_ldm.py:168-173).
//
// Why one kernel: the staged path (k_vae.hip: k_vae_rows / k_vae_attn alternating, 19 launches) moves q, k, v, o and the
// residual stream through HBM between launches (79 MB per attention launch at 256 clips, more than the 32 MB of L2) and
// re-streams a block's 393 KB of weights once per 16-row tile.  Here a clip never leaves its CU:
//
//   * workgroup = one clip = 4 waves, ONE per SIMD (512 registers each); wave w owns the five 16-row tiles w, w+4, .., w+16
//     (19 real tiles; the 20th is padding that costs nothing - every wave does five) and keeps their fp32 residual stream in
//     registers for the whole network, in the row-lane layout of amuse_dev.hpp (accumulators of one GEMM are the operands
//     of the next).  Row work (projections, FFN, LayerNorm, skip linears, final layer) needs no other wave: no split-K, no
//     combine, no exchange.
//   * every weight unit (1 KiB MFMA A-fragment) a wave pulls from L2 feeds FIVE MFMAs (one per row tile, five independent
//     accumulators - the ILP that keeps the matrix pipe busy with one wave per SIMD); the stream is the same for the four
//     waves (L1 absorbs most of the re-reads) and runs through a 16-slot register ring, re-armed as it is consumed.
//   * the only cross-wave traffic is K_h and V_h^T of the current head: written to LDS as ready MFMA fragments (one
//     ds_write_b128 / two ds_write_b64 per row tile), double-buffered over heads - ONE barrier per head.  Scores
//     S^T = K.Q^T and O^T = V^T.P^T run on v_mfma_f32_16x16x32_bf16 with the softmax along registers; the attention output
//     of head h is, unchanged, the B operand of out_proj's k-slice h, accumulated straight into the residual registers.
//   * the U-Net skip stack (4 x 300 x 128) goes to global memory as packed bf16 MFMA operands - exactly the rounding the
//     skip linear applies anyway - written and read back by the same lanes (no synchronisation); 82 MB per 256 clips.
//   * the one-token cross-attention is the per-clip constant of k_vae_ca (k_misc.hip), as in the staged path.
//
// HBM traffic per clip: 320 KB skip write + 320 KB skip read + 201.6 KB poses/trans out + 4.6 KB constants in; weights
// (3.8 MB bf16 for the whole decoder) come out of L2.
#include "amuse_dev.hpp"
#include "amuse_kernels.hpp"

namespace amuse {
namespace {

constexpr int NT = 5;                     // row tiles per wave
constexpr int kFR = kVaeFusedRing;        // weight ring depth (units)
constexpr int kKeyRows = 320;             // 300 keys padded to 20 tiles
constexpr int kPairs = kKeyRows / 32;     // 10 key-tile pairs
constexpr int kKvBytes = kKeyRows * 64 + kPairs * 2 * 16 * 64;   // K fragments 20 KiB + V^T fragments 20 KiB
constexpr int kFeatStride = 388;
constexpr int kStageBytes = 16 * kFeatStride * 4;                // per-wave staging tile of the last stage
constexpr int kPvBytes = PV_BLOCK * 4;
// LDS: [2 x K/V image | 4 x per-wave staging (reuses the K/V images and beyond)] | 2 x block params
constexpr int kMainBytes = (2 * kKvBytes > 4 * kStageBytes) ? 2 * kKvBytes : 4 * kStageBytes;
static_assert(kMainBytes + 2 * kPvBytes == kVaeFusedLdsBytes, "LDS layout and amuse_kernels.hpp disagree");

typedef WRing<kFR> Ring;

// ablation switches for timing experiments (tools/build_variant.sh): 1 no weight re-arm loads, 2 no attention, 4 no FFN,
// 8 no softmax arithmetic (scores fed to PV as they are), 16 no cross-attention constant loads.  0 in the product.
#ifndef AMUSE_FABL
#define AMUSE_FABL 0
#endif

// consume the unit in ring slot `ph` (a compile-time constant after unrolling) and re-arm the slot kFR units ahead
__device__ __forceinline__ bf16x8 take(Ring& rg, int ph) {
    const uint4 u = rg.s[ph % kFR];
    if constexpr (!(AMUSE_FABL & 1)) {
        rg.s[ph % kFR] = ldw(rg.next);
        rg.next += 64;
    }
    return __builtin_bit_cast(bf16x8, u);
}

// acc[j][o] += W_o . x_j for the NT row tiles of this wave: NC k-tile pairs x NO output tiles, stream order k-pair outer.
// SWAP: activations are the A operand (result in feature-lane layout: lane (g, f) holds rows 4 g + m of feature f).
template <int NO, int NC, bool SWAP, int PH>
__device__ __forceinline__ void gemm5(f32x4 (&acc)[NT][NO], const bf16x8 (&xb)[NT][NC], Ring& rg) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
#pragma unroll
        for (int o = 0; o < NO; ++o) {
            const bf16x8 wf = take(rg, PH + c * NO + o);
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[j][o] = SWAP ? mfma_bf16(xb[j][c], wf, acc[j][o]) : mfma_bf16(wf, xb[j][c], acc[j][o]);
        }
    }
}

__device__ __forceinline__ void pack_rows(bf16x8 (&xb)[NT][4], const f32x4 (&x)[NT][kTiles]) {
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) xb[j][c] = pack_bf16(x[j][2 * c], x[j][2 * c + 1]);
}

// softmax(Q K^T) V for one 16-query tile against all keys of head h, K / V^T fragments in LDS.  Two key chunks of five
// tile pairs (160 keys) each, merged online: the scores of a chunk (40 registers) are complete before its exponentials.
__device__ __forceinline__ bf16x8 attend(const uint4* Kb, const uint4* Vt, bf16x8 qb, int len, int g, int r) {
    if constexpr ((AMUSE_FABL & 2) != 0) return qb;
    float m_run = -INFINITY, l_run = 0.f;
    f32x4 o[2] = {splat4(0.f), splat4(0.f)};
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
        f32x4 st[10];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int jp = 5 * ch + i;
            const bf16x8 k0 = __builtin_bit_cast(bf16x8, Kb[(32 * jp + r) * 4 + g]);
            const bf16x8 k1 = __builtin_bit_cast(bf16x8, Kb[(32 * jp + 16 + r) * 4 + g]);
            st[2 * i] = mfma_bf16(k0, qb, splat4(0.f));      // lane (g, i): S[query i][key 32 jp + 4 g + m] (log2 units)
            st[2 * i + 1] = mfma_bf16(k1, qb, splat4(0.f));  //              S[query i][key 32 jp + 16 + 4 g + m]
        }
        if (160 * (ch + 1) > len) {  // wave-uniform: only a chunk that holds the sequence end is masked
#pragma unroll
            for (int i = 0; i < 10; ++i)
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    st[i][m] = (160 * ch + 16 * i + 4 * g + m) < len ? st[i][m] : -INFINITY;
        }
        float mx = st[0][0];
#pragma unroll
        for (int i = 0; i < 10; ++i)
#pragma unroll
            for (int m = 0; m < 4; ++m) mx = fmaxf(mx, st[i][m]);
        mx = allreduce_g_max(mx);
        const float m_new = fmaxf(m_run, mx);
        const float msub = (m_new == -INFINITY) ? 0.f : m_new;   // a fully masked chunk (len <= 160 in chunk 1) adds zeros
        const float alpha = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m_run - msub);
        float ps = 0.f;
        o[0] *= alpha;
        o[1] *= alpha;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int jp = 5 * ch + i;
            f32x4 p0, p1;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                if constexpr ((AMUSE_FABL & 8) != 0) {
                    p0[m] = st[2 * i][m];
                    p1[m] = st[2 * i + 1][m];
                } else {
                    p0[m] = __builtin_amdgcn_exp2f(st[2 * i][m] - msub);
                    p1[m] = __builtin_amdgcn_exp2f(st[2 * i + 1][m] - msub);
                }
            }
            ps += ((p0[0] + p0[1]) + (p0[2] + p0[3])) + ((p1[0] + p1[1]) + (p1[2] + p1[3]));
            const bf16x8 pb = pack_bf16(p0, p1);
            const bf16x8 v0 = __builtin_bit_cast(bf16x8, Vt[((jp * 2 + 0) * 16 + r) * 4 + g]);
            const bf16x8 v1 = __builtin_bit_cast(bf16x8, Vt[((jp * 2 + 1) * 16 + r) * 4 + g]);
            o[0] = mfma_bf16(v0, pb, o[0]);   // O^T[d][i] += sum_key V[key][d] P[i][key]
            o[1] = mfma_bf16(v1, pb, o[1]);
        }
        l_run = l_run * alpha + ps;
        m_run = m_new;
    }
    const float inv = __builtin_amdgcn_rcpf(allreduce_g_sum(l_run));
    return pack_bf16(o[0] * inv, o[1] * inv);
}

// MODE 0: input block (push the skip), 1: middle block, 2: output block (skip linear first)
template <int MODE>
__device__ __forceinline__ void decoder_block(f32x4 (&x)[NT][kTiles], Ring& rg, const VaeFusedArgs& a, int blk, int b,
                                              const float* pv, const float* pv_next_src, float* pv_next_dst, char* kv,
                                              uint4* skipbuf, int len, int wave, int lane) {
    const int g = lane >> 4, r = lane & 15;
    bf16x8 xb[NT][4];
    if constexpr (MODE == 2) {
        // x = linear_blocks[blk - 5](cat(x, xs.pop()))   (cross_attention.py:118-120); the popped skip comes back from
        // global memory as the packed operands this wave stored after input block 8 - blk
        pack_rows(xb, x);
        const float* sb = a.pvec + PV_SKIP_B + (blk - 5) * kD;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int t = 0; t < kTiles; ++t) x[j][t] = ld4(sb + 16 * t + 4 * g);
        gemm5<kTiles, 4, false, 0>(x, xb, rg);
        const uint4* sk = skipbuf + (size_t)(8 - blk) * (20 * 4 * 64);
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c) xb[j][c] = __builtin_bit_cast(bf16x8, sk[((wave + 4 * j) * 4 + c) * 64 + lane]);
        gemm5<kTiles, 4, false, 32>(x, xb, rg);
    }
    // ---------------- self-attention (cross_attention.py:323-330): x = norm1(x + out_proj(softmax(q k^T) v))
    pack_rows(xb, x);
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int t = 0; t < kTiles; ++t) x[j][t] += ld4(pv + PV_OUT_B + 16 * t + 4 * g);
    constexpr float kQScale = 0.17677669529663687f * 1.44269504088896340736f;  // 1/sqrt(32) * log2(e): softmax in exp2
#pragma unroll 1
    for (int h = 0; h < kHeads; ++h) {
        char* buf = kv + (h & 1) * kKvBytes;
        uint4* Kb = reinterpret_cast<uint4*>(buf);
        char* Vt = buf + kKeyRows * 64;
        {   // k, v of this head for the wave's rows -> LDS fragment images
            f32x4 kk[NT][2], vv[NT][2];
            const f32x4 bk0 = ld4(pv + PV_IN_B + kD + 32 * h + 4 * g), bk1 = ld4(pv + PV_IN_B + kD + 32 * h + 16 + 4 * g);
            const float bv0 = pv[PV_IN_B + 2 * kD + 32 * h + r], bv1 = pv[PV_IN_B + 2 * kD + 32 * h + 16 + r];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                kk[j][0] = bk0; kk[j][1] = bk1;
                vv[j][0] = splat4(bv0); vv[j][1] = splat4(bv1);
            }
            // stream: per k-pair c: k tiles (2), v tiles (2)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
#pragma unroll
                for (int o = 0; o < 2; ++o) {
                    const bf16x8 wf = take(rg, 4 * c + o);
#pragma unroll
                    for (int j = 0; j < NT; ++j) kk[j][o] = mfma_bf16(wf, xb[j][c], kk[j][o]);
                }
#pragma unroll
                for (int o = 0; o < 2; ++o) {
                    const bf16x8 wf = take(rg, 4 * c + 2 + o);
#pragma unroll
                    for (int j = 0; j < NT; ++j) vv[j][o] = mfma_bf16(xb[j][c], wf, vv[j][o]);  // operand-swapped: V^T
                }
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int tile = wave + 4 * j;
                const bool ok = 16 * tile + r < kFrames;
                const uint4 kf = __builtin_bit_cast(uint4, pack_bf16(kk[j][0], kk[j][1]));
                Kb[(16 * tile + r) * 4 + g] = ok ? kf : uint4{0u, 0u, 0u, 0u};
                // V^T: lane (g, d) holds V[row 4 g + m][16 td + d]; rows beyond the clip are zeroed (0 x p stays 0)
#pragma unroll
                for (int td = 0; td < 2; ++td) {
                    f32x4 v = vv[j][td];
#pragma unroll
                    for (int m = 0; m < 4; ++m) v[m] = (16 * tile + 4 * g + m < kFrames) ? v[m] : 0.f;
                    *reinterpret_cast<uint2*>(Vt + (((tile >> 1) * 2 + td) * 16 + r) * 64 + g * 16 + (tile & 1) * 8) = f32_to_bf16x4(v);
                }
            }
        }
        __syncthreads();
        if (h == 0 && pv_next_src) {   // next block's small parameters -> the other LDS slot (free since the last barrier)
            for (int i = threadIdx.x; i < PV_BLOCK / 4; i += 256) st4(pv_next_dst + 4 * i, ld4(pv_next_src + 4 * i));
        }
        bf16x8 qb[NT];
        {
            f32x4 q[NT][2];
            const f32x4 bq0 = ld4(pv + PV_IN_B + 32 * h + 4 * g), bq1 = ld4(pv + PV_IN_B + 32 * h + 16 + 4 * g);
#pragma unroll
            for (int j = 0; j < NT; ++j) { q[j][0] = bq0; q[j][1] = bq1; }
            gemm5<2, 4, false, 16>(q, xb, rg);
#pragma unroll
            for (int j = 0; j < NT; ++j) qb[j] = pack_bf16(q[j][0] * kQScale, q[j][1] * kQScale);
        }
        bf16x8 ob[NT][1];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            ob[j][0] = attend(Kb, reinterpret_cast<const uint4*>(Vt), qb[j], len, g, r);
            __builtin_amdgcn_sched_barrier(0);
        }
        gemm5<kTiles, 1, false, 24>(x, ob, rg);   // out_proj, k-slice of head h, accumulated into the residual
    }
    const float* ca = a.ca + ((size_t)b * kLayers + blk) * kD;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        layer_norm_rows<true>(x[j], pv + PV_LN1_W, pv + PV_LN1_B, g);
        // cross-attention onto the single latent token == per-clip constant; x = norm2(x + ca)  (cross_attention.py:331-337)
#pragma unroll
        for (int t = 0; t < kTiles; ++t)
            if constexpr (!(AMUSE_FABL & 16)) x[j][t] += ld4(ca + 16 * t + 4 * g);
        layer_norm_rows<true>(x[j], pv + PV_LN2_W, pv + PV_LN2_B, g);
    }
    // ---------------- FFN (cross_attention.py:338-340): x = norm3(x + linear2(gelu(linear1(x)))), 16 chunks of 32 hidden
    pack_rows(xb, x);
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int t = 0; t < kTiles; ++t) x[j][t] += ld4(pv + PV_L2_B + 16 * t + 4 * g);
#pragma unroll 1
    for (int c = 0; c < ((AMUSE_FABL & 4) ? 0 : 16); ++c) {
        f32x4 hid[NT][2];
        const f32x4 b0 = ld4(pv + PV_L1_B + 32 * c + 4 * g), b1 = ld4(pv + PV_L1_B + 32 * c + 16 + 4 * g);
#pragma unroll
        for (int j = 0; j < NT; ++j) { hid[j][0] = b0; hid[j][1] = b1; }
        gemm5<2, 4, false, 0>(hid, xb, rg);
        bf16x8 hb[NT][1];
#pragma unroll
        for (int j = 0; j < NT; ++j) hb[j][0] = pack_bf16(gelu_poly4(hid[j][0]), gelu_poly4(hid[j][1]));
        gemm5<kTiles, 1, false, 8>(x, hb, rg);
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) layer_norm_rows<true>(x[j], pv + PV_LN3_W, pv + PV_LN3_B, g);
    if constexpr (MODE == 0) {   // xs.append(x): packed operands of the skip linear that pops them
        uint4* sk = skipbuf + (size_t)blk * (20 * 4 * 64);
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c)
                sk[((wave + 4 * j) * 4 + c) * 64 + lane] = __builtin_bit_cast(uint4, pack_bf16(x[j][2 * c], x[j][2 * c + 1]));
    }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_vae_fused(VaeFusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* kv = smem;
    float* pvl = reinterpret_cast<float*>(smem + kMainBytes);   // [2][PV_BLOCK]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, r = lane & 15;
    const int b = blockIdx.x;
    const int len = a.lengths ? a.lengths[b] : kFrames;
    uint4* skipbuf = a.skip + (size_t)b * (4 * 20 * 4 * 64);
    Ring rg;
    ring_fill(rg, a.wstream + lane);
    for (int i = threadIdx.x; i < PV_BLOCK / 4; i += 256) st4(pvl + 4 * i, ld4(a.pvec + 4 * i));
    // queries = zeros + query_pos_decoder.pe[:300]  (vae.py:220,258)
    f32x4 x[NT][kTiles];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int frame = 16 * (wave + 4 * j) + r;
#pragma unroll
        for (int t = 0; t < kTiles; ++t)
            x[j][t] = frame < kFrames ? ld4(a.pe + (size_t)frame * kD + 16 * t + 4 * g) : splat4(0.f);
    }
    __syncthreads();
#pragma unroll 1
    for (int blk = 0; blk < 4; ++blk)
        decoder_block<0>(x, rg, a, blk, b, pvl + (blk & 1) * PV_BLOCK, a.pvec + (blk + 1) * PV_BLOCK,
                         pvl + ((blk + 1) & 1) * PV_BLOCK, kv, skipbuf, len, wave, lane);
    decoder_block<1>(x, rg, a, 4, b, pvl, a.pvec + 5 * PV_BLOCK, pvl + PV_BLOCK, kv, skipbuf, len, wave, lane);
#pragma unroll 1
    for (int blk = 5; blk < kLayers; ++blk)
        decoder_block<2>(x, rg, a, blk, b, pvl + (blk & 1) * PV_BLOCK, blk + 1 < kLayers ? a.pvec + (blk + 1) * PV_BLOCK : nullptr,
                         pvl + ((blk + 1) & 1) * PV_BLOCK, kv, skipbuf, len, wave, lane);
    // ---------------- decoder.norm -> final_layer (333 outputs in 24 tiles) -> rotation epilogue, one row tile at a time
    __syncthreads();   // the K/V images become the staging tiles
    float* fst = reinterpret_cast<float*>(smem + wave * kStageBytes);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        layer_norm_rows<true>(x[j], a.pvec + PV_FINAL_W, a.pvec + PV_FINAL_B, g);
        bf16x8 xb1[1][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) xb1[0][c] = pack_bf16(x[j][2 * c], x[j][2 * c + 1]);
        const int tile = wave + 4 * j;
        const int frame = 16 * tile + r;
        const bool keep = frame < kFrames && frame < len;   // output[~mask.T] = 0 (vae.py:274)
#pragma unroll
        for (int half = 0; half < 2; ++half) {   // 12 output tiles at a time (stream: per half, k-pair outer)
            f32x4 f[12];
#pragma unroll
            for (int o = 0; o < 12; ++o) f[o] = ld4(a.final_bias + 16 * (12 * half + o) + 4 * g);
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int o = 0; o < 12; ++o) f[o] = mfma_bf16(take(rg, 12 * c + o), xb1[0][c], f[o]);   // 48 units = 3 ring turns
#pragma unroll
            for (int o = 0; o < 12; ++o) st4(fst + r * kFeatStride + 16 * (12 * half + o) + 4 * g, keep ? f[o] : splat4(0.f));
        }
        __syncthreads();
        const int rows_here = min(16, kFrames - 16 * tile);   // <= 0 for the padding tile
        const size_t row0 = (size_t)b * kFrames + 16 * tile;
        if (a.feats_out) {
            for (int i = lane; i < rows_here * kFeats; i += 64) {
                const int rr = i / kFeats, c = i - rr * kFeats;
                a.feats_out[(row0 + rr) * kFeats + c] = fst[rr * kFeatStride + c];
            }
        }
        if (a.poses_out) {
            for (int i = lane; i < rows_here * kJoints; i += 64) {
                const int rr = i / kJoints, jn = i - rr * kJoints;
                float aa[3];
                rot6d_to_axis_angle(fst + rr * kFeatStride + 6 * jn, a.quat_mode, aa);
                float* dst = a.poses_out + ((row0 + rr) * kJoints + jn) * 3;
                dst[0] = aa[0]; dst[1] = aa[1]; dst[2] = aa[2];
            }
        }
        if (a.trans_out) {
            for (int i = lane; i < rows_here * 3; i += 64) {
                const int rr = i / 3, c = i - rr * 3;
                a.trans_out[(row0 + rr) * 3 + c] = fst[rr * kFeatStride + 330 + c];
            }
        }
        __syncthreads();
    }
}

}  // namespace

hipError_t launch_vae_fused(const VaeFusedArgs& a, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_vae_fused), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           kVaeFusedLdsBytes);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(k_vae_fused, dim3(a.B), dim3(256), kVaeFusedLdsBytes, stream, a);
    return hipGetLastError();
}

}  // namespace amuse
