// Persistent sampling kernel: the WHOLE T-step denoising loop of
// PretrainedLPDM_v1.diffusion_backward (reference models/latent_diffusion/infer_ldm.py:137-161) in one
// launch - Denoiser.forward (denoiser.py:135-204: 5-token skip-transformer encoder,
// cross_attention.py:41-64,259-272) and the diffusers scheduler update fused, latent + activations
// resident in registers for all T steps, no inter-workgroup communication.
//
// Work decomposition (gfx950):
//   workgroup = 4 wavefronts = one tile of G clips (G*S <= 16 token rows, S = 5 tokens per clip).
//   All four waves hold the same [16 x 128] residual stream in row-lane layout (amuse_dev.hpp).
//   wave h owns attention head h end to end:  q_h,k_h,v_h = its 6 of the 24 in_proj output tiles;
//   scores / softmax / PV are three-or-so MFMAs entirely inside the wave (S^T = K.Q^T, O^T = V^T.P^T
//   so that softmax runs along registers + two xor-shuffles and the result lands in row-lane
//   layout); out_proj is split-K over heads, FFN1 is split over output features (wave w owns hidden
//   features 128w..128w+127) and FFN2 is split-K over exactly those features, so a block needs only
//   TWO workgroup barriers (the two split-K combines through LDS).  The U-Net skip linears are
//   split-K as well (wave w takes k-tiles 4w..4w+3 of cat(x, skip)).
//   Each wave streams only its own quarter of the weights, as one sequential pass per step over a
//   host-packed stream of 1 KiB units (L2-resident: 7.7 MB fp32 / 3.8 MB bf16 for the network).
#include "amuse_dev.hpp"
#include "amuse_kernels.hpp"

namespace amuse {

namespace {

// NC weight-stream units are issued after the score MFMA and again after the softmax (ring_issue, amuse_dev.hpp)
template <int PREC, int NC, int IPH>
__device__ __forceinline__ void attention_head(const f32x4 (&q)[2], const f32x4 (&k)[2], const f32x4 (&v)[2],
                                               const bool (&kvalid)[4], f32x4 (&o)[2], WRing<kRing>& rg) {
    // S^T[j][i] = sum_d K[j][d] Q[i][d]  ->  lane (g, i) holds S[i][4 g + m]
    constexpr bool EXACT = (PREC != PREC_BF16);   // fp32 / fp32x: libm exp and IEEE division, as torch.softmax
    f32x4 st = splat4(0.f);
    if constexpr (PREC == PREC_F32) {
#pragma unroll
        for (int td = 0; td < 2; ++td)
#pragma unroll
            for (int m = 0; m < 4; ++m) st = mfma_f32(k[td][m], q[td][m], st);
    } else if constexpr (PREC == PREC_F16X2) {
        const F16Pair ks = split_f16(k[0], k[1]), qs = split_f16(q[0], q[1]);
        st = mfma_f16(ks.lo, qs.hi, st);
        st = mfma_f16(ks.hi, qs.lo, st);
        st = mfma_f16(ks.hi, qs.hi, st);
    } else {
        st = mfma_bf16(pack_bf16(k[0], k[1]), pack_bf16(q[0], q[1]), st);
    }
    ring_issue<NC, kRing, IPH % kRing>(rg);
    float mx = -INFINITY;
#pragma unroll
    for (int m = 0; m < 4; ++m) mx = kvalid[m] ? fmaxf(mx, st[m]) : mx;
    mx = allreduce_g_max(mx);
    f32x4 p;
    float sum = 0.f;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const float e = EXACT ? expf(st[m] - mx) : __builtin_amdgcn_exp2f(1.44269504088896340736f * (st[m] - mx));
        p[m] = kvalid[m] ? e : 0.f;
        sum += p[m];
    }
    sum = allreduce_g_sum(sum);
    {
        const float inv = EXACT ? 0.f : __builtin_amdgcn_rcpf(sum);
#pragma unroll
        for (int m = 0; m < 4; ++m) p[m] = EXACT ? p[m] / sum : p[m] * inv;
    }
    ring_issue<NC, kRing, (IPH + NC) % kRing>(rg);
    // O^T[d][i] = sum_j V[j][d] P[i][j]; v is feature-lane: lane (g, d) holds V[4 g + m][d]
#pragma unroll
    for (int td = 0; td < 2; ++td) {
        o[td] = splat4(0.f);
        if constexpr (PREC == PREC_F32) {
#pragma unroll
            for (int m = 0; m < 4; ++m) o[td] = mfma_f32(v[td][m], p[m], o[td]);
        } else if constexpr (PREC == PREC_F16X2) {
            const F16Pair vs = split_f16(v[td], splat4(0.f)), ps = split_f16(p, splat4(0.f));
            o[td] = mfma_f16(vs.lo, ps.hi, o[td]);
            o[td] = mfma_f16(vs.hi, ps.lo, o[td]);
            o[td] = mfma_f16(vs.hi, ps.hi, o[td]);
        } else {
            o[td] = mfma_bf16(pack_bf16(v[td], splat4(0.f)), pack_bf16(p, splat4(0.f)), o[td]);
        }
    }
}

// optional phase timeline (s_memtime stamps by lane 0 of every wave of workgroup 0 during ONE step);
// compiled out of the production instantiation
struct Prof {
    unsigned long long* out;
    int idx;
    bool on;
};
template <bool PROF>
__device__ __forceinline__ void stamp(Prof& pf) {
    if constexpr (PROF) {
        if (pf.on) pf.out[pf.idx++] = __builtin_readcyclecounter();
    }
}

// One TransformerEncoderLayer.forward_post (cross_attention.py:259-272) on the row-lane tile x.
template <int PREC, bool PROF>
__device__ __forceinline__ void encoder_block(f32x4 (&x)[kTiles], WRing<kRing>& rg, const float* pv,
                                              const bool (&kvalid)[4], char* comb, int wave, int lane,
                                              bool next_has_skip, Prof& pf) {
    const int g = lane >> 4, r = lane & 15;
    // ring phases of the five GEMMs of a block (compile-time; a block consumes a whole number of revolutions)
    constexpr int U_QK = gemm_units(PREC, 4, kTiles), U_V = gemm_units(PREC, 2, kTiles);
    constexpr int U_OUT = gemm_units(PREC, kTiles, 2), U_FF = gemm_units(PREC, kTiles, kTiles);
    constexpr int P_QK = 0, P_V = (P_QK + U_QK) % kRing, P_OUT = (P_V + U_V) % kRing;
    constexpr int P_F1 = (P_OUT + U_OUT) % kRing, P_F2 = (P_F1 + U_FF) % kRing;
    static_assert((P_F2 + U_FF) % kRing == 0, "a block must leave the ring at phase 0");
    constexpr bool FAST = (PREC == PREC_BF16);
    // GELU: libm erff in both parity modes.  (Measured on one box, 256 clips, fp32x: erff 96.3 ms per 1000 steps; the
    // Abramowitz-Stegun form on the hardware rcp / exp2 that the 4-wave bf16 kernel uses 102.0 ms:
    // its two quarter-rate transcendentals per element sit on the wave's critical path, erff's plain fma chains do not.)
    constexpr bool FAST_ACT = (PREC == PREC_BF16);
    // bf16 mode decouples load issue from consumption (ring_issue): per block the ring (full on entry = in_proj +
    // out_proj units) is re-armed 8+8+8 units around the attention, 8 during combine 1 (-> holds all of linear1),
    // 32 inside linear1 (immediately: linear2 needs them next), 32 during combine 2 (-> next block's first 32).
    // fp32 mode is MFMA-issue-bound and keeps the simple re-arm-at-consumption ring.
    constexpr bool DELAY = (PREC == PREC_BF16);
    constexpr int A8 = DELAY ? 8 : 0;
    // fp32x is bound by the CU's 64 B/clk load path (7.6 MB per step = 119 k cycles): GEMMs that re-arm as they consume run
    // at that rate, and the path idles through the attention arithmetic and the two combines (55 k cycles per step).
    // Letting the GEMM in front of each of those phases leave its slots empty and the phase re-arm them was measured SLOWER
    // (105.6 against 102.0 ms per 1000 steps, docs/history.md): with one wave per SIMD a wave blocked at load issue inside a combine
    // arrives late at its barriers, so issue time and chain time still add up - the way out is the 8-wave
    // role split (k_sampler8x.hip), not the placement of the issue.
    // Small parameters (LDS) are read one phase ahead of their use; biases are added AFTER the GEMM that
    // they belong to, so no LDS round trip sits in front of a GEMM's first MFMA.
    f32x4 b_qk[4];
    float b_v[2];
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        b_qk[o] = ld4(pv + PV_IN_B + 16 * (2 * wave + o) + 4 * g);
        b_qk[2 + o] = ld4(pv + PV_IN_B + kD + 16 * (2 * wave + o) + 4 * g);
        b_v[o] = pv[PV_IN_B + 2 * kD + 16 * (2 * wave + o) + r];
    }
    // ---- in_proj: q_h, k_h (row-lane) and v_h (feature-lane) for head h = wave
    f32x4 qk[4], v[2];
#pragma unroll
    for (int o = 0; o < 4; ++o) qk[o] = splat4(0.f);
    v[0] = v[1] = splat4(0.f);
    if constexpr (PREC == PREC_F16X2) {   // one split of the rows serves q, k and v
        F16Pair xs[kTiles / 2];
        split_rows<kTiles>(xs, x);
        gemm_ring_s<4, kTiles / 2, false, kRing, P_QK>(qk, xs, rg);
        gemm_ring_s<2, kTiles / 2, true, kRing, P_V, true>(v, xs, rg);
    } else {
        gemm_ring<PREC, 4, kTiles, false, kRing, P_QK, !DELAY>(qk, x, rg);
        gemm_ring<PREC, 2, kTiles, true, kRing, P_V, !DELAY>(v, x, rg);
    }
    ring_issue<A8, kRing, 0>(rg);
    stamp<PROF>(pf);  // 1: in_proj done
    const float scaling = 0.17677669529663687f;  // sqrt(1/32): q * scaling (F.multi_head_attention_forward)
    f32x4 q[2] = {(qk[0] + b_qk[0]) * scaling, (qk[1] + b_qk[1]) * scaling};
    f32x4 k[2] = {qk[2] + b_qk[2], qk[3] + b_qk[3]};
    v[0] += splat4(b_v[0]);
    v[1] += splat4(b_v[1]);
    f32x4 o[2];
    attention_head<PREC, A8, 8>(q, k, v, kvalid, o, rg);
    stamp<PROF>(pf);  // 2: attention done
    // ---- out_proj, split-K over heads; combine; residual; LayerNorm1
    f32x4 part[kTiles];
#pragma unroll
    for (int t = 0; t < kTiles; ++t) part[t] = splat4(0.f);
    gemm_ring<PREC, kTiles, 2, false, kRing, P_OUT, !DELAY>(part, o, rg);
    stamp<PROF>(pf);  // 3: out_proj partial done
    // x = LN1(x + sum_w part + b_out): reduce-scatter / LayerNorm / all-gather (amuse_dev.hpp combine_rs)
    combine_rs<true, FAST, A8, kRing, 24>(part, x, true, pv + PV_OUT_B, pv + PV_LN1_W, pv + PV_LN1_B, comb, wave, lane, &rg);
    stamp<PROF>(pf);  // 4: combine 1 + LN1 done
    stamp<PROF>(pf);  // 5: (kept for timeline compatibility)
    // ---- FFN in four interleaved quarters: linear1 for 2 of this wave's 8 hidden tiles -> bias + GELU ->
    // linear2 split-K contribution of exactly those 32 hidden features.  The weight stream is packed in the same
    // order, so the 32-slot ring (holding quarters 0,1 on entry) is re-armed with quarters 2,3 while quarters 0,1
    // compute, and the GELU VALU work sits between the MFMA/load bursts instead of after all of them.
    constexpr int U_Q = gemm_units(PREC, 2, kTiles);  // units of one half-quarter (= gemm_units(PREC, 8, 2))
    static_assert(U_Q == gemm_units(PREC, kTiles, 2) && P_F1 == 0, "FFN quarter phases");
#pragma unroll
    for (int t = 0; t < kTiles; ++t) part[t] = splat4(0.f);
    // software-pipelined by one quarter: linear1 of quarter q+1 is issued BEFORE the GELU of quarter q, so the
    // matrix pipe works through it while the VALU evaluates the erf (one wave per SIMD: nothing else would fill
    // that latency).  Stream order: F1q0 F1q1 F2q0 F1q2 F2q1 F1q3 F2q2 F2q3 (8 / 16 units each).
    f32x4 hq[4][2];
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) hq[qd][0] = hq[qd][1] = splat4(0.f);
    auto gelu_quarter = [&](int qd) {
        const f32x4 b1a = ld4(pv + PV_L1_B + 16 * (kTiles * wave + 2 * qd) + 4 * g);
        const f32x4 b1b = ld4(pv + PV_L1_B + 16 * (kTiles * wave + 2 * qd + 1) + 4 * g);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const float h0 = hq[qd][0][m] + b1a[m], h1 = hq[qd][1][m] + b1b[m];
            hq[qd][0][m] = FAST_ACT ? gelu_erf_fast(h0) : gelu_erf(h0);
            hq[qd][1][m] = FAST_ACT ? gelu_erf_fast(h1) : gelu_erf(h1);
        }
    };
    // fp32: always re-arm at consumption; bf16: the first four unit groups re-arm (with the last four), the last
    // four leave the ring to be re-armed during combine 2
    if constexpr (PREC == PREC_F16X2) {   // the four linear1 quarters read the same rows: split them once
        F16Pair xs[kTiles / 2];
        split_rows<kTiles>(xs, x);
        gemm_ring_s<2, kTiles / 2, false, kRing, (0 * U_Q) % kRing>(hq[0], xs, rg);         // F1 q0
        gemm_ring_s<2, kTiles / 2, false, kRing, (1 * U_Q) % kRing>(hq[1], xs, rg);         // F1 q1
        gelu_quarter(0);
        gemm_ring<PREC, kTiles, 2, false, kRing, (2 * U_Q) % kRing, true>(part, hq[0], rg); // F2 q0
        stamp<PROF>(pf);
        gemm_ring_s<2, kTiles / 2, false, kRing, (3 * U_Q) % kRing>(hq[2], xs, rg);         // F1 q2
        gelu_quarter(1);
        gemm_ring<PREC, kTiles, 2, false, kRing, (4 * U_Q) % kRing, true>(part, hq[1], rg); // F2 q1
        stamp<PROF>(pf);
        gemm_ring_s<2, kTiles / 2, false, kRing, (5 * U_Q) % kRing>(hq[3], xs, rg);         // F1 q3
    } else {
    gemm_ring<PREC, 2, kTiles, false, kRing, (0 * U_Q) % kRing, true>(hq[0], x, rg);        // F1 q0
    gemm_ring<PREC, 2, kTiles, false, kRing, (1 * U_Q) % kRing, true>(hq[1], x, rg);        // F1 q1
    gelu_quarter(0);
    gemm_ring<PREC, kTiles, 2, false, kRing, (2 * U_Q) % kRing, true>(part, hq[0], rg);     // F2 q0
    stamp<PROF>(pf);
    gemm_ring<PREC, 2, kTiles, false, kRing, (3 * U_Q) % kRing, true>(hq[2], x, rg);        // F1 q2
    gelu_quarter(1);
    gemm_ring<PREC, kTiles, 2, false, kRing, (4 * U_Q) % kRing, !DELAY>(part, hq[1], rg);   // F2 q1
    stamp<PROF>(pf);
    gemm_ring<PREC, 2, kTiles, false, kRing, (5 * U_Q) % kRing, !DELAY>(hq[3], x, rg);      // F1 q3
    }
    gelu_quarter(2);
    gemm_ring<PREC, kTiles, 2, false, kRing, (6 * U_Q) % kRing, !DELAY>(part, hq[2], rg);   // F2 q2
    stamp<PROF>(pf);
    gelu_quarter(3);
    gemm_ring<PREC, kTiles, 2, false, kRing, (7 * U_Q) % kRing, !DELAY>(part, hq[3], rg);   // F2 q3
    stamp<PROF>(pf);  // 9: FFN done (linear2 partial)
    // x = LN2(x + sum_w part + b_l2).  bf16: the ring is re-armed meanwhile with the next block's first units - its
    // in_proj + out_proj (32) or, ahead of an output block, only the 16 skip-linear units (slots 16..31 stay empty;
    // the skip combine re-arms all 32)
    if constexpr (DELAY) {
        if (next_has_skip)
            combine_rs<true, FAST, 16, kRing, 0>(part, x, true, pv + PV_L2_B, pv + PV_LN2_W, pv + PV_LN2_B, comb, wave, lane, &rg);
        else
            combine_rs<true, FAST, 32, kRing, 0>(part, x, true, pv + PV_L2_B, pv + PV_LN2_W, pv + PV_LN2_B, comb, wave, lane, &rg);
    } else {
        combine_rs<true, FAST>(part, x, true, pv + PV_L2_B, pv + PV_LN2_W, pv + PV_LN2_B, comb, wave, lane);
    }
    stamp<PROF>(pf);  // 9: combine 2 + LN2 done
    stamp<PROF>(pf);  // 10: LN2 done
}

__device__ __forceinline__ void store_tap(float* tap, int slot, const f32x4 (&x)[kTiles], int g, int r) {
#pragma unroll
    for (int t = 0; t < kTiles; ++t) st4(tap + ((size_t)slot * 16 + r) * kD + 16 * t + 4 * g, x[t]);
}

template <int PREC, bool PROF>
__global__ __launch_bounds__(256, 1) void k_sample(SampleArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* comb = smem;                                           // split-K combine buffers (kCombBytes)
    f32x4* skip = reinterpret_cast<f32x4*>(smem + kCombBytes);  // [4][8 tiles][64 lanes]
    float* pvl = reinterpret_cast<float*>(smem + kCombBytes + kSkipBytes);  // small params, compact (kEncPv / block)
    for (int i = threadIdx.x; i < kLayers * kEncPv / 4; i += 256) {
        const int blk = (4 * i) / kEncPv, off = 4 * i - blk * kEncPv;
        st4(pvl + 4 * i, ld4(a.pvec + blk * PV_BLOCK + off));
    }
    for (int i = threadIdx.x; i < (4 * kD + 2 * kD) / 4; i += 256) st4(pvl + kLayers * kEncPv + 4 * i, ld4(a.pvec + PV_SKIP_B + 4 * i));
    const float* pv_skip = pvl + kLayers * kEncPv;
    const float* pv_final = pv_skip + 4 * kD;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, r = lane & 15;
    const int S = a.S, R = S * a.G;
    const int cl = r / S, tok = r - cl * S;
    const long clip = (long)blockIdx.x * a.G + cl;
    const bool valid = (r < R) && (clip < (long)a.B);
    const bool is_lat = valid && tok == 0;
    // attention key mask for this lane's query row: keys j = 4 g + m of the SAME clip; padding rows
    // attend to themselves only (keeps them finite, they never touch valid rows)
    bool kvalid[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int j = 4 * g + m;
        kvalid[m] = valid ? (j < R && (j / S) == cl) : (j == r);
    }
    f32x4 stat[kTiles], lat[kTiles];
#pragma unroll
    for (int t = 0; t < kTiles; ++t) {
        const int f = 16 * t + 4 * g;
        stat[t] = splat4(0.f);
        lat[t] = splat4(0.f);
        if (valid) {
            if (tok == 0) {
                stat[t] = ld4(a.pe0 + f);
                lat[t] = a.x_init ? ld4(a.x_init + (size_t)clip * kD + f)
                                  : counter_normal4(a.seed, a.clip0 + (uint64_t)clip, 0u, (uint32_t)(4 * t + g), 0u);
            } else if (tok >= 2) {
                stat[t] = ld4(a.cond_tok + ((size_t)clip * (S - 2) + (tok - 2)) * kD + f);
            }
        }
    }
    const bool tap = a.tap_out != nullptr && blockIdx.x == 0 && wave == 0;
    const uint4* wbase = a.wstream + (size_t)wave * (a.wave_units + kRing) * 64 + lane;
    WRing<kRing> rg;
    ring_fill(rg, wbase);
    Prof pf{a.prof_out ? a.prof_out + (size_t)wave * kProfStamps : nullptr, 0, false};
    // `stat` of the time-token rows (tok == 1) holds the token of the NEXT step, fetched a whole step ahead
    if (valid && tok == 1) {
#pragma unroll
        for (int t = 0; t < kTiles; ++t)
            stat[t] = ld4((a.time_tok_clip ? a.time_tok_clip + (size_t)clip * kD : a.time_tok) + 16 * t + 4 * g);
    }
#pragma unroll 1
    for (int step = 0; step < a.T; ++step) {
        // ---- token assembly (denoiser.py:174,180-181)
        f32x4 x[kTiles];
#pragma unroll
        for (int t = 0; t < kTiles; ++t) x[t] = !valid ? splat4(0.f) : ((tok == 0) ? lat[t] + stat[t] : stat[t]);
        if (valid && tok == 1) {
            const float* tt = a.time_tok + (size_t)(step + 1 < a.T ? step + 1 : step) * kD;
#pragma unroll
            for (int t = 0; t < kTiles; ++t) stat[t] = ld4(tt + 16 * t + 4 * g);
        }
        if (tap && step == 0) store_tap(a.tap_out, 0, x, g, r);
        rg.next = wbase + kRing * 64;  // the ring already holds units 0..R-1 of this step (stream tail = its head)
        if constexpr (PROF) {
            pf.on = a.prof_out != nullptr && blockIdx.x == 0 && lane == 0 && step == a.prof_step;
            pf.idx = 0;
        }
        stamp<PROF>(pf);  // 0: step start (token assembly done)
        // ---- SkipTransformerEncoder.forward (cross_attention.py:41-64)
#pragma unroll 1
        for (int blk = 0; blk < kLayers; ++blk) {
            if (blk >= 5) {  // x = Linear(cat(x, skips.pop())), split-K: wave w takes k-tiles 4w..4w+3
                f32x4 src[4];
                const f32x4* sk = skip + (size_t)(8 - blk) * kTiles * 64;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (wave < 2) src[i] = (wave == 1) ? x[4 + i] : x[i];
                    else src[i] = sk[(4 * (wave - 2) + i) * 64 + lane];
                }
                f32x4 part[kTiles];
#pragma unroll
                for (int t = 0; t < kTiles; ++t) part[t] = splat4(0.f);
                constexpr int U_SK = gemm_units(PREC, kTiles, 4);
                const float* sb = pv_skip + (blk - 5) * kD;
                if constexpr (PREC == PREC_BF16) {
                    // the ring holds just the 16 skip-linear units (slots 0..15); the whole ring is re-armed
                    // during the combine with the block's first 32 units
                    gemm_ring<PREC, kTiles, 4, false, kRing, 0, false>(part, src, rg);
                    combine_rs<false, true, 32, kRing, 0>(part, x, false, sb, nullptr, nullptr, comb, wave, lane, &rg);
                } else {
                    gemm_ring<PREC, kTiles, 4, false, kRing, 0>(part, src, rg);
                    ring_discard<skip_pad_units(PREC), kRing, U_SK % kRing>(rg);
                    combine_rs<false, false>(part, x, false, sb, nullptr, nullptr, comb, wave, lane);
                }
            }
            stamp<PROF>(pf);  // block start (after the skip linear, if any)
            encoder_block<PREC, PROF>(x, rg, pvl + blk * kEncPv, kvalid, comb, wave, lane,
                                      blk >= 4 && blk < kLayers - 1, pf);
            if (blk < 4 && wave == 0) {
                f32x4* sk = skip + (size_t)blk * kTiles * 64;
#pragma unroll
                for (int t = 0; t < kTiles; ++t) sk[t * 64 + lane] = x[t];
            }
            if (tap && step == 0) store_tap(a.tap_out, 1 + blk, x, g, r);
        }
        layer_norm_rows<PREC == PREC_BF16>(x, pv_final, pv_final + kD, g);
        if (tap && step == 0) store_tap(a.tap_out, 10, x, g, r);
        if (a.eps_out && is_lat && wave == 0 && step == a.T - 1) {
#pragma unroll
            for (int t = 0; t < kTiles; ++t) st4(a.eps_out + (size_t)clip * kD + 16 * t + 4 * g, x[t]);
        }
        // ---- scheduler.step (diffusers 0.17.1 DDIM / DDPM; amuse_hip.h amuse_schedule)
        if (!a.no_update) {
            const float* cf = a.coef + (size_t)step * 8;
            const float sb = cf[0], sa = cf[1], c0 = cf[2], cx = cf[3], ce = cf[4], sg = cf[5], clipv = cf[6];
            // ancestral noise z for the latent rows.  In-kernel generation is spread over the whole wave: lane L
            // draws the 4 normals of feature group L % 32 of clip (L / 32) of the tile - ONE Philox call per lane
            // per two clips instead of eight per lane (ceil(G / 2) calls: up to 3 for the G = 5 tiles of con-only
            // conditioning, S = 3) - and the latent-row lanes fetch their 8 groups by bpermute.
            // Same counters (global clip, step, feature group) as before, so values are bit-identical.
            f32x4 zt[kTiles];
#pragma unroll
            for (int t = 0; t < kTiles; ++t) zt[t] = splat4(0.f);
            if (sg != 0.f) {
                if (a.step_noise) {
                    if (is_lat) {
#pragma unroll
                        for (int t = 0; t < kTiles; ++t)
                            zt[t] = ld4(a.step_noise + ((size_t)step * a.B + clip) * kD + 16 * t + 4 * g);
                    }
                } else {
#pragma unroll
                    for (int call = 0; call < 3; ++call) {
                        if (2 * call < a.G) {
                            const int cc = 2 * call + (lane >> 5);
                            const uint64_t gc = a.clip0 + (uint64_t)((long)blockIdx.x * a.G + cc);
                            const f32x4 n = counter_normal4(a.seed, gc, (uint32_t)step, (uint32_t)(lane & 31), 1u);
                            const bool mine = (cl >> 1) == call;
                            const int src = 32 * (cl & 1) + g;
#pragma unroll
                            for (int t = 0; t < kTiles; ++t)
#pragma unroll
                                for (int m = 0; m < 4; ++m) {
                                    const float v = __shfl(n[m], src + 4 * t);
                                    zt[t][m] = mine ? v : zt[t][m];
                                }
                        }
                    }
                }
            }
            constexpr bool FASTU = (PREC == PREC_BF16);  // bf16 mode: reciprocal multiply instead of IEEE division
            const float inv_sa = 1.0f / sa;
            {
// each product and sum rounded on its own, like the scheduler's tensor ops (hipcc contracts even __fmul_rn /
// __fadd_rn pairs into v_fma under its default -ffp-contract=fast)
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
            stamp<PROF>(pf);  // scheduler update done
            if (a.traj_out && is_lat && wave == 0) {
#pragma unroll
                for (int t = 0; t < kTiles; ++t)
                    st4(a.traj_out + ((size_t)step * a.B + clip) * kD + 16 * t + 4 * g, lat[t]);
            }
        }
    }
    if (is_lat && wave == 0) {
#pragma unroll
        for (int t = 0; t < kTiles; ++t) {
            if (a.latents_out) st4(a.latents_out + (size_t)clip * kD + 16 * t + 4 * g, lat[t]);
        }
    }
}

}  // namespace

// The 4-wave kernel serves the fp32 parity mode only: bf16 / fp16 / fp32x sample on the 8-wave kernels (k_sampler8.hip, k_sampler8x.hip).  The template stays
// parametric in PREC (its building blocks in amuse_dev.hpp are shared with k_sampler_dec.hip and k_vae.hip, which instantiate every mode); the bf16 and fp32x
// instantiations of THIS kernel and their agreement tests with the 8-wave kernels are shelved under tools/probes/sampler_4wave/.
hipError_t launch_sample(const SampleArgs& a, int precision, hipStream_t stream) {
    if (precision != PREC_F32) return hipErrorInvalidValue;
    const int tiles = (a.B + a.G - 1) / a.G;
    const dim3 grid(tiles), block(256);
    static DeviceOnce once;
    int dev_;
    if (!once.done(&dev_)) {
        for (const void* k : {reinterpret_cast<const void*>(&k_sample<PREC_F32, false>), reinterpret_cast<const void*>(&k_sample<PREC_F32, true>)}) {
            hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, kSampleLdsBytes);
            if (e != hipSuccess) return e;
        }
        once.set(dev_);
    }
    if (a.prof_out) hipLaunchKernelGGL((k_sample<PREC_F32, true>), grid, block, kSampleLdsBytes, stream, a);
    else hipLaunchKernelGGL((k_sample<PREC_F32, false>), grid, block, kSampleLdsBytes, stream, a);
    return hipGetLastError();
}

}  // namespace amuse
