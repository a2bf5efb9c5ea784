// "Wide" bf16 sampling kernel for large clip batches: the same T-step loop as k_sampler8.hip (reference
// models/latent_diffusion/infer_ldm.py:137-161; Denoiser.forward denoiser.py:135-204; encoder blocks
// cross_attention.py:41-64,259-272; diffusers scheduler step), decomposed the other way round.
//
// k_sample8 gives ONE 16-row tile (three clips) to a whole CU and splits every GEMM's K dimension over its eight waves: the
// right shape for BASELINE's 256 clips, where nothing but one tile's step latency counts - but its matrix pipes idle 87 % of
// the time (two split-K combines per block), and every CU streams the whole network (3.8 MB) per step for three clips: 6 M
// frames/s is where it saturates, whatever the batch.  With thousands of clips per GPU the step latency no longer matters;
// bytes per clip and MFMA occupancy do.  Here:
//
//   * a WAVE owns a tile for the whole network and all T steps: residual stream, latent and U-Net skip stack in its registers
//     (row-lane layout, amuse_dev.hpp).  Every GEMM is complete inside the wave - no split-K, no combine, no exchange;
//     attention over the tile's 5-token clips, LayerNorm (permlane reductions), GELU, scheduler update and Philox noise too.
//   * the W waves of a workgroup (W = 2, 4: one per SIMD, 512 registers) share ONE weight stream: per step 58 stages of 64
//     units (64 KiB), staged global -> registers -> LDS into two buffers, one stage ahead (a stage is 1.1 k cycles of MFMAs per
//     wave; 64 KiB at the CU's 64 B/clk take 1.0 k), one barrier per stage.  A block's small parameters (biases, LayerNorm)
//     ride with the stream as an 8 KiB piece group in front of the block's first stage.
//   * so a CU moves the same 3.8 MB per step as in k_sample8, but for W tiles = 3 W clips, and each of its busy SIMDs issues
//     MFMAs back to back (8 independent accumulators per GEMM, fragments read from LDS three units ahead).
//
// Selected by amuse_api.hip for bf16 sampling from 1024 clips per launch (amuse_set_sampler_path); below that k_sample8's
// latency wins.  Results differ from k_sample8's by summation order only (same operand roundings, same polynomial GELU).
#include "amuse_dev.hpp"
#include "amuse_kernels.hpp"

namespace amuse {
namespace {

#ifndef AMUSE_WABL
#define AMUSE_WABL 0   // timing ablations (tools/build_variant.sh): 1 no weight loads, 2 no MFMAs / fragment reads, 4 no LDS stage writes, 8 loads may sink
#endif
constexpr int kWStage = kWideStageUnits;          // 64 units
constexpr int kWStageBytes = kWStage * 1024;      // 64 KiB
constexpr int kWPvBytes = kWidePvUnits * 1024;    // 8 KiB: [kEncPv block params | 4 x 128 skip biases (block 5..8: its own) | final LN]
constexpr int kOffStage = 0;
constexpr int kOffPvW = 2 * kWStageBytes;
static_assert(kOffPvW + 3 * kWPvBytes == kSampleWideLdsBytes, "LDS layout");   // three parameter slots: block b uses b % 3
// float offsets inside a block's parameter group
constexpr int PW_SKIP_B = kEncPv;            // [128] bias of the skip linear in front of this block (blocks 5..8)
constexpr int PW_FINAL_W = kEncPv + 128;     // [128], [128] final LayerNorm (every block carries it; block 8's copy is used)
constexpr int PW_FINAL_B = kEncPv + 256;

// The workgroup's share of the weight stream, register-staged: at the start of stage s every wave issues its pieces of
// stage s + 1 as plain global loads into registers (kWStage / W pieces of 1 KiB: 64 VGPRs at W = 4), computes stage s from
// LDS, then writes the pieces into the other LDS buffer and meets the others at the barrier.  (LDS-DMA, the fused decoder's
// transport, delivers ~11 B/clk per CU here - 64 KiB per 5.6 k cycles, measured: a quarter of a stage's MFMA time per KiB
// is all this kernel has - while global_load -> VGPR moves 64 B/clk.)
template <int W>
struct WStream {
    static constexpr int NP = kWStage / W;                          // weight pieces per wave and stage
    static constexpr int NPV = (kWidePvUnits + W - 1) / W;          // parameter pieces per wave and block
    const uint4* src;    // lane-offset address of the next piece group to fetch
    const uint4* base;   // start of the step's stream (lane-offset)
    const uint4* end;    // one past the step's stream
    char* ring;          // stage buffers (generic pointer) + lane * 16
    char* pvbase;        // parameter slots + lane * 16
    uint4 st[NP], pvst[NPV];
    int wave, wbuf, rbuf, pv_pending;

    // issue the loads of the next stage (and, in front of a block's first stage, of the block's parameter group for slot pv_slot)
    __device__ __forceinline__ void fetch(bool with_params, int pv_slot) {
        pv_pending = -1;
        if (with_params) {
#pragma unroll
            for (int i = 0; i < NPV; ++i) {
                const int pc = wave + i * W;
                pvst[i] = pc < kWidePvUnits ? src[pc * 64] : uint4{0u, 0u, 0u, 0u};
            }
            src += kWidePvUnits * 64;
            pv_pending = pv_slot;
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) st[i] = (AMUSE_WABL & 1) ? uint4{0u, 0u, 0u, 0u} : src[(wave + i * W) * 64];
        if (!(AMUSE_WABL & 8)) __builtin_amdgcn_sched_barrier(0);   // the loads are issued HERE, a stage of compute ahead of their use
        src += kWStage * 64;
        if (src == end) src = base;   // the stream repeats every step
    }
    // write the staged pieces into the other buffer; after the barrier the stage just computed is free and the next one is
    // complete for every wave
    __device__ __forceinline__ void done() {
        char* dst = ring + wbuf * kWStageBytes;
#pragma unroll
        for (int i = 0; i < NP; ++i)
            if (!(AMUSE_WABL & 4)) *reinterpret_cast<uint4*>(dst + (wave + i * W) * 1024) = st[i];
        if (pv_pending >= 0) {
#pragma unroll
            for (int i = 0; i < NPV; ++i) {
                const int pc = wave + i * W;
                if (pc < kWidePvUnits) *reinterpret_cast<uint4*>(pvbase + pv_pending * kWPvBytes + pc * 1024) = pvst[i];
            }
        }
        __syncthreads();
        wbuf ^= 1;
        rbuf ^= 1;
    }
    __device__ __forceinline__ bf16x8 frag(int u) const {
        return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(ring + rbuf * kWStageBytes + u * 1024));
    }
};

// acc[o] (+)= W_o . x over NC k-tile pairs; units U0.. of the current stage (k-pair outer, output tile inner); fragments are
// read PF units ahead of their MFMAs.  SWAP: activations are the A operand (feature-lane result).
template <int W, int NO, int NC, int U0, bool SWAP = false>
__device__ __forceinline__ void gemm1(f32x4 (&acc)[NO], const bf16x8 (&xb)[NC], const WStream<W>& ws) {
    constexpr int NU = NO * NC, PF = NU < 4 ? NU : 4;
    if (AMUSE_WABL & 2) return;
    bf16x8 wf[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) wf[u] = ws.frag(U0 + u);
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int c = u / NO, o = u - c * NO;
        const bf16x8 cur = wf[u % PF];
        if (u + PF < NU) wf[u % PF] = ws.frag(U0 + u + PF);
        acc[o] = SWAP ? mfma_bf16(xb[c], cur, acc[o]) : mfma_bf16(cur, xb[c], acc[o]);
    }
}

__device__ __forceinline__ void pack4(bf16x8 (&xb)[4], const f32x4 (&x)[kTiles]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) xb[c] = pack_bf16(x[2 * c], x[2 * c + 1]);
}

// one head over the tile's rows (clips of S tokens: block-diagonal key mask), as k_sampler8.hip attention_head8
__device__ __forceinline__ void attention_tile(const f32x4 (&q)[2], const f32x4 (&k)[2], const f32x4 (&v)[2],
                                               const bool (&kvalid)[4], f32x4 (&o)[2]) {
    f32x4 st = mfma_bf16(pack_bf16(k[0], k[1]), pack_bf16(q[0], q[1]), splat4(0.f));   // lane (g, i): S[i][4 g + m]
    float mx = -INFINITY;
#pragma unroll
    for (int m = 0; m < 4; ++m) mx = kvalid[m] ? fmaxf(mx, st[m]) : mx;
    mx = allreduce_g_max(mx);
    f32x4 p;
    float sum = 0.f;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const float e = __builtin_amdgcn_exp2f(1.44269504088896340736f * (st[m] - mx));
        p[m] = kvalid[m] ? e : 0.f;
        sum += p[m];
    }
    sum = allreduce_g_sum(sum);
    const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
    for (int m = 0; m < 4; ++m) p[m] *= inv;
#pragma unroll
    for (int td = 0; td < 2; ++td)
        o[td] = mfma_bf16(pack_bf16(v[td], splat4(0.f)), pack_bf16(p, splat4(0.f)), splat4(0.f));
}

template <int W>
__global__ __launch_bounds__(64 * W) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_sample_wide(SampleArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, r = lane & 15;
    const int S = a.S, R = S * a.G;
    const int cl = r / S, tok = r - cl * S;
    const long tile = (long)blockIdx.x * W + wave;
    const long clip = tile * a.G + cl;
    const bool valid = (r < R) && (clip < (long)a.B);
    const bool is_lat = valid && tok == 0;
    bool kvalid[4];   // keys j = 4 g + m of the SAME clip; padding rows attend to themselves only (finite, never read)
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int j = 4 * g + m;
        kvalid[m] = valid ? (j < R && (j / S) == cl) : (j == r);
    }
    WStream<W> ws;
    ws.base = a.wstream_w + lane;
    ws.src = ws.base;
    ws.end = ws.base + (size_t)a.wide_step_units * 64;
    ws.ring = smem + kOffStage + lane * 16;
    ws.pvbase = smem + kOffPvW + lane * 16;
    ws.wave = wave;
    ws.wbuf = 0;
    ws.rbuf = 1;     // (the prologue's done() flips both: stage 0 lands in buffer 0 and is read from there)
    ws.pv_pending = -1;
    // latent rows (tok == 0): initial noise
    f32x4 lat[kTiles];
#pragma unroll
    for (int t = 0; t < kTiles; ++t) {
        lat[t] = splat4(0.f);
        if (is_lat)
            lat[t] = a.x_init ? ld4(a.x_init + (size_t)clip * kD + 16 * t + 4 * g)
                              : counter_normal4(a.seed, a.clip0 + (uint64_t)clip, 0u, (uint32_t)(4 * t + g), 0u);
    }
    ws.fetch(true, 0);   // block 0's parameters + stage 0
    ws.done();
    constexpr float kScaling = 0.17677669529663687f;  // sqrt(1/32): q * scaling (F.multi_head_attention_forward)
#pragma unroll 1
    for (int step = 0; step < a.T; ++step) {
        // ---- token assembly (denoiser.py:174,180-181); no DMA is in flight here
        f32x4 x[kTiles];
        {
            const float* tt = a.time_tok_clip ? a.time_tok_clip + (size_t)(valid ? clip : 0) * kD : a.time_tok + (size_t)step * kD;
#pragma unroll
            for (int t = 0; t < kTiles; ++t) {
                const int f = 16 * t + 4 * g;
                f32x4 v = splat4(0.f);
                if (valid) {
                    if (tok == 0) v = lat[t] + ld4(a.pe0 + f);
                    else if (tok == 1) v = ld4(tt + f);
                    else v = ld4(a.cond_tok + ((size_t)clip * (S - 2) + (tok - 2)) * kD + f);
                }
                x[t] = v;
            }
        }
        bf16x8 skipst[4][4];   // U-Net skip stack: the packed operands of the skip linear that pops them
#pragma unroll 1
        for (int blk = 0; blk < kLayers; ++blk) {
            const float* pv = reinterpret_cast<const float*>(smem + kOffPvW + (blk % 3) * kWPvBytes);
            bf16x8 xb[4];
            if (blk >= 5) {
                // x = Linear(cat(x, skips.pop())) (cross_attention.py:58-61): one stage, k-pairs 0..3 = x, 4..7 = skip
                ws.fetch(false, 0);
                pack4(xb, x);
                bf16x8 cat[8];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    cat[c] = xb[c];
                    cat[4 + c] = blk == 5 ? skipst[3][c] : blk == 6 ? skipst[2][c] : blk == 7 ? skipst[1][c] : skipst[0][c];
                }
#pragma unroll
                for (int t = 0; t < kTiles; ++t) x[t] = ld4(pv + PW_SKIP_B + 16 * t + 4 * g);
                gemm1<W, kTiles, 8, 0>(x, cat, ws);
                ws.done();
            }
            // ---- self-attention (cross_attention.py:259-266): x = norm1(x + out_proj(attention)); two heads per stage:
            // per head [q, k tiles (16 units)] [v tiles, operand-swapped (8)] [out_proj k-slice (8)]
            pack4(xb, x);
#pragma unroll
            for (int t = 0; t < kTiles; ++t) x[t] += ld4(pv + PV_OUT_B + 16 * t + 4 * g);
#pragma unroll 1
            for (int hp = 0; hp < 2; ++hp) {
                ws.fetch(false, 0);
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int h = 2 * hp + hh;
                    f32x4 qk[4], v[2];
#pragma unroll
                    for (int o = 0; o < 2; ++o) {
                        qk[o] = ld4(pv + PV_IN_B + 16 * (2 * h + o) + 4 * g);
                        qk[2 + o] = ld4(pv + PV_IN_B + kD + 16 * (2 * h + o) + 4 * g);
                        v[o] = splat4(pv[PV_IN_B + 2 * kD + 16 * (2 * h + o) + r]);
                    }
                    if (hh == 0) {
                        gemm1<W, 4, 4, 0>(qk, xb, ws);
                        gemm1<W, 2, 4, 16, true>(v, xb, ws);
                    } else {
                        gemm1<W, 4, 4, 32>(qk, xb, ws);
                        gemm1<W, 2, 4, 48, true>(v, xb, ws);
                    }
                    const f32x4 q[2] = {qk[0] * kScaling, qk[1] * kScaling}, k[2] = {qk[2], qk[3]};
                    f32x4 o[2];
                    attention_tile(q, k, v, kvalid, o);
                    const bf16x8 ob[1] = {pack_bf16(o[0], o[1])};
                    if (hh == 0) gemm1<W, kTiles, 1, 24>(x, ob, ws);
                    else gemm1<W, kTiles, 1, 56>(x, ob, ws);
                }
                ws.done();
            }
            layer_norm_rows<true>(x, pv + PV_LN1_W, pv + PV_LN1_B, g);
            // ---- FFN (cross_attention.py:267-271): x = norm2(x + linear2(gelu(linear1(x)))); four 32-feature chunks per
            // stage: [linear1 (8 units) | linear2 k-pair (8 units)] each
            pack4(xb, x);
#pragma unroll
            for (int t = 0; t < kTiles; ++t) x[t] += ld4(pv + PV_L2_B + 16 * t + 4 * g);
#pragma unroll 1
            for (int cs = 0; cs < 4; ++cs) {
                // the stage fetched now is the next one: the block's next FFN stage, or (last FFN stage) the next block's
                // first stage with its parameter group - at the end of a step, block 0's of the next step
                const bool last = cs == 3;
                ws.fetch(last, (blk == kLayers - 1 ? 0 : blk + 1) % 3);
                f32x4 hid[4][2];
#pragma unroll
                for (int ci = 0; ci < 4; ++ci) {
                    const float* b1 = pv + PV_L1_B + 32 * (4 * cs + ci) + 4 * g;
                    hid[ci][0] = ld4(b1);
                    hid[ci][1] = ld4(b1 + 16);
                }
                // linear1 of the four chunks first (32 MFMAs, independent), then GELU + linear2 chunk by chunk: the GELU of
                // chunk i + 1 has no dependence on the MFMAs of chunk i's linear2
                gemm1<W, 2, 4, 0>(hid[0], xb, ws);
                gemm1<W, 2, 4, 16>(hid[1], xb, ws);
                gemm1<W, 2, 4, 32>(hid[2], xb, ws);
                gemm1<W, 2, 4, 48>(hid[3], xb, ws);
                {
                    const bf16x8 hb0[1] = {pack_bf16(gelu_poly4(hid[0][0]), gelu_poly4(hid[0][1]))};
                    gemm1<W, kTiles, 1, 8>(x, hb0, ws);
                    const bf16x8 hb1[1] = {pack_bf16(gelu_poly4(hid[1][0]), gelu_poly4(hid[1][1]))};
                    gemm1<W, kTiles, 1, 24>(x, hb1, ws);
                    const bf16x8 hb2[1] = {pack_bf16(gelu_poly4(hid[2][0]), gelu_poly4(hid[2][1]))};
                    gemm1<W, kTiles, 1, 40>(x, hb2, ws);
                    const bf16x8 hb3[1] = {pack_bf16(gelu_poly4(hid[3][0]), gelu_poly4(hid[3][1]))};
                    gemm1<W, kTiles, 1, 56>(x, hb3, ws);
                }
                ws.done();
            }
            layer_norm_rows<true>(x, pv + PV_LN2_W, pv + PV_LN2_B, g);
            if (blk < 4) {   // xs.append(x)
                bf16x8 pk[4];
                pack4(pk, x);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (blk == 0) skipst[0][c] = pk[c];
                    else if (blk == 1) skipst[1][c] = pk[c];
                    else if (blk == 2) skipst[2][c] = pk[c];
                    else skipst[3][c] = pk[c];
                }
            }
        }
        // ---- final LayerNorm (SkipTransformerEncoder.norm) + scheduler.step (diffusers 0.17.1 DDIM / DDPM; amuse_hip.h
        // amuse_schedule) on the latent rows.  Block 8's parameter slot (8 % 3 = 2) holds the final LayerNorm; the next
        // step's block 0 went into slot 0.
        {
            const float* pv = reinterpret_cast<const float*>(smem + kOffPvW + ((kLayers - 1) % 3) * kWPvBytes);
            layer_norm_rows<true>(x, pv + PW_FINAL_W, pv + PW_FINAL_B, g);
        }
        if (a.eps_out && is_lat && step == a.T - 1) {
#pragma unroll
            for (int t = 0; t < kTiles; ++t) st4(a.eps_out + (size_t)clip * kD + 16 * t + 4 * g, x[t]);
        }
        if (!a.no_update) {
            const float* cf = a.coef + (size_t)step * 8;
            const float sb = cf[0], sa = cf[1], c0 = cf[2], cx = cf[3], ce = cf[4], sg = cf[5], clipv = cf[6];
            const float inv_sa = 1.0f / sa;
#pragma unroll
            for (int t = 0; t < kTiles; ++t) {
                f32x4 z = splat4(0.f);
                if (sg != 0.f && is_lat)
                    z = a.step_noise ? ld4(a.step_noise + ((size_t)step * a.B + clip) * kD + 16 * t + 4 * g)
                                     : counter_normal4(a.seed, a.clip0 + (uint64_t)clip, (uint32_t)step, (uint32_t)(4 * t + g), 1u);
                {
// each product and sum rounded on its own, like the scheduler's tensor ops (see k_sampler.hip)
#pragma clang fp contract(off)
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const float e = x[t][m], xl = lat[t][m];
                    const float num = __fsub_rn(xl, __fmul_rn(sb, e));
                    float x0 = num * inv_sa;
                    if (clipv > 0.f) x0 = fminf(fmaxf(x0, -clipv), clipv);
                    float nx = __fmul_rn(c0, x0);
                    if (cx != 0.f) nx = __fadd_rn(nx, __fmul_rn(cx, xl));
                    if (ce != 0.f) nx = __fadd_rn(nx, __fmul_rn(ce, e));
                    if (sg != 0.f) nx = __fadd_rn(nx, __fmul_rn(sg, z[m]));
                    lat[t][m] = is_lat ? nx : 0.f;
                }
                }
                if (a.traj_out && is_lat) st4(a.traj_out + ((size_t)step * a.B + clip) * kD + 16 * t + 4 * g, lat[t]);
            }
        }
    }
    if (is_lat && a.latents_out) {
#pragma unroll
        for (int t = 0; t < kTiles; ++t) st4(a.latents_out + (size_t)clip * kD + 16 * t + 4 * g, lat[t]);
    }
}

template <int W>
hipError_t launch_w(const SampleArgs& a, int tiles, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sample_wide<W>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           kSampleWideLdsBytes);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(k_sample_wide<W>, dim3((tiles + W - 1) / W), dim3(64 * W), kSampleWideLdsBytes, stream, a);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_sample_wide(const SampleArgs& a, int waves, hipStream_t stream) {
    const int tiles = (a.B + a.G - 1) / a.G;
    if (waves >= 4) return launch_w<4>(a, tiles, stream);
    return launch_w<2>(a, tiles, stream);   // (one wave per workgroup would need 256 staging registers)
}

}  // namespace amuse
