// One denoising step of the Denoiser's diffusion_only + trans_enc variant as ONE persistent kernel per clip (bf16 / fp16 operands):
// reference models/latent_diffusion/denoiser.py:64-66,177-187 - pose_embd of the 300 x 333 pose sequence x_t, the condition tokens
// [time, con, (emo), (sty)] in front (S = 302..304 rows), query_pos, SkipTransformerEncoder (cross_attention.py:18-64) of nine
// TransformerEncoderLayer.forward_post blocks (:259-272) over ALL S rows - self-attention over ~300 frames in EVERY step, no key mask -,
// encoder.norm, pose_proj of the frame rows, `sample[~mask.T] = 0`; then the scheduler update of x_t (diffusers 0.17.1 DDIM / DDPM as
// restated in amuse_amd/scheduler.py), in place.
//
// The design is k_vae_fused.hip's (amuse_fused.hpp has the shared pieces and the rationale): workgroup = one clip = 8 waves, two per
// SIMD, the clip's 19 row tiles split 3 + 2 over them with the fp32 residual stream in registers for the whole network; the weights
// reach the CU once per step as a stream of 16 KiB stages through a three-deep LDS-DMA ring; K_h / V_h^T of the current head are the
// only cross-wave data (fragment images in LDS); the U-Net skip stack travels through global memory as packed operands.  What differs:
//   * stage 0 is a GEMM: pose_embd (K = 333 -> 11 k-pairs x 8 output tiles = 6 stages), its B operands read straight from x_t
//     (rows are 1332 B apart: scalar loads, issued a stage ahead), the condition-token rows swapped in afterwards;
//   * encoder blocks: no cross-attention, two norms (decoder_block<..., ENCL = true>);
//   * the last stage is pose_proj + the scheduler update instead of final_layer + the rotation epilogue: a wave stages 96 features of
//     its tile in LDS, then walks the elements: eps_hat (optionally stored), x_{t-1} = f(x_t, eps_hat, z) written over x_t - every
//     element is read and written by the wave that owns its row tile, so the update is race-free in place.
// One launch per step: the 400 KB state stays in L2 / Infinity Cache between launches, the step itself is ~0.6 ms.
#include "amuse_fused.hpp"

#define OP_KERNEL OP_SUFFIX(k_den_fused)
#define OP_LAUNCH OP_SUFFIX(launch_den_fused)

namespace amuse {
namespace {

// (333 input features in 22 k-tiles = 11 k-pairs; zero weights / operands beyond 333)
constexpr int kEmbStages = 6;                    // 88 units + 8 of padding

// the B operands of k-pairs c0, c0 + 1 for the wave's tiles: lane (g, r) holds features 32 c + 4 g + m | 32 c + 16 + 4 g + m of row r.
// x_t rows are 1332 B apart, so the four-float groups are only 4-byte aligned: unaligned dwordx4 loads (global memory takes them).
// k-pair 10 ends at feature 351 > 332: its groups are read element-wise under a bound (the last row of the array must not be overrun).
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
template <int NT>
__device__ __forceinline__ void load_xt(f32x4 (&v)[NT][4], const float* xb, int tile0, int r, int g, int npre, int c0) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int fi = 16 * (tile0 + 4 * j) + r - npre;
        const bool fv = fi >= 0 && fi < kFrames;
        const float* row = xb + (size_t)(fv ? fi : 0) * kFeats + 4 * g;
#pragma unroll
        for (int h = 0; h < 4; ++h) {   // h = 2 cc + half: features 32 (c0 + cc) + 16 half + 4 g + m
            const int f = 32 * c0 + 16 * h;
            f32x4 t;
            if (f + 15 < kFeats) {   // the group lies inside the row for every g (f + 4 g + 3 <= 332)
                t = *reinterpret_cast<const f32x4u*>(row + f);
            } else {
#pragma unroll
                for (int m = 0; m < 4; ++m) t[m] = (f + 4 * g + m < kFeats) ? row[f + m] : 0.f;
            }
            v[j][h] = fv ? t : splat4(0.f);
        }
    }
}

template <int NT, bool NOATTN>
__device__ __forceinline__ void den_tiles(const DenFusedArgs& a, char* smem, Stager& sg, int tile0, int b, int wave, int lane) {
    const int g = lane >> 4, r = lane & 15;
    char* kv = smem + kOffKv;
    const float* pvl = reinterpret_cast<const float*>(smem + kOffPv);   // [2][kPvSlot / 4]
    const unsigned lds0 = lds_addr(smem);
    uint4* skipbuf = a.skip + (size_t)b * (4 * 20 * 4 * 64);
    const int npre = a.npre, S = kFrames + npre;
    float* xst = a.x + (size_t)b * kFrames * kFeats;
    // ---------------- pose_embd (denoiser.py:178): six stages of two k-pairs x eight output tiles
    f32x4 x[NT][kTiles];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int t = 0; t < kTiles; ++t) x[j][t] = ld4(a.emb_bias + 16 * t + 4 * g);
    f32x4 cur[NT][4];
    load_xt<NT>(cur, xst, tile0, r, g, npre, 0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // block 0's parameters and weight stages 0, 1 are in
#pragma unroll 1
    for (int s6 = 0; s6 < kEmbStages; ++s6) {
        stage_fetch(sg);
        OPV xb2[NT][2];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) xb2[j][cc] = OP_PACK(cur[j][2 * cc], cur[j][2 * cc + 1]);
        if (s6 + 1 < kEmbStages) load_xt<NT>(cur, xst, tile0, r, g, npre, 2 * (s6 + 1));   // the next stage's operands, a stage ahead
        for_units<kStage, 0>(sg, [&](int u, OPV wf) {
            const int cc = u >> 3, o = u & 7;
#pragma unroll
            for (int j = 0; j < NT; ++j) x[j][o] = OP_MFMA(wf, xb2[j][cc], x[j][o]);
        });
        stage_end(sg);
    }
    // xseq = cat(emb_latent, pose_embd(sample)) + query_pos (denoiser.py:180-181); the tokens arrive with their positions added
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int frame = 16 * (tile0 + 4 * j) + r;
        const float* pt = frame == 0 ? a.ttok + (size_t)b * a.ttok_stride : a.ctok + ((size_t)b * (npre - 1) + (frame < npre ? frame - 1 : 0)) * kD;
#pragma unroll
        for (int t = 0; t < kTiles; ++t) {
            const int c = 16 * t + 4 * g;
            x[j][t] = frame >= S ? splat4(0.f) : (frame < npre ? ld4(pt + c) : x[j][t] + ld4(a.pe + (size_t)frame * kD + c));
        }
    }
    const float* nocal = pvl;   // (no cross-attention constants in encoder blocks; never read)
#pragma unroll 1
    for (int blk = 0; blk < 4; ++blk)
        decoder_block<NT, 0, false, true, NOATTN>(x, sg, a.pvec, nullptr, blk, tile0, pvl + (blk & 1) * (kPvSlot / 4), a.pvec + (blk + 1) * PV_BLOCK,
                                          lds0 + kOffPv + ((blk + 1) & 1) * kPvSlot, nocal, kv, skipbuf, S, wave, lane, S);
    decoder_block<NT, 1, false, true, NOATTN>(x, sg, a.pvec, nullptr, 4, tile0, pvl, a.pvec + 5 * PV_BLOCK, lds0 + kOffPv + kPvSlot, nocal, kv, skipbuf, S,
                                      wave, lane, S);
#pragma unroll 1
    for (int blk = 5; blk < kLayers; ++blk)
        decoder_block<NT, 2, false, true, NOATTN>(x, sg, a.pvec, nullptr, blk, tile0, pvl + (blk & 1) * (kPvSlot / 4),
                                          blk + 1 < kLayers ? a.pvec + (blk + 1) * PV_BLOCK : nullptr,
                                          lds0 + kOffPv + ((blk + 1) & 1) * kPvSlot, nocal, kv, skipbuf, S, wave, lane, S);
    // ---------------- encoder.norm -> pose_proj (333 outputs in 24 tiles) -> mask -> eps_hat / scheduler update.  As in the decode kernel
    // (k_vae_fused.hip) the last stage leaves the stage protocol: the whole pose_proj image goes to LDS once, and no wave waits on vmcnt again but for
    // the x_t (and noise) values of the scheduler update - loaded a quarter tile at a time IN FRONT of that quarter's stores (a load behind a store sits
    // out the store's round trip: stores count in vmcnt on gfx950).
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    {
        const uint4* fsrc = sg.src - (size_t)(2 * kStage + 2 * wave) * 64 + (size_t)wave * 12 * 64;
#pragma unroll
        for (int i = 0; i < 12; ++i) glds16(fsrc + i * 64, lds0 + kOffFinalW + (wave * 12 + i) * 1024);
        float* lpar = reinterpret_cast<float*>(smem + kOffFinalPar);
        const int t = wave * 64 + lane;
        if (t < 96) st4(lpar + 4 * t, ld4(a.final_bias + 4 * t));
        else if (t < 160) st4(lpar + 4 * t, ld4(a.pvec + PV_FINAL_W + 4 * (t - 96)));
        static_assert(PV_FINAL_B == PV_FINAL_W + kD, "the last norm's parameters are read as one run");
    }
    const int len = a.lengths ? a.lengths[b] : kFrames;
    const float* cf = a.coef;
    const float sb = cf ? cf[0] : 0.f, sa = cf ? cf[1] : 1.f, c0 = cf ? cf[2] : 0.f, cx = cf ? cf[3] : 0.f, ce = cf ? cf[4] : 0.f,
                sgm = cf ? cf[5] : 0.f, clipv = cf ? cf[6] : 0.f;
    const float inv_sa = 1.0f / sa;
    const float* nz = a.step_noise ? a.step_noise + (size_t)b * kFrames * kFeats : nullptr;
    float* eo = a.eps_out ? a.eps_out + (size_t)b * kFrames * kFeats : nullptr;
    asm volatile("" ::"v"(len), "v"(sb), "v"(sa), "v"(c0), "v"(cx), "v"(ce), "v"(sgm), "v"(clipv));   // (consumed here: nothing stays pending across the stores)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const float* lbias = reinterpret_cast<const float*>(smem + kOffFinalPar);
    const float* lnorm = lbias + 384;
    const char* wimg = smem + kOffFinalW + lane * 16;
    float* fst = reinterpret_cast<float*>(smem + kOffFinalStage) + wave * 16 * kQStride;
#pragma unroll 1
    for (int j = 0; j < NT; ++j) {
        const int tile = tile0 + 4 * j;
        const int rows_here = min(16, S - 16 * tile);              // <= 0 for a tile beyond the sequence
        if (rows_here <= 0) {
            rotate_tiles<NT>(x);
            continue;
        }
        layer_norm_rows<true>(x[0], lnorm, lnorm + kD, g);
        OPV xb1[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) xb1[c] = OP_PACK(x[0][2 * c], x[0][2 * c + 1]);
        rotate_tiles<NT>(x);
        const int fr_lane = 16 * tile + r - npre;                  // this lane's frame (row r of the tile)
        const bool keep = fr_lane >= 0 && fr_lane < len;           // sample[~mask.T] = 0 (denoiser.py:187)
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {   // 12 output tiles x 4 k-pairs = 48 units (k-pair outer, output tile inner)
            f32x4 f[12];
#pragma unroll
            for (int o = 0; o < 12; ++o) f[o] = ld4(lbias + 16 * (12 * half + o) + 4 * g);
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int o = 0; o < 12; ++o)
                    f[o] = OP_MFMA(__builtin_bit_cast(OPV, *reinterpret_cast<const uint4*>(wimg + ((half * 4 + c) * 12 + o) * 1024)), xb1[c], f[o]);
#pragma unroll 1
            for (int qq = 0; qq < 2; ++qq) {
                const int quarter = 2 * half + qq;
                const int f0 = 96 * quarter, nfe = quarter == 3 ? kFeats - 288 : 96;
                const int n = rows_here * nfe;
#pragma unroll
                for (int o = 0; o < 6; ++o) st4(fst + r * kQStride + 16 * o + 4 * g, keep ? (qq == 0 ? f[o] : f[6 + o]) : splat4(0.f));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the staging tile is wave-private: no barrier
                // x_t of a batch of kPer elements per lane first (all in flight together), then their updates and stores
                constexpr int kPer = 8;
#pragma unroll 1
                for (int k0 = 0; k0 < 16 * 96 / 64; k0 += kPer) {
                    float xv[kPer];
                    if (cf) {
#pragma unroll
                        for (int k = 0; k < kPer; ++k) {
                            const int i = lane + 64 * (k0 + k), rr = i / nfe, c = i - rr * nfe, fr = 16 * tile + rr - npre;
                            const bool ok = i < n && fr >= 0;
                            xv[k] = ok ? xst[(size_t)fr * kFeats + f0 + c] : 0.f;
                        }
                    }
#pragma unroll
                    for (int k = 0; k < kPer; ++k) {
                        const int i = lane + 64 * (k0 + k), rr = i / nfe, c = i - rr * nfe, fr = 16 * tile + rr - npre;
                        if (i >= n || fr < 0) continue;              // (fr < 0: a condition-token row - its output is dropped, denoiser.py:184)
                        const float e = fst[rr * kQStride + c];
                        const size_t el = (size_t)fr * kFeats + f0 + c;
                        if (eo) eo[el] = e;
                        if (cf) {   // scheduler.step, the 16-bit modes' form of k_sampler.hip's update (reciprocal multiply)
#pragma clang fp contract(off)
                            const float xl = xv[k];
                            float x0 = (xl - sb * e) * inv_sa;
                            if (clipv > 0.f) x0 = fminf(fmaxf(x0, -clipv), clipv);
                            float nx = c0 * x0;
                            if (cx != 0.f) nx = nx + cx * xl;
                            if (ce != 0.f) nx = nx + ce * e;
                            if (sgm != 0.f) {   // (explicit noise - tests - is read in place: that path pays a round trip per element)
                                const float z = nz ? nz[el] : counter_normal4(a.seed, a.clip0 + (uint64_t)b, (uint32_t)a.step, (uint32_t)(el >> 2), 1u)[el & 3];
                                nx = nx + sgm * z;
                            }
                            xst[el] = nx;
                        }
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }
    }
}

template <bool NOATTN>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void OP_KERNEL(DenFusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x;
    const unsigned lds0 = lds_addr(smem);
    glds16(reinterpret_cast<const uint4*>(a.pvec) + wave * 64 + lane, lds0 + kOffPv + wave * 1024);   // block 0's parameters
    Stager sg;
    sg.src = a.wstream + (size_t)wave * 2 * 64 + lane;
    sg.dst0 = lds0 + kOffW + wave * 2048;
    sg.ring = smem + kOffW + lane * 16;
    sg.widx = 0;
    sg.ridx = 0;
    stage_fetch(sg);
    stage_fetch(sg);
    if (wave < 4) den_tiles<3, NOATTN>(a, smem, sg, wave, b, wave, lane);
    else den_tiles<2, NOATTN>(a, smem, sg, wave + 8, b, wave, lane);
}

}  // namespace

hipError_t OP_LAUNCH(const DenFusedArgs& a, hipStream_t stream) {
    static DeviceOnce once;
    int dev_;
    if (!once.done(&dev_)) {
        for (const void* k : {reinterpret_cast<const void*>(&OP_KERNEL<false>), reinterpret_cast<const void*>(&OP_KERNEL<true>)}) {
            hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, kFusedFinalLdsBytes);
            if (e != hipSuccess) return e;
        }
        once.set(dev_);
    }
    if (a.ablate_attention) hipLaunchKernelGGL(OP_KERNEL<true>, dim3(a.B), dim3(512), kFusedFinalLdsBytes, stream, a);   // timing ablation (bench.py)
    else hipLaunchKernelGGL(OP_KERNEL<false>, dim3(a.B), dim3(512), kFusedFinalLdsBytes, stream, a);
    return hipGetLastError();
}

}  // namespace amuse
