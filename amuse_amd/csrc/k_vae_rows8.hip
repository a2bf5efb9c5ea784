// The row stages of the staged decode (MotionPrior.decode, reference models/latent_diffusion/vae.py:216-278; decoder layer
// cross_attention.py:323-345; final_layer + 6D -> axis-angle infer_ldm.py:168-173) in the fp32x arithmetic (AMUSE_PREC_F32X: split-fp16
// operands, three v_mfma_f32_16x16x32_f16 per product, fp32 everything else) WITHOUT split-K: the shape of the fused decoder's row half.
//
// k_vae_rows (k_vae.hip) gives a workgroup of four waves ONE 16-row tile: split-K over the waves, two combines per block through LDS, and
// the stage's weights - 786 KB in this mode - streamed from L2 once per TILE: 42 % of the load-path floor, 230 us per launch at 256 clips.
// Here a workgroup is eight to twelve waves (chosen per launch, launch_vae_rows8x below) with a tile EACH:
//   * the stage's weights reach the CU once per workgroup: one stream in consumption order, cut into 16 KiB LDS stages (8 unit pairs:
//     hi | lo fp16 fragments of one 16-feature x 32-k block), copied global -> LDS by LDS-DMA three stages deep (the protocol of
//     k_vae_fused.hip: two 1 KiB pieces per wave and stage, vmcnt(2) + s_barrier at a stage's end);
//   * every GEMM is full-K inside the wave, straight into the residual registers - no partial sums, no combine, LayerNorm inside the
//     wave; the operand rows are split into hi / lo once per GEMM group;
//   * one stage = [out_proj + norm1 + cross-attention constant + norm2 + FFN + norm3 (+ skip push / skip linear)] of block i and the
//     q, k, v projection of block i + 1 (or decoder.norm + final_layer + the rotation epilogue), exactly the cut of k_vae_rows, so the
//     attention kernels between the stages stay as they are.
// Every LDS stage is either one k-pair x 8 output tiles or (linear1) 4 k-pairs x 2 output tiles; the last stage's 24 output tiles go in
// four quarters of 6 (one staging tile of 16 x 96 features per wave, as in k_vae_fused.hip).
#include "amuse_dev.hpp"
#include "amuse_kernels.hpp"

namespace amuse {
namespace {

// erf of the FFN activation: libm erff (as k_vae_rows<f16x2>; the bits of the round-3 kernel).  Abramowitz-Stegun 7.1.26 on the hardware rcp / exp2 and the branch-free
// fit of the fp32x sampler time 1.96 / 2.05 against 1.97 ms per 256-clip decode (profiles/r04_rows8x_erf_ab.txt).
// kProd: the weight stream is copied by ONE extra wave that does nothing else (the row waves never wait on vmcnt after their input loads), and the
// biases read behind the first activation store come from LDS: on this ISA stores count in vmcnt like loads, so a row wave that also copies
// (vmcnt(2) at every stage end) or loads a bias behind its q stores sits out every store's round trip to L2 / HBM - 0.43 of the decode's 2.35 ms
// at 256 clips (profiles/r04_rows8x_store_ablation.txt; false = eight of the row waves copy, the round-3 kernel).
constexpr bool kProd = true;
constexpr int kRowTiles = 19;                 // ceil(300 / 16)
constexpr int kTilesPerWave = 1;              // row tiles per wave: a weight fragment read from LDS feeds 3 x NT MFMAs
// the waves that copy the stream (16 units per LDS stage): eight with two pieces each, four with four in the small workgroups
constexpr int dma_waves(int waves) { return waves >= 8 ? 8 : 4; }
constexpr int kStage = 16;                    // units per LDS stage (8 hi | lo pairs)
constexpr int kStageBytes = kStage * 1024;
// LDS stages in the weight ring (the copying wave keeps kWBufs - 1 stages ahead of the stage being read, kWBufs - 2 of them may still be in flight at a stage's end)
constexpr int kWBufs = 3;                     // (deeper rings measured flat: profiles/r04_rows8x_ring_depth_ab.txt)
static_assert(kWBufs == 3 || (kProd && kWBufs <= 5), "deeper rings: copying-wave protocol only; vmcnt is a 6-bit counter");
constexpr int kQStride = 100;                 // staging row stride (floats) of one 96-feature quarter of the last stage
constexpr int kOffW = 0;
constexpr int kOffBias = kWBufs * kStageBytes;          // [384] floats: in_proj bias of block `stage` / final_layer.bias (last stage)
constexpr int kOffStage = kOffBias + 384 * 4;
constexpr int rows8_lds_bytes(int waves, bool last_stage) { return kOffStage + (last_stage ? waves * 16 * kQStride * 4 : 0); }   // (the staging tiles serve the final stage only)

__device__ __forceinline__ void glds16(const uint4* gsrc, unsigned lds_dst) {   // (k_vae_fused.hip: LDS-DMA outside hipcc's waitcnt bookkeeping)
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}
template <int P>   // P = pieces (1 KiB units) a copying wave moves per stage
struct Stager {
    const uint4* src;   // this lane's source address of the wave's pieces of the NEXT stage to fetch
    unsigned dst0;      // LDS byte address of the wave's pieces inside buffer 0
    const char* ring;   // weight ring base (generic pointer) + lane * 16
    int widx, ridx;     // buffer the next fetch fills / buffer the current stage reads
    bool dma;           // this wave copies (waves 0..7)
};
template <int P>
__device__ __forceinline__ void stage_fetch(Stager<P>& s) {
    if constexpr (P == 0) return;
    if (s.dma) {   // (wave-uniform; the other waves' vmcnt(P) at the stage's end is trivially true - the barrier is what they need)
        const unsigned d = __builtin_amdgcn_readfirstlane(s.dst0 + s.widx * kStageBytes);
#pragma unroll
        for (int i = 0; i < P; ++i) glds16(s.src + i * 64, d + i * 1024);
    }
    s.src += kStage * 64;
    s.widx = s.widx == kWBufs - 1 ? 0 : s.widx + 1;
}
template <int P>
__device__ __forceinline__ void stage_end(Stager<P>& s) {
    if constexpr (P == 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // (row wave beside a copying wave: no vmcnt)
    else if constexpr (P == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else if constexpr (P == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
    s.ridx = s.ridx == kWBufs - 1 ? 0 : s.ridx + 1;
}
template <int P>
__device__ __forceinline__ f16x8 wfrag(const Stager<P>& s, int u) {
    return __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(s.ring + s.ridx * kStageBytes + u * 1024));
}
// one product of split operands: acc += Wl.xh + Wh.xl + Wh.xh (the term order of gemm_ring_s, amuse_dev.hpp)
__device__ __forceinline__ f32x4 mfma3(f16x8 wh, f16x8 wl, const F16Pair& x, f32x4 acc) {
    acc = mfma_f16(wl, x.hi, acc);
    acc = mfma_f16(wh, x.lo, acc);
    return mfma_f16(wh, x.hi, acc);
}
// the 8 unit pairs of the current LDS stage, pair i -> f(i, hi, lo); the fragments of pair i + 1 are read before pair i's MFMAs
template <int P, class F>
__device__ __forceinline__ void for_pairs(Stager<P>& s, F&& f) {
    stage_fetch(s);
    f16x8 h = wfrag(s, 0), l = wfrag(s, 1);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const f16x8 ch = h, cl = l;
        if (i + 1 < 8) { h = wfrag(s, 2 * i + 2); l = wfrag(s, 2 * i + 3); }
        f(i, ch, cl);
    }
    stage_end(s);
}
// acc[j][o] += W[o-tile][k-pairs 0..3] . x_j over FOUR LDS stages (k-pair outer, 8 output tiles inner); a fragment pair read from LDS
// feeds the MFMAs of all NT row tiles of the wave
template <int NT, int P>
__device__ __forceinline__ void gemm_k128_o8(f32x4 (&acc)[NT][8], const F16Pair (&xs)[NT][4], Stager<P>& s) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
        for_pairs(s, [&](int o, f16x8 wh, f16x8 wl) {
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[j][o] = mfma3(wh, wl, xs[j][c], acc[j][o]);
        });
}
template <int NT>
__device__ __forceinline__ void split_x(F16Pair (&xs)[NT][4], const f32x4 (&x)[NT][kTiles]) {
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) xs[j][c] = split_f16(x[j][2 * c], x[j][2 * c + 1]);
}

// kWaves waves per workgroup (12 = three per SIMD: the kernel needs ~158 registers; 8 for launches of few tiles - the same bits: a tile's
// arithmetic does not depend on its workgroup), NT row tiles per wave: wave w of workgroup wg owns tiles (wg * kWaves + w) * NT + j
// (workgroups of up to six waves go two to a CU: three waves per SIMD either way)
constexpr int waves_per_eu(int waves) { return waves <= 6 ? 3 : (waves + 3) / 4; }
constexpr int kProdWaves = kProd ? 1 : 0;   // the copying wave sits behind the kWaves row waves
// ENC: the stages of MotionPrior.encode behind its embedding stage (vae.py:154-214; encoder layers cross_attention.py:259-272: no cross-attention, two
// norms): rows are [2 distribution tokens | 300 frames], S = 302; stage 9 ends with encoder.norm and writes the two distribution rows (a.tiles = 1: only
// tile 0 of a clip is launched).  Stage 0 - skel_embedding over K = 333 - stays with k_vae_rows<f16x2, M_ENC>, same arrays.  The Denoiser's diffusion_only
// step (denoiser.py:177-187: the same encoder layers over S = a.S rows) runs its stages 1..8 here too; pose_embd and pose_proj + update stay with k_vae_rows.
template <int NT, int kWaves, bool ENC>
__global__ __launch_bounds__(64 * (kWaves + kProdWaves)) __attribute__((amdgpu_waves_per_eu(waves_per_eu(kWaves + kProdWaves), waves_per_eu(kWaves + kProdWaves)))) void k_vae_rows8x(VaeRowsArgs a) {
    constexpr int kDmaWaves = dma_waves(kWaves), kPieces = kProd ? 0 : kStage / kDmaWaves;
    // rows per clip: 300 frames; + the two distribution tokens (encode); a.S = condition tokens + 300 (302..304) for the Denoiser's diffusion_only stages
    const int S = ENC ? (a.S > 0 ? a.S : kFrames + 2) : kFrames;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, r = lane & 15;
    int b[NT], rt[NT], frame[NT];
    bool tvalid[NT], rvalid[NT];
    size_t row[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int gt = (blockIdx.x * kWaves + wave) * NT + j;   // this wave's j-th (clip, row tile)
        tvalid[j] = gt < a.B * a.tiles;                         // (tiles beyond the batch only keep the stage protocol turning)
        b[j] = !tvalid[j] ? 0 : a.tiles == 1 ? gt : gt / kRowTiles;   // (a.tiles: 19, or 1 in encode's last stage)
        rt[j] = tvalid[j] ? gt - b[j] * a.tiles : 0;
        frame[j] = rt[j] * 16 + r;
        rvalid[j] = tvalid[j] && frame[j] < S;
        row[j] = (size_t)b[j] * S + (rvalid[j] ? frame[j] : 0);
    }
    const size_t nrows = (size_t)a.B * S;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)smem;
    Stager<kPieces> sg;
    sg.dma = wave < kDmaWaves;
    const bool hoisted = a.stage == 1 && a.c1 != nullptr;   // block 0's self-attention half comes as the constant c1 (VaeRowsArgs)
    sg.src = a.wstream + (size_t)a.stage_base[a.stage] * 64 + (hoisted ? 4 * kStage * 64 : 0) + (size_t)(wave % kDmaWaves) * kPieces * 64 + lane;
    sg.dst0 = lds0 + kOffW + (wave % kDmaWaves) * kPieces * 1024;
    sg.ring = smem + kOffW + lane * 16;
    sg.widx = 0;
    sg.ridx = 0;
    if constexpr (kProd) {
        if (wave == kWaves) {
            // the copying wave: stage n + 2 goes out when stage n - 1's barrier has passed, stage n's barrier waits for stage n + 1 - the
            // protocol of the row waves' own copies, 16 pieces per stage in one wave; LDS stages of this launch:
            const int blk = a.stage - 1;
            const bool hoisted = a.stage == 1 && a.c1 != nullptr, c1_only = a.stage == 1 && a.c1_out != nullptr;
            const int nst = c1_only ? 4 : (ENC && a.stage == kLayers ? 0 : 12) + (a.stage >= 1 ? 36 + (blk >= 4 && blk <= 7 ? 8 : 0) : 0) - (hoisted ? 4 : 0);
            const uint4* src = a.wstream + (size_t)a.stage_base[a.stage] * 64 + (hoisted ? 4 * kStage * 64 : 0) + lane;   // (hoisted: no out_proj)
            int wb = 0, left = nst;   // (the kWBufs - 1 fetches past the launch's last stage repeat it: they land in free buffers, nobody reads them)
            auto fetch = [&]() {
                const unsigned d = __builtin_amdgcn_readfirstlane(lds0 + kOffW + wb * kStageBytes);
#pragma unroll
                for (int i = 0; i < kStage; ++i) glds16(src + i * 64, d + i * 1024);
                if (kWBufs == 3 || --left > 0) src += kStage * 64;
                wb = wb == kWBufs - 1 ? 0 : wb + 1;
            };
#pragma unroll
            for (int i = 0; i < kWBufs - 1; ++i) fetch();
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(16 * (kWBufs - 2)) : "memory");
#pragma unroll 1
            for (int n = 0; n < nst; ++n) {
                fetch();
                asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(16 * (kWBufs - 2)) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the surplus fetches must not outlive the workgroup's LDS
            return;
        }
    }
    stage_fetch(sg);
    stage_fetch(sg);
    // biases read behind the first activation store, from LDS (see kProd): in_proj of block `stage` / final_layer
    float* lbias = reinterpret_cast<float*>(smem + kOffBias);
    if (threadIdx.x < 96 && !(ENC && a.stage == kLayers))
        st4(lbias + 4 * threadIdx.x, ld4((a.stage < kLayers ? a.pvec + a.stage * PV_BLOCK + PV_IN_B : a.final_bias) + 4 * threadIdx.x));
    int len[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) len[j] = a.lengths ? a.lengths[b[j]] : S;
    f32x4 x[NT][kTiles];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        // stage 0: queries = zeros + query_pos_decoder.pe[:300]  (vae.py:220,258)
        const float* src = a.stage == 0 ? a.pe + (size_t)frame[j] * kD : hoisted ? a.c1 + (size_t)(rvalid[j] ? frame[j] : 0) * kD : a.x + row[j] * kD;
#pragma unroll
        for (int t = 0; t < kTiles; ++t) x[j][t] = rvalid[j] ? ld4(src + 16 * t + 4 * g) : splat4(0.f);
    }
    // The loads above are consumed HERE on every path: stage 0 does not touch x before its first store, and a load still pending at the
    // merge in front of the in_proj loop makes hipcc wait there with vmcnt(0) - behind the x / skip stores of every other stage.
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int t = 0; t < kTiles; ++t) asm volatile("" ::"v"(x[j][t]));
#pragma unroll
    for (int j = 0; j < NT; ++j) asm volatile("" ::"v"(len[j]));
    // (the protocol's first wait: both prefetched stages but the second one's pieces)
    if constexpr (kPieces == 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else if constexpr (kPieces == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    F16Pair xs[NT][4];
    if (a.stage >= 1) {
        const int blk = a.stage - 1;
        const float* pv = a.pvec + blk * PV_BLOCK;
        // ---- self-attention out_proj + residual + norm1  (cross_attention.py:323-330)
        if (!hoisted) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            f32x4 o[kTiles];
#pragma unroll
            for (int t = 0; t < kTiles; ++t) o[t] = rvalid[j] ? ld4(a.attn_o + row[j] * kD + 16 * t + 4 * g) : splat4(0.f);
#pragma unroll
            for (int c = 0; c < 4; ++c) xs[j][c] = split_f16(o[2 * c], o[2 * c + 1]);
#pragma unroll
            for (int t = 0; t < kTiles; ++t) x[j][t] += ld4(pv + PV_OUT_B + 16 * t + 4 * g);
        }
        gemm_k128_o8<NT>(x, xs, sg);
#pragma unroll
        for (int j = 0; j < NT; ++j) layer_norm_rows<false>(x[j], pv + PV_LN1_W, pv + PV_LN1_B, g);
        if (a.stage == 1 && a.c1_out) {   // the hoisted constant's own computation (one clip): write it and stop
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if (!rvalid[j]) continue;
#pragma unroll
                for (int t = 0; t < kTiles; ++t) st4(a.c1_out + (size_t)frame[j] * kD + 16 * t + 4 * g, x[j][t]);
            }
            if constexpr (!kProd) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            return;
        }
        }
        if constexpr (!ENC) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            // cross-attention onto the single latent token == per-clip constant; norm2  (cross_attention.py:331-337)
            const float* ca = a.ca + ((size_t)b[j] * kLayers + blk) * kD;
#pragma unroll
            for (int t = 0; t < kTiles; ++t) x[j][t] += ld4(ca + 16 * t + 4 * g);
            layer_norm_rows<false>(x[j], pv + PV_LN2_W, pv + PV_LN2_B, g);
        }
        }
        // ---- FFN in 16 chunks of 32 hidden features: linear1 (4 k-pairs x 2 tiles: one LDS stage) -> erf-GELU -> linear2's k-pair of
        // those features (8 output tiles: one LDS stage), accumulated into the residual; norm3  (cross_attention.py:338-340)
        split_x<NT>(xs, x);
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int t = 0; t < kTiles; ++t) x[j][t] += ld4(pv + PV_L2_B + 16 * t + 4 * g);
#pragma unroll 1
        for (int ch = 0; ch < 16; ++ch) {
            f32x4 hid[NT][2];
            const f32x4 b0 = ld4(pv + PV_L1_B + 32 * ch + 4 * g), b1 = ld4(pv + PV_L1_B + 32 * ch + 16 + 4 * g);
#pragma unroll
            for (int j = 0; j < NT; ++j) { hid[j][0] = b0; hid[j][1] = b1; }
            for_pairs(sg, [&](int i, f16x8 wh, f16x8 wl) {
#pragma unroll
                for (int j = 0; j < NT; ++j) hid[j][i & 1] = mfma3(wh, wl, xs[j][i >> 1], hid[j][i & 1]);
            });
            F16Pair hs[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int m = 0; m < 4; ++m)
                        hid[j][i][m] = gelu_erf(hid[j][i][m]);
                hs[j] = split_f16(hid[j][0], hid[j][1]);
            }
            for_pairs(sg, [&](int o, f16x8 wh, f16x8 wl) {
#pragma unroll
                for (int j = 0; j < NT; ++j) x[j][o] = mfma3(wh, wl, hs[j], x[j][o]);
            });
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) layer_norm_rows<false>(x[j], pv + (ENC ? PV_LN2_W : PV_LN3_W), pv + (ENC ? PV_LN2_B : PV_LN3_B), g);   // (an encoder layer's norm2)
        // ---- U-Net wiring (cross_attention.py:104-121): input blocks push, the skip linear runs ahead of the next output block
        if (blk < 4) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if (!rvalid[j]) continue;
                float* sk = a.skip + ((size_t)blk * nrows + row[j]) * kD;
#pragma unroll
                for (int t = 0; t < kTiles; ++t) st4(sk + 16 * t + 4 * g, x[j][t]);
            }
        }
        if (blk >= 4 && blk <= 7) {   // x = linear_blocks[blk - 4](cat(x, xs.pop()))
            const float* bs = a.pvec + PV_SKIP_B + (blk - 4) * kD;
            split_x<NT>(xs, x);
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int t = 0; t < kTiles; ++t) x[j][t] = ld4(bs + 16 * t + 4 * g);
            gemm_k128_o8<NT>(x, xs, sg);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const float* sk = a.skip + ((size_t)(7 - blk) * nrows + row[j]) * kD;
                f32x4 sv[kTiles];
#pragma unroll
                for (int t = 0; t < kTiles; ++t) sv[t] = rvalid[j] ? ld4(sk + 16 * t + 4 * g) : splat4(0.f);
#pragma unroll
                for (int c = 0; c < 4; ++c) xs[j][c] = split_f16(sv[2 * c], sv[2 * c + 1]);
            }
            gemm_k128_o8<NT>(x, xs, sg);
        }
    }
    if (a.stage < kLayers) {
        // ---- residual stream for the next stage + in_proj of block `stage`: q | k | v as three groups of 8 output tiles
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            if (!rvalid[j]) continue;
#pragma unroll
            for (int t = 0; t < kTiles; ++t) st4(a.x + row[j] * kD + 16 * t + 4 * g, x[j][t]);
        }
        split_x<NT>(xs, x);
        const float scaling = 0.17677669529663687f;   // 1 / sqrt(32): q * scaling (F.multi_head_attention_forward)
#pragma unroll 1
        for (int grp = 0; grp < 3; ++grp) {
            f32x4 acc[NT][8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const f32x4 bi = ld4(lbias + grp * kD + 16 * t + 4 * g);
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[j][t] = bi;
            }
            gemm_k128_o8<NT>(acc, xs, sg);
            float* dst = grp == 0 ? a.q : grp == 1 ? a.k : a.v;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if (!rvalid[j]) continue;
#pragma unroll
                for (int t = 0; t < 8; ++t) {   // output tile t = head t / 2, features 16 (t & 1) ..
                    const size_t hrow = (((size_t)b[j] * kHeads + (t >> 1)) * S + frame[j]) * 32;
                    st4(dst + hrow + 16 * (t & 1) + 4 * g, grp == 0 ? acc[j][t] * scaling : acc[j][t]);
                }
            }
        }
    } else if constexpr (ENC) {
        // ---- encoder.norm of the distribution rows (mu | logvar, vae.py:196-203): rows 0, 1 of a clip's tile 0
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            layer_norm_rows<false>(x[j], a.pvec + PV_FINAL_W, a.pvec + PV_FINAL_B, g);
            if (tvalid[j] && rt[j] == 0 && r < 2) {
#pragma unroll
                for (int t = 0; t < kTiles; ++t) st4(a.stats_out + ((size_t)b[j] * 2 + r) * kD + 16 * t + 4 * g, x[j][t]);
            }
        }
    } else {
        // ---- decoder.norm -> final_layer (333 outputs in 24 tiles, four quarters of 6) -> rotation epilogue
#pragma unroll
        for (int j = 0; j < NT; ++j) layer_norm_rows<false>(x[j], a.pvec + PV_FINAL_W, a.pvec + PV_FINAL_B, g);
        split_x<NT>(xs, x);
        float* fst = reinterpret_cast<float*>(smem + kOffStage) + wave * 16 * kQStride;
#pragma unroll 1
        for (int quarter = 0; quarter < 4; ++quarter) {
            f32x4 f[NT][6];
#pragma unroll
            for (int o = 0; o < 6; ++o) {
                const f32x4 bi = ld4(lbias + 16 * (6 * quarter + o) + 4 * g);
#pragma unroll
                for (int j = 0; j < NT; ++j) f[j][o] = bi;
            }
#pragma unroll
            for (int s3 = 0; s3 < 3; ++s3)   // 6 output tiles x 4 k-pairs = 24 pairs = 3 LDS stages (k-pair outer)
                for_pairs(sg, [&](int i, f16x8 wh, f16x8 wl) {
                    const int lin = 8 * s3 + i, c = lin / 6, o = lin - 6 * c;
#pragma unroll
                    for (int j = 0; j < NT; ++j) f[j][o] = mfma3(wh, wl, xs[j][c], f[j][o]);
                });
            const int f0 = 96 * quarter, nfe = quarter == 3 ? kFeats - 288 : 96, njo = quarter == 3 ? kJoints - 48 : 16;
#pragma unroll
            for (int j = 0; j < NT; ++j) {   // one tile at a time through the wave's staging tile (wave-private: no barrier)
                const bool keep = rvalid[j] && frame[j] < len[j];   // output[~mask.T] = 0 (vae.py:274)
                const int rows_here = tvalid[j] ? min(16, S - rt[j] * 16) : 0;
                const size_t row0 = (size_t)b[j] * S + rt[j] * 16;
#pragma unroll
                for (int o = 0; o < 6; ++o) st4(fst + r * kQStride + 16 * o + 4 * g, keep ? f[j][o] : splat4(0.f));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (a.feats_out) {
                    for (int i = lane; i < rows_here * nfe; i += 64) {
                        const int rr = i / nfe, c = i - rr * nfe;
                        a.feats_out[(row0 + rr) * kFeats + f0 + c] = fst[rr * kQStride + c];
                    }
                }
                if (a.poses_out) {
                    for (int i = lane; i < rows_here * njo; i += 64) {
                        const int rr = i / njo, jn = i - rr * njo;
                        float aa[3];
                        rot6d_to_axis_angle(fst + rr * kQStride + 6 * jn, a.quat_mode, aa);
                        float* dst = a.poses_out + ((row0 + rr) * kJoints + 16 * quarter + jn) * 3;
                        dst[0] = aa[0]; dst[1] = aa[1]; dst[2] = aa[2];
                    }
                }
                if (a.trans_out && quarter == 3) {
                    for (int i = lane; i < rows_here * 3; i += 64) {
                        const int rr = i / 3, c = i - rr * 3;
                        a.trans_out[(row0 + rr) * 3 + c] = fst[rr * kQStride + (330 - 288) + c];
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }
    }
    if constexpr (!kProd) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the two surplus fetches must not outlive the workgroup's LDS
}

template <int W, bool ENC>
hipError_t launch_rows8_w(const VaeRowsArgs& a, hipStream_t stream) {
    static DeviceOnce once;
    int dev_;
    if (!once.done(&dev_)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_vae_rows8x<kTilesPerWave, W, ENC>), hipFuncAttributeMaxDynamicSharedMemorySize, rows8_lds_bytes(W, true));
        if (e != hipSuccess) return e;
        once.set(dev_);
    }
    const int tiles = a.B * a.tiles, per_wg = W * kTilesPerWave;
    hipLaunchKernelGGL((k_vae_rows8x<kTilesPerWave, W, ENC>), dim3((tiles + per_wg - 1) / per_wg), dim3(64 * (W + kProdWaves)), rows8_lds_bytes(W, !ENC && a.stage == kLayers), stream, a);
    return hipGetLastError();
}
template <bool ENC>
hipError_t launch_rows8_mode(const VaeRowsArgs& a, int best, hipStream_t stream) {
    switch (best) {
        case 4: return launch_rows8_w<4, ENC>(a, stream);
        case 5: return launch_rows8_w<5, ENC>(a, stream);
        case 6: return launch_rows8_w<kProd ? 5 : 6, ENC>(a, stream);
        case 8: return launch_rows8_w<8, ENC>(a, stream);
        case 9: return launch_rows8_w<9, ENC>(a, stream);
        case 10: return launch_rows8_w<10, ENC>(a, stream);
        case 11: return launch_rows8_w<11, ENC>(a, stream);
        default: return launch_rows8_w<kProd ? 11 : 12, ENC>(a, stream);
    }
}
}  // namespace

// Waves (= row tiles) per workgroup: the launch's time is its rounds over the chip's 256 CUs times the waves that share a CU's pipes in a
// round, so the shape is chosen per launch to minimise ceil(workgroups / 256) x waves - 256 clips are 4,864 tiles = 19 per CU: two rounds of
// 10-wave workgroups (95 % full) instead of two of 12 (58 % in the second).  All instantiations produce the same bits (a tile's arithmetic does
// not depend on its workgroup).
hipError_t launch_vae_rows8x(const VaeRowsArgs& a, hipStream_t stream, int mode) {
    if (mode != VAE_MODE_DEC && mode != VAE_MODE_ENC) return hipErrorInvalidValue;
    if (mode == VAE_MODE_ENC && a.stage == 0) return hipErrorInvalidValue;   // (the embedding stage is k_vae_rows<f16x2, M_ENC>'s)
    const int tiles = a.B * a.tiles;
    // measured (profiles/r03_rows8_variants.txt): 10 waves beat 11 and 12 at equal rounds, 8 and 9 lose except where 8 waves make more
    // workgroups than CUs busy (launches of up to ~100 clips)
    int best = 8, best_cost = 1 << 30;
    if (tiles > 8 * 256) {
        for (int w : {10, 11, 12}) {
            if (kProd && w == 12) continue;   // (12 row waves + the copying wave would be four waves per SIMD: 128 registers)
            const int wgs = (tiles + w * kTilesPerWave - 1) / (w * kTilesPerWave);
            const int cost = ((wgs + 255) / 256) * w;
            if (cost < best_cost) { best_cost = cost; best = w; }   // (ties: the smaller workgroup)
        }
    }
    return mode == VAE_MODE_ENC ? launch_rows8_mode<true>(a, best, stream) : launch_rows8_mode<false>(a, best, stream);
}

}  // namespace amuse
