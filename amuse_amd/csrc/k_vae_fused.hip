// MotionPrior.decode as ONE persistent kernel per clip (bf16 throughput mode): reference models/latent_diffusion/vae.py:216-278
// = zeros(300,B,128) + learned PE -> SkipTransformerDecoder (cross_attention.py:66-125) of 9 TransformerDecoderLayer.forward_post
// blocks (cross_attention.py:297-345) -> final_layer Linear(128 -> 333) -> 6D -> matrix -> axis-angle (infer_ldm.py:168-173).
//
// Why one kernel: the staged path (k_vae.hip: k_vae_rows / k_vae_attn alternating, 19 launches) moves q, k, v, o and the
// residual stream through HBM between launches (79 MB per attention launch at 256 clips, more than the 32 MB of L2) and
// re-streams a block's 393 KB of weights once per 16-row tile.  Here a clip never leaves its CU:
//
//   * workgroup = one clip = 8 waves, TWO per SIMD (256 registers each): waves w and w+4 share SIMD w and split its five
//     16-row tiles 3 + 2 (wave w < 4: tiles w, w+4, w+8; wave w >= 4: tiles w+8, w+12; 19 real tiles, the 20th is padding),
//     so every SIMD carries the same work and its two waves fill each other's LDS / MFMA-to-VALU / dependency stalls (a
//     first version with ONE 512-register wave per SIMD and five tiles each ran at a quarter of its instruction-issue
//     bound: nothing covers a lone wave's stalls).  A wave keeps its tiles' fp32 residual stream in
//     registers for the whole network, in the row-lane layout of amuse_dev.hpp (accumulators of one GEMM are the operands
//     of the next).  Row work (projections, FFN, LayerNorm, skip linears, final layer) needs no other wave: no split-K, no
//     combine, no exchange.
//   * weights reach the CU ONCE for the four waves: the stream (4.3 MB bf16, consumption order) is cut into 16 KiB stages
//     that the waves copy global -> LDS with LDS-DMA (global_load_lds_dwordx4, two 1 KiB pieces per wave and stage) into a
//     three-buffer ring, two stages ahead of their use; every 1 KiB MFMA A-fragment a wave reads back (one ds_read_b128)
//     feeds one MFMA per row tile of the wave (independent accumulators).  A stage ends with s_waitcnt vmcnt(2) (everything but the pieces just issued has landed) + s_barrier.
//     (A per-wave register ring - the sampling kernels' scheme - does not survive here: at 512 registers hipcc moves the
//     slots through AGPR copies behind s_waitcnt vmcnt(0), one full L2 round trip per 16 units: 1.04 ms per clip against
//     the 0.19 ms of its MFMAs.)
//   * code size is a first-order cost: a decoder block unrolled over the five tiles is ~100 KB of straight-line code, more
//     than the instruction cache, and runs at a few instructions per 100 cycles (measured: the first pass of a loop body
//     18 k cycles, later passes 4 k).  Everything that is per tile and not a weight-sharing GEMM - attention, LayerNorm, the
//     last stage - is therefore a RUNTIME loop over the tiles whose body works on tile slot 0 while the register arrays
//     rotate by one slot per iteration (moves instead of dynamic register indexing).
//   * the other cross-wave traffic is K_h and V_h^T of the current head: written to LDS as ready MFMA fragments (one
//     ds_write_b128 / two ds_write_b64 per row tile); the barrier that ends the k,v stage
//     publishes them.  Scores S^T = K.Q^T and O^T = V^T.P^T run on v_mfma_f32_16x16x32_bf16 with the softmax along
//     registers, two query tiles at a time on shared K / V fragments; the attention output of head h is, unchanged, the B
//     operand of out_proj's k-slice h, accumulated straight into the residual registers.
//   * FFN: 16 chunks of 32 hidden features, software-pipelined - linear1 of chunk i+1 (MFMA) runs beside the GELU of chunk
//     i (VALU), then linear2 of chunk i accumulates into the residual registers.
//   * the U-Net skip stack (4 x 300 x 128) goes to global memory as packed bf16 MFMA operands - exactly the rounding the
//     skip linear applies anyway - written and read back by the same lanes (no synchronisation); 82 MB per 256 clips.
//   * the one-token cross-attention is the per-clip constant of k_vae_ca (k_misc.hip), as in the staged path.
//
// HBM traffic per clip: 320 KB skip write + 320 KB skip read + 201.6 KB poses/trans out + 4.6 KB constants in; weights
// come out of L2 (4.3 MB per clip and CU).
#include "amuse_fused.hpp"

#define OP_KERNEL OP_SUFFIX(k_vae_fused)
#define OP_LAUNCH OP_SUFFIX(launch_vae_fused)

namespace amuse {
namespace {

// everything one wave does for its NT row tiles tile0, tile0 + 4, ...
template <int NT, bool TAP, bool NOATTN>
__device__ __forceinline__ void decode_tiles(const VaeFusedArgs& a, char* smem, Stager& sg, int tile0, int b, int len,
                                             int wave, int lane) {
    const int g = lane >> 4, r = lane & 15;
    char* kv = smem + kOffKv;
    const float* pvl = reinterpret_cast<const float*>(smem + kOffPv);   // [2][kPvSlot / 4]
    const float* cal = reinterpret_cast<const float*>(smem + kOffCa);   // [9][128]
    const unsigned lds0 = lds_addr(smem);
    uint4* skipbuf = a.skip + (size_t)b * (4 * 20 * 4 * 64);
    // queries = zeros + query_pos_decoder.pe[:300]  (vae.py:220,258)
    f32x4 x[NT][kTiles];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int frame = 16 * (tile0 + 4 * j) + r;
#pragma unroll
        for (int t = 0; t < kTiles; ++t)
            x[j][t] = frame < kFrames ? ld4(a.pe + (size_t)frame * kD + 16 * t + 4 * g) : splat4(0.f);
    }
    asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // parameters, constants and stage 0 are in
#pragma unroll 1
    for (int blk = 0; blk < 4; ++blk)
        decoder_block<NT, 0, TAP, false, NOATTN>(x, sg, a.pvec, a.tap_out, blk, tile0, pvl + (blk & 1) * (kPvSlot / 4), a.pvec + (blk + 1) * PV_BLOCK,
                             lds0 + kOffPv + ((blk + 1) & 1) * kPvSlot, cal, kv, skipbuf, len, wave, lane, kFrames,
                             (blk == 0 && len == kFrames) ? a.c1 : nullptr);
    decoder_block<NT, 1, TAP, false, NOATTN>(x, sg, a.pvec, a.tap_out, 4, tile0, pvl, a.pvec + 5 * PV_BLOCK, lds0 + kOffPv + kPvSlot, cal, kv, skipbuf, len, wave, lane);
#pragma unroll 1
    for (int blk = 5; blk < kLayers; ++blk)
        decoder_block<NT, 2, TAP, false, NOATTN>(x, sg, a.pvec, a.tap_out, blk, tile0, pvl + (blk & 1) * (kPvSlot / 4),
                             blk + 1 < kLayers ? a.pvec + (blk + 1) * PV_BLOCK : nullptr,
                             lds0 + kOffPv + ((blk + 1) & 1) * kPvSlot, cal, kv, skipbuf, len, wave, lane);
    // ---------------- decoder.norm -> final_layer (333 outputs in 24 tiles) -> rotation epilogue.
    // The last stage leaves the stage protocol: the WHOLE final_layer image (96 units, once in the stream) goes to LDS in one go - everything of the
    // nine blocks is dead by now: K/V images, ring, parameter slots - together with its bias and decoder.norm's parameters, and from there on no wave
    // touches vmcnt or a barrier again: a tile's 96 MFMAs, its staging tile (wave-private) and its output stores run free.  (On gfx950 stores count in
    // vmcnt like loads: with the layer streamed through the ring - five copies, 30 stages - every stage end and every bias load sat out the round trip of
    // the output stores in front of it: 12-14 % of the launch, profiles/r04_decode_output_store_ablation.txt.)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the ring's two stages ahead have landed; block 8's last LDS reads are done
    {
        const uint4* fsrc = sg.src - (size_t)(2 * kStage + 2 * wave) * 64 + (size_t)wave * 12 * 64;   // (sg.src: this wave's pieces of the stage after next)
#pragma unroll
        for (int i = 0; i < 12; ++i) glds16(fsrc + i * 64, lds0 + kOffFinalW + (wave * 12 + i) * 1024);
        float* lpar = reinterpret_cast<float*>(smem + kOffFinalPar);
        const int t = wave * 64 + lane;
        if (t < 96) st4(lpar + 4 * t, ld4(a.final_bias + 4 * t));
        else if (t < 160) st4(lpar + 4 * t, ld4(a.pvec + PV_FINAL_W + 4 * (t - 96)));   // decoder.norm weight | bias (contiguous in the parameter vector)
        static_assert(PV_FINAL_B == PV_FINAL_W + kD, "decoder.norm parameters are read as one run");
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const float* lbias = reinterpret_cast<const float*>(smem + kOffFinalPar);
    const float* lnorm = lbias + 384;
    const char* wimg = smem + kOffFinalW + lane * 16;
    float* fst = reinterpret_cast<float*>(smem + kOffFinalStage) + wave * 16 * kQStride;
#pragma unroll 1
    for (int j = 0; j < NT; ++j) {
        const int tile = tile0 + 4 * j;
        const int rows_here = min(16, kFrames - 16 * tile);   // <= 0 for the padding tile
        if (rows_here <= 0) {
            rotate_tiles<NT>(x);
            continue;
        }
        layer_norm_rows<true>(x[0], lnorm, lnorm + kD, g);
        if constexpr (TAP) {   // slot 9: decoder.norm of this tile
            const int frame_t = 16 * tile + r;
            if (a.tap_out && blockIdx.x == 0 && frame_t < kFrames) {
#pragma unroll
                for (int t = 0; t < kTiles; ++t) st4(a.tap_out + ((size_t)9 * kFrames + frame_t) * kD + 16 * t + 4 * g, x[0][t]);
            }
        }
        OPV xb1[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) xb1[c] = OP_PACK(x[0][2 * c], x[0][2 * c + 1]);
        rotate_tiles<NT>(x);
        const int frame = 16 * tile + r;
        const bool keep = frame < kFrames && frame < len;   // output[~mask.T] = 0 (vae.py:274)
        const size_t row0 = (size_t)b * kFrames + 16 * tile;
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
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                const int quarter = 2 * half + qq;
#pragma unroll
                for (int o = 0; o < 6; ++o) st4(fst + r * kQStride + 16 * o + 4 * g, keep ? f[6 * qq + o] : splat4(0.f));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the staging tile is wave-private: no barrier
                const int f0 = 96 * quarter, nfe = quarter == 3 ? kFeats - 288 : 96, njo = quarter == 3 ? kJoints - 48 : 16;
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
}

template <bool TAP, bool NOATTN = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void OP_KERNEL(VaeFusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x;
    const int len = a.lengths ? a.lengths[b] : kFrames;
    const unsigned lds0 = lds_addr(smem);
    // block 0's parameters, the clip's cross-attention constants, then the first two weight stages
    glds16(reinterpret_cast<const uint4*>(a.pvec) + wave * 64 + lane, lds0 + kOffPv + wave * 1024);
    if (wave < 5) glds16(reinterpret_cast<const uint4*>(a.ca + (size_t)b * kLayers * kD) + wave * 64 + lane, lds0 + kOffCa + wave * 1024);
    Stager sg;
    // (a full-length clip with the block-0 constant at hand starts behind block 0's eight attention stages: decoder_block, c1)
    const size_t skip_units = (a.c1 && len == kFrames) ? (size_t)8 * kStage : 0;
    sg.src = a.wstream + (skip_units + (size_t)wave * 2) * 64 + lane;
    sg.dst0 = lds0 + kOffW + wave * 2048;
    sg.ring = smem + kOffW + lane * 16;
    sg.widx = 0;
    sg.ridx = 0;
    stage_fetch(sg);
    stage_fetch(sg);
    if (wave < 4) decode_tiles<3, TAP, NOATTN>(a, smem, sg, wave, b, len, wave, lane);
    else decode_tiles<2, TAP, NOATTN>(a, smem, sg, wave + 8, b, len, wave, lane);
}

}  // namespace

hipError_t OP_LAUNCH(const VaeFusedArgs& a, hipStream_t stream) {
    static DeviceOnce once;
    int dev_;
    if (!once.done(&dev_)) {
        for (const void* k : {reinterpret_cast<const void*>(&OP_KERNEL<false>), reinterpret_cast<const void*>(&OP_KERNEL<true>),
                              reinterpret_cast<const void*>(&OP_KERNEL<false, true>)}) {
            hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, kFusedFinalLdsBytes);
            if (e != hipSuccess) return e;
        }
        once.set(dev_);
    }
#if AMUSE_FPROF
    {
        int zero = 0;
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fprof_n), &zero, sizeof(int));
    }
#endif
    if (a.ablate_attention) hipLaunchKernelGGL((OP_KERNEL<false, true>), dim3(a.B), dim3(512), kFusedFinalLdsBytes, stream, a);   // timing ablation (bench.py)
    else if (a.tap_out) hipLaunchKernelGGL(OP_KERNEL<true>, dim3(a.B), dim3(512), kFusedFinalLdsBytes, stream, a);   // the tapped instantiation (tests)
    else hipLaunchKernelGGL(OP_KERNEL<false>, dim3(a.B), dim3(512), kFusedFinalLdsBytes, stream, a);
#if AMUSE_FPROF
    {
        static int calls = 0;
        (void)hipStreamSynchronize(stream);
        unsigned long long h[512];
        int n = 0;
        (void)hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_fprof_n), sizeof(int));
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_fprof), sizeof(h));
        if (++calls == 3) {
            for (int i = 1; i < n; ++i) fprintf(stderr, "FPROF %3d tag %2llu  +%llu\n", i, h[2 * i + 1], h[2 * i] - h[2 * i - 2]);
        }
    }
#endif
    return hipGetLastError();
}

}  // namespace amuse
