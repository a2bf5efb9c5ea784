// The building blocks of the persistent per-clip kernels (k_vae_fused.hip: MotionPrior.decode; k_den_fused.hip: one step of the
// diffusion_only Denoiser): operand-type macros of the two builds (bf16 / fp16), LDS map, LDS-DMA weight-stage ring (Stager),
// fragment-walking GEMM cores, and the S ~ 300 attention of one 16-query tile against K / V^T fragment images in LDS.
// Included by a .hip AFTER it has (optionally) defined AMUSE_OP_F16; everything lives in namespace amuse's anonymous namespace.
#pragma once
#include "amuse_dev.hpp"
#include "amuse_kernels.hpp"
#include <cstdio>

// Every kernel built on this header is compiled twice: as is (bf16 operands: k_vae_fused / launch_vae_fused, ...) and through a
// two-line *h.hip with AMUSE_OP_F16 defined (fp16 operands, AMUSE_PREC_F16: k_vae_fusedh / launch_vae_fusedh - the same instruction
// stream on the fp16 MFMA, fp16 weight stream, K / V^T / skip-stack images in fp16, GELU polynomial one degree higher).
#ifdef AMUSE_OP_F16
#define OPV f16x8
#define OP_PACK pack_f16
#define OP_MFMA mfma_f16
#define OP_GELU gelu_poly4h
#define OP_CVT4 f32_to_f16x4
#define OP_ONE2 0x3c003c00u
#define OP_SUFFIX(name) name##h
#else
#define OPV bf16x8
#define OP_PACK pack_bf16
#define OP_MFMA mfma_bf16
#define OP_GELU gelu_poly4
#define OP_CVT4 f32_to_bf16x4
#define OP_ONE2 0x3f803f80u
#define OP_SUFFIX(name) name
#endif

namespace amuse {
namespace {

// (The skip stack - 78 MB each way at 256 clips, written once and read once 0.1 - 0.5 ms later - goes through the caches: non-temporal
// stores / loads measured flat, profiles/r04_decode_skip_nt_ab.txt.)
constexpr int kWaves = 8;
constexpr int kKeyRows = 320;             // 300 keys padded to 20 tiles
constexpr int kPairs = kKeyRows / 32;     // 10 key-tile pairs
constexpr int kKvBytes = kKeyRows * 64 + kPairs * 2 * 16 * 64;   // K fragments 20 KiB + V^T fragments 20 KiB
constexpr int kStage = kVaeFusedStageUnits;                      // 16 units = 16 KiB per stage
constexpr int kStageBytes = kStage * 1024;
constexpr int kWBufs = 3;
constexpr int kPvSlot = 8192;             // one block's small parameters (7,680 B) rounded up to whole DMA pieces
constexpr int kCaBytes = 5120;            // [9][128] floats rounded up to whole DMA pieces
constexpr int kQStride = 100;            // staging row stride (floats) of one 96-feature quarter (16 joints) of the last stage
// LDS map (bytes)
constexpr int kOffKv = 0;                                  // K/V images of the current head; the last stage's 4 staging tiles reuse it
constexpr int kOffW = kOffKv + kKvBytes;                   // weight ring
constexpr int kOffPv = kOffW + kWBufs * kStageBytes;       // 2 x block parameters
constexpr int kOffCa = kOffPv + 2 * kPvSlot;               // cross-attention constants of the clip
static_assert(kOffCa + kCaBytes == kVaeFusedLdsBytes, "LDS layout and amuse_kernels.hpp disagree");
// LDS map of the LAST stage (final_layer / pose_proj; the blocks' map above is dead by then): the projection's whole image (96 units) | one staging
// tile per wave | its bias + the last norm's parameters.  The kernels are launched with this (larger) size.
constexpr int kOffFinalW = 0;
constexpr int kOffFinalStage = 96 * 1024;
constexpr int kOffFinalPar = kOffFinalStage + kWaves * 16 * kQStride * 4;
constexpr int kFusedFinalLdsBytes = kOffFinalPar + (384 + 2 * kD) * 4;   // 152,064 B
static_assert(kFusedFinalLdsBytes >= kVaeFusedLdsBytes && kFusedFinalLdsBytes <= 160 * 1024, "LDS");

// -DAMUSE_FPROF=1 (variant builds only): wave 0 of workgroup 0 stamps s_memtime at phase boundaries of blocks 1 and 6 and
// OP_LAUNCH prints the deltas (tools/gpu_decode_phases.py)
#ifndef AMUSE_FPROF
#define AMUSE_FPROF 0
#endif
// weight fragments are read from the LDS ring this many units ahead of their MFMAs
constexpr int kFragAhead = 2;
#if AMUSE_FPROF
__device__ unsigned long long g_fprof[512];
__device__ int g_fprof_n;
#define FSTAMP(tag)                                                                      \
    do {                                                                                 \
        if (prof_on) {                                                                   \
            const int i_ = g_fprof_n;                                                    \
            if (i_ < 255) { g_fprof[2 * i_] = __builtin_readcyclecounter(); g_fprof[2 * i_ + 1] = (tag); g_fprof_n = i_ + 1; } \
        }                                                                                \
    } while (0)
#else
#define FSTAMP(tag) do { } while (0)
#endif

// ---- LDS-DMA: 64 lanes x 16 B from per-lane global addresses to LDS [dst, dst + 1 KiB), lane-linear.  Inline asm: the
// compiler neither counts it in its s_waitcnt bookkeeping (its own waits can only become longer, never too short: vmcnt
// retires in order) nor drains it at barriers; the stage protocol below does the counting.
__device__ __forceinline__ void glds16(const uint4* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}

struct Stager {
    const uint4* src;   // this lane's source address of the wave's pieces of the NEXT stage to fetch
    unsigned dst0;      // LDS byte address of the wave's pieces inside buffer 0
    const char* ring;   // weight ring base (generic pointer) + lane * 16
    int widx, ridx;     // buffer the next fetch fills / buffer the current stage reads
};
// (A wave-uniform source base in SGPRs + a 32-bit lane offset - the saddr form of global_load_lds_dwordx4 - and 32-bit LDS addresses for
// the ring were tried in round 4 to free the four registers these pointers hold: hipcc's allocation got WORSE, 28 -> 46 spilled registers.)
__device__ __forceinline__ void stage_fetch(Stager& s) {
    const unsigned d = __builtin_amdgcn_readfirstlane(s.dst0 + s.widx * kStageBytes);
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16(s.src + i * 64, d + i * 1024);
    s.src += kStage * 64;
    s.widx = s.widx == kWBufs - 1 ? 0 : s.widx + 1;
}
// end of a stage: all of this wave's DMA except the two pieces of the fetch issued in this stage has landed, its LDS reads
// and writes are done; after the barrier that holds for every wave - the next stage's buffer is complete, this stage's is free
__device__ __forceinline__ void stage_end(Stager& s) {
    asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    s.ridx = s.ridx == kWBufs - 1 ? 0 : s.ridx + 1;
}
__device__ __forceinline__ OPV wfrag(const Stager& s, int u) {
    return __builtin_bit_cast(OPV, *reinterpret_cast<const uint4*>(s.ring + s.ridx * kStageBytes + u * 1024));
}

// f(u, fragment) for the units U0 .. U0 + NU - 1 of the current stage, the fragments read kFragAhead units ahead of their use.
// (Left to hipcc, a unit loop recycles ONE fragment register: read, wait for the whole LDS round trip, MFMAs, next read - the
// k,v stage ran at a third of its MFMA rate that way.)
template <int NU, int U0, class F>
__device__ __forceinline__ void for_units(const Stager& s, F&& f) {
    constexpr int PF = kFragAhead < NU ? kFragAhead : NU;
    OPV wf[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) wf[u] = wfrag(s, U0 + u);
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const OPV cur = wf[u % PF];
        if (u + PF < NU) wf[u % PF] = wfrag(s, U0 + u + PF);
        f(u, cur);
    }
}

// acc[j][o] += W_o . x_j for the NT row tiles of this wave; units U0.. of the current stage, k-pair outer, output tile inner
template <int NT, int NO, int NC, int U0>
__device__ __forceinline__ void gemm5(f32x4 (&acc)[NT][NO], const OPV (&xb)[NT][NC], const Stager& s) {
    // fragments are read two ahead of their MFMAs (a read waited for on the spot costs an LDS round trip per unit)
    constexpr int NU = NO * NC, PF = kFragAhead;
    OPV wf[NU < PF ? NU : PF];
#pragma unroll
    for (int u = 0; u < PF && u < NU; ++u) wf[u] = wfrag(s, U0 + u);
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int c = u / NO, o = u - c * NO;
        const OPV cur = wf[u % PF];
        if (u + PF < NU) wf[u % PF] = wfrag(s, U0 + u + PF);
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j][o] = OP_MFMA(cur, xb[j][c], acc[j][o]);
    }
}

// x[0] <- x[1] <- ... <- x[NT-1] <- x[0]: NT applications restore the order
template <int NT>
__device__ __forceinline__ void rotate_tiles(f32x4 (&x)[NT][kTiles]) {
#pragma unroll
    for (int t = 0; t < kTiles; ++t) {
        const f32x4 first = x[0][t];
#pragma unroll
        for (int j = 0; j + 1 < NT; ++j) x[j][t] = x[j + 1][t];
        x[NT - 1][t] = first;
    }
}

template <int NT>
__device__ __forceinline__ void pack_rows(OPV (&xb)[NT][4], const f32x4 (&x)[NT][kTiles]) {
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) xb[j][c] = OP_PACK(x[j][2 * c], x[j][2 * c + 1]);
}

// 16-byte slot of lane (g, r) inside a 1 KiB K / V^T fragment.  Lane-linear (4 r + g) makes ds_read_b128 2-way bank
// conflicted: its four 16-lane groups ({0-3,12-15,20-27}, ...) pair rows r and r + 12 / r + 4 and r + 8 of the same g on
// the same 16-byte bank quad.  With the row quad r >> 2 in the low bits every group touches 16 distinct quads.
__device__ __forceinline__ int frag_slot(int g, int r) { return 16 * g + 4 * (r & 3) + (r >> 2); }

// max of three.  The file is built with -fno-honor-nans (Makefile; no NaN can reach the scores: finite operands, -inf only through
// the mask): hipcc then drops the v_max_f32 x, x canonicalisation it otherwise puts in front of every fmaxf operand and fuses pairs
// into v_max3_f32.  Not inline asm: the hazard recogniser does not see through asm, and a VALU read of an MFMA result needs
// software wait states (k_audio.hip's attention read stale score registers through an asm v_max3_f32).
__device__ __forceinline__ float max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

// softmax(Q K^T) V of ONE 16-query tile against all keys of head h, K / V^T fragments in LDS.  Five chunks of two key-tile
// pairs (64 keys), merged online.  What shapes the loop: a lone ds_read_b128 -> s_waitcnt -> MFMA sequence exposes the whole
// LDS latency per fragment (the first version did exactly that: 40 round trips per tile, 5 k cycles), and registers are
// too scarce (x + xb of three tiles = 144 of 256) for hipcc to hoist the reads itself.  So a chunk's four K fragments are
// read as ONE batch in front of its four score MFMAs, and its four V^T fragments as one batch right behind them - they
// land while the softmax arithmetic runs.
// (lazy rescaling of the running maxima: kAttnTau, amuse_dev.hpp)
template <bool NOATTN = false>
__device__ __forceinline__ OPV attend(const uint4* Kb, const uint4* Vt, OPV qb, int len, int g, int r) {
    if constexpr (NOATTN) return qb;
    const int fs = frag_slot(g, r);
    // Scores leave the MFMAs RELATIVE to the row's running maximum (C operand = -m_run; chunk 0 starts from 0 and takes its own
    // maximum - the sequence has at least one key), so in the common chunk - the maximum did not move for any row of the wave -
    // p = exp2(result): no subtraction, no rescale (k_audio.hip's attention has the same scheme).
    float m_run = 0.f;
    f32x4 o[2] = {splat4(0.f), splat4(0.f)};
    // The row sums of the softmax ride the matrix pipe: a constant "V^T" fragment whose row d = 0 is all ones makes
    // O^T[0][i] = sum_key P[i][key] - two MFMAs per chunk instead of sixteen v_add_f32 and a butterfly (the attention is
    // VALU-bound: 16 quarter-rate v_exp_f32 per lane and chunk are half of it, the rest was max / sum / pack).  What is
    // summed is the bf16 P the PV product uses.  Lane (g = 0, i) ends up with query i's sum in os[0]; rows d = 4, 8, 12 of the
    // fragment are zero, so os[0] of the other three lanes of the row is 0 and one butterfly at the end broadcasts it.
    const OPV ones = __builtin_bit_cast(OPV, r == 0 ? uint4{OP_ONE2, OP_ONE2, OP_ONE2, OP_ONE2} : uint4{0u, 0u, 0u, 0u});
    f32x4 os = splat4(0.f);
#pragma unroll
    for (int ch = 0; ch < kPairs / 2; ++ch) {
        uint4 kf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) kf[i] = Kb[(4 * ch + i) * 64 + fs];
        f32x4 st[4];
        const f32x4 c0 = splat4(-m_run);
#pragma unroll
        for (int i = 0; i < 4; ++i)   // lane (g, i): S[query i][key 64 ch + 16 i + 4 g + m] - m_run (log2 units)
            st[i] = OP_MFMA(__builtin_bit_cast(OPV, kf[i]), qb, c0);
        uint4 vf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) vf[i] = Vt[(4 * ch + i) * 64 + fs];   // (pair, td) = (2 ch + i / 2, i % 2)
        __builtin_amdgcn_sched_barrier(0);   // keep both batches where they are
        // key-padding mask: only a chunk that reaches past the sequence end is touched (wave-uniform branch; with len = 300
        // that is the last chunk alone).  The per-lane limit is recomputed here on purpose: hoisted out of the head loop the
        // lane masks would live in SGPR pairs and spill.
        const int k0 = 64 * ch;
        if (k0 + 64 > len) {
            int lim = len - k0 - 4 * g;   // element (i, m) of the chunk is valid iff 16 i + m < lim
            asm volatile("" : "+v"(lim));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int m = 0; m < 4; ++m) st[i][m] = (16 * i + m < lim) ? st[i][m] : -INFINITY;
        }
        // this lane's 16 scores of the chunk; the row's maximum (four lanes) is only formed when it matters: a chunk moves the
        // running maximum of SOME row of the wave iff some lane holds a positive score (scores are relative to m_run)
        float mx = max3(max3(st[0][0], st[0][1], st[0][2]), max3(st[0][3], st[1][0], st[1][1]), max3(st[1][2], st[1][3], st[2][0]));
        mx = max3(mx, max3(st[2][1], st[2][2], st[2][3]), max3(st[3][0], st[3][1], st[3][2]));
        mx = fmaxf(mx, st[3][3]);
        if (ch == 0 || __builtin_amdgcn_ballot_w64(mx > kAttnTau) != 0) {   // (wave-uniform)
            mx = allreduce_g_max(mx);   // the same in the four lanes of a row; -inf for a fully masked chunk (ch > 0 only)
            const float d = ch == 0 ? mx : fmaxf(mx, 0.f);
#pragma unroll
            for (int i = 0; i < 4; ++i) st[i] -= splat4(d);
            if (ch > 0) {
                const float alpha = __builtin_amdgcn_exp2f(-d);
                os *= alpha;
                o[0] *= alpha;
                o[1] *= alpha;
            }
            m_run += d;
        }
        f32x4 p[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // (key tile 19 = keys 304..319 lies beyond every sequence - kFrames = 300 -: its probabilities are 0 without asking the
            // quarter-rate exponential; the mask above has set its scores to -inf, which the row maximum ignores)
            const bool beyond = 64 * ch + 16 * i >= kFrames;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                p[i][m] = beyond ? 0.f : __builtin_amdgcn_exp2f(st[i][m]);
            }
        }
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {   // O^T[d][i] += sum_key V[key][d] P[i][key], 32 keys per MFMA
            const OPV pb = OP_PACK(p[2 * pr], p[2 * pr + 1]);
            o[0] = OP_MFMA(__builtin_bit_cast(OPV, vf[2 * pr]), pb, o[0]);
            o[1] = OP_MFMA(__builtin_bit_cast(OPV, vf[2 * pr + 1]), pb, o[1]);
            os = OP_MFMA(ones, pb, os);
        }
    }
    const float inv = __builtin_amdgcn_rcpf(allreduce_g_sum(os[0]));
    return OP_PACK(o[0] * inv, o[1] * inv);
}

// MODE 0: input block (push the skip), 1: middle block, 2: output block (skip linear first)
// debugging taps (TAP instantiation only, clip 0): the wave's tiles of the fp32 residual stream, row-major [300][128]
template <int NT>
__device__ __forceinline__ void store_tap(float* tap, int slot, const f32x4 (&x)[NT][kTiles], int tile0, int g, int r) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int frame = 16 * (tile0 + 4 * j) + r;
        if (frame < kFrames) {
#pragma unroll
            for (int t = 0; t < kTiles; ++t) st4(tap + ((size_t)slot * kFrames + frame) * kD + 16 * t + 4 * g, x[j][t]);
        }
    }
}

// ENCL: a TransformerEncoderLayer (cross_attention.py:259-272: no cross-attention, norm2 behind the FFN) instead of the
// TransformerDecoderLayer with its one-token memory - the diffusion_only Denoiser's blocks (k_den_fused.hip).  S = rows of the clip
// (300 frames; 302..304 with the Denoiser's condition tokens in front).  pvec_g: the global parameter vector (skip-linear biases).
// NOATTN: the timing ablation behind bench.py's attention-only figure (amuse_debug_set_ablation): the block without its
// softmax(Q K^T) V - q, k, v, the K / V images, out_proj and every barrier stay, so full - ablated = the attention's time.
template <int NT, int MODE, bool TAP, bool ENCL = false, bool NOATTN = false>
__device__ __forceinline__ void decoder_block(f32x4 (&x)[NT][kTiles], Stager& sg, const float* pvec_g, float* tap_out, int blk, int tile0,
                                              const float* pv, const float* pv_next_src, unsigned pv_next_dst, const float* cal,
                                              char* kv, uint4* skipbuf, int len, int wave, int lane, const int S = kFrames,
                                              const float* c1 = nullptr) {
    const int g = lane >> 4, r = lane & 15;
    [[maybe_unused]] const bool prof_on = blockIdx.x == 0 && threadIdx.x == 0 && (blk == 1 || blk == 6);
    // c1 (block 0 of the DECODER, full-length clips only): the decoder's input is zeros + query_pos_decoder.pe (vae.py:220,252-259) and the
    // latent enters through the cross-attention alone, so block 0's self-attention half - q, k, v, softmax(q k^T) v, out_proj, norm1 - is the
    // same [300][128] array for every clip of a weight set.  The library computes it once per weight set WITH THIS KERNEL (the tapped
    // instantiation's slot 10: the same instruction stream, hence the same bits) and the block starts from it; ragged clips (another key mask)
    // take the full path.  The kernel then starts its weight stream behind block 0's eight attention stages (OP_KERNEL).
    const bool hoist = MODE == 0 && !ENCL && c1 != nullptr;
    FSTAMP(1);   // block start
    OPV xb[NT][4];
    if constexpr (MODE == 2) {
        // x = linear_blocks[blk - 5](cat(x, xs.pop()))   (cross_attention.py:118-120); the popped skip comes back from
        // global memory as the packed operands this wave stored after input block 8 - blk.  Four stages: the x half
        // (k-pairs 0..3), then the skip half.
        OPV sb[NT][4];
        const uint4* sk = skipbuf + (size_t)(8 - blk) * (20 * 4 * 64);
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                sb[j][c] = __builtin_bit_cast(OPV, sk[((tile0 + 4 * j) * 4 + c) * 64 + lane]);
            }
        pack_rows<NT>(xb, x);
        const float* bias = pvec_g + PV_SKIP_B + (blk - 5) * kD;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int t = 0; t < kTiles; ++t) x[j][t] = ld4(bias + 16 * t + 4 * g);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            stage_fetch(sg);
            for_units<2 * kTiles, 0>(sg, [&](int u, OPV wf) {
                const int c = 2 * (s4 & 1) + u / kTiles, o = u % kTiles;
#pragma unroll
                for (int j = 0; j < NT; ++j) x[j][o] = OP_MFMA(wf, s4 < 2 ? xb[j][c] : sb[j][c], x[j][o]);
            });
            stage_end(sg);
        }
    }
    FSTAMP(2);   // skip linear done
    if (hoist) {
        if (pv_next_src) {   // (the next block's small parameters: otherwise issued inside the head loop)
            const unsigned d = __builtin_amdgcn_readfirstlane(pv_next_dst + wave * 1024);
            glds16(reinterpret_cast<const uint4*>(pv_next_src) + wave * 64 + lane, d);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int frame = 16 * (tile0 + 4 * j) + r;
#pragma unroll
            for (int t = 0; t < kTiles; ++t) x[j][t] = frame < kFrames ? ld4(c1 + (size_t)frame * kD + 16 * t + 4 * g) : splat4(0.f);
        }
    } else {
    // ---------------- self-attention (cross_attention.py:323-330): x = norm1(x + out_proj(softmax(q k^T) v))
    pack_rows<NT>(xb, x);
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int t = 0; t < kTiles; ++t) x[j][t] += ld4(pv + PV_OUT_B + 16 * t + 4 * g);
    constexpr float kQScale = 0.17677669529663687f * 1.44269504088896340736f;  // 1/sqrt(32) * log2(e): softmax in exp2
#pragma unroll 1
    for (int h = 0; h < kHeads; ++h) {
        char* buf = kv;   // ONE image: the barrier that ends stage B of head h - 1 is behind every wave's last read of it
        uint4* Kb = reinterpret_cast<uint4*>(buf);
        char* Vt = buf + kKeyRows * 64;
        // ---- stage A: k, v of this head for the wave's rows -> LDS fragment images (published by the stage's barrier)
        stage_fetch(sg);
        {
            f32x4 kk[NT][2], vv[NT][2];
            const f32x4 bk0 = ld4(pv + PV_IN_B + kD + 32 * h + 4 * g), bk1 = ld4(pv + PV_IN_B + kD + 32 * h + 16 + 4 * g);
            const float bv0 = pv[PV_IN_B + 2 * kD + 32 * h + r], bv1 = pv[PV_IN_B + 2 * kD + 32 * h + 16 + r];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                kk[j][0] = bk0; kk[j][1] = bk1;
                vv[j][0] = splat4(bv0); vv[j][1] = splat4(bv1);
            }
            for_units<16, 0>(sg, [&](int u, OPV wf) {   // stream: per k-pair c: k tiles (2), v tiles (2)
                const int c = u >> 2, t = u & 3;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    if (t < 2) kk[j][t] = OP_MFMA(wf, xb[j][c], kk[j][t]);
                    else vv[j][t - 2] = OP_MFMA(xb[j][c], wf, vv[j][t - 2]);   // operand-swapped: V^T
                }
            });
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int tile = tile0 + 4 * j;
                const bool ok = 16 * tile + r < S;
                const uint4 kf = __builtin_bit_cast(uint4, OP_PACK(kk[j][0], kk[j][1]));
                Kb[tile * 64 + frag_slot(g, r)] = ok ? kf : uint4{0u, 0u, 0u, 0u};
                // V^T: lane (g, d) holds V[row 4 g + m][16 td + d]; rows beyond the clip are zeroed (0 x p stays 0)
#pragma unroll
                for (int td = 0; td < 2; ++td) {
                    f32x4 v = vv[j][td];
#pragma unroll
                    for (int m = 0; m < 4; ++m) v[m] = (16 * tile + 4 * g + m < S) ? v[m] : 0.f;
                    *reinterpret_cast<uint2*>(Vt + (((tile >> 1) * 2 + td) * 64 + frag_slot(g, r)) * 16 + (tile & 1) * 8) = OP_CVT4(v);
                }
            }
        }
        FSTAMP(3);   // k, v computed and written
        stage_end(sg);
        FSTAMP(4);   // barrier of stage A passed
        // ---- stage B: q of this head, attention, out_proj's k-slice of the head
        if (h == 0 && pv_next_src) {   // next block's small parameters -> the other LDS slot (free since the last barrier)
            const unsigned d = __builtin_amdgcn_readfirstlane(pv_next_dst + wave * 1024);
            glds16(reinterpret_cast<const uint4*>(pv_next_src) + wave * 64 + lane, d);
        }
        stage_fetch(sg);
        OPV qb[NT];
        {
            f32x4 q[NT][2];
            const f32x4 bq0 = ld4(pv + PV_IN_B + 32 * h + 4 * g), bq1 = ld4(pv + PV_IN_B + 32 * h + 16 + 4 * g);
#pragma unroll
            for (int j = 0; j < NT; ++j) { q[j][0] = bq0; q[j][1] = bq1; }
            gemm5<NT, 2, 4, 0>(q, xb, sg);
#pragma unroll
            for (int j = 0; j < NT; ++j) qb[j] = OP_PACK(q[j][0] * kQScale, q[j][1] * kQScale);
        }
        FSTAMP(5);   // q
        // attention, one 16-query tile per iteration of a RUNTIME loop over ONE rotating array: the tile's q operand leaves
        // at the front (ob[0]) and its output enters at the back, so after NT iterations ob[] holds the outputs in tile order
        // (register moves instead of indexing; a second array for the outputs would cost NT more live operands).
        // The loop body is the attention's only copy in the instruction stream (see the note on code size at the kernel).
        OPV ob[NT][1];
#pragma unroll
        for (int j = 0; j < NT; ++j) ob[j][0] = qb[j];
        {
            const uint4* Vq = reinterpret_cast<const uint4*>(Vt);
#pragma unroll 1
            for (int j = 0; j < NT; ++j) {
                const OPV o1 = attend<NOATTN>(Kb, Vq, ob[0][0], len, g, r);
#pragma unroll
                for (int jj = 0; jj + 1 < NT; ++jj) ob[jj][0] = ob[jj + 1][0];
                ob[NT - 1][0] = o1;
            }
            FSTAMP(7);   // attention of the five tiles
        }
        gemm5<NT, kTiles, 1, 8>(x, ob, sg);   // out_proj, k-slice of head h, accumulated into the residual
        FSTAMP(8);   // out_proj
        stage_end(sg);
        FSTAMP(9);   // barrier of stage B passed
    }
    }
    [[maybe_unused]] const float* ca = cal + blk * kD;
    if constexpr (ENCL) {
#pragma unroll 1
        for (int j = 0; j < NT; ++j) {   // runtime loop, the tiles rotate through x[0]
            layer_norm_rows<true>(x[0], pv + PV_LN1_W, pv + PV_LN1_B, g);
            rotate_tiles<NT>(x);
        }
    } else {
#pragma unroll 1
    for (int j = 0; j < NT; ++j) {   // runtime loop, the tiles rotate through x[0]
        if (!hoist) layer_norm_rows<true>(x[0], pv + PV_LN1_W, pv + PV_LN1_B, g);
        if constexpr (TAP && MODE == 0) {   // slot 10 of the taps: block 0 behind norm1 - what the hoist loads (see c1 above)
            const int frame = 16 * (tile0 + 4 * j) + r;
            if (tap_out && blk == 0 && blockIdx.x == 0 && frame < kFrames) {
#pragma unroll
                for (int t = 0; t < kTiles; ++t) st4(tap_out + ((size_t)10 * kFrames + frame) * kD + 16 * t + 4 * g, x[0][t]);
            }
        }
        // cross-attention onto the single latent token == per-clip constant; x = norm2(x + ca)  (cross_attention.py:331-337)
#pragma unroll
        for (int t = 0; t < kTiles; ++t) x[0][t] += ld4(ca + 16 * t + 4 * g);
        layer_norm_rows<true>(x[0], pv + PV_LN2_W, pv + PV_LN2_B, g);
        rotate_tiles<NT>(x);
    }
    }
    FSTAMP(10);   // norm1, cross-attention constant, norm2
    // ---------------- FFN (cross_attention.py:338-340): x = norm3(x + linear2(gelu(linear1(x)))), 16 chunks of 32 hidden
    // features.  Stages: [linear1(0) | pad], 15 x [linear1(i + 1) | linear2(i)], [linear2(15) | pad]
    pack_rows<NT>(xb, x);
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int t = 0; t < kTiles; ++t) x[j][t] += ld4(pv + PV_L2_B + 16 * t + 4 * g);
    f32x4 hid[NT][2];
    {
        stage_fetch(sg);
        const f32x4 b0 = ld4(pv + PV_L1_B + 4 * g), b1 = ld4(pv + PV_L1_B + 16 + 4 * g);
#pragma unroll
        for (int j = 0; j < NT; ++j) { hid[j][0] = b0; hid[j][1] = b1; }
        gemm5<NT, 2, 4, 0>(hid, xb, sg);
        stage_end(sg);
        FSTAMP(11);   // FFN prologue stage
    }
#pragma unroll 1
    for (int c = 0; c < 15; ++c) {
        stage_fetch(sg);
        f32x4 nxt[NT][2];
        const f32x4 b0 = ld4(pv + PV_L1_B + 32 * (c + 1) + 4 * g), b1 = ld4(pv + PV_L1_B + 32 * (c + 1) + 16 + 4 * g);
#pragma unroll
        for (int j = 0; j < NT; ++j) { nxt[j][0] = b0; nxt[j][1] = b1; }
        OPV hb[NT][1];
        gemm5<NT, 2, 4, 0>(nxt, xb, sg);   // linear1 of the next chunk: MFMAs that do not depend on ...
#pragma unroll
        for (int j = 0; j < NT; ++j) hb[j][0] = OP_PACK(OP_GELU(hid[j][0]), OP_GELU(hid[j][1]));   // ... this VALU
        gemm5<NT, kTiles, 1, 8>(x, hb, sg);
#pragma unroll
        for (int j = 0; j < NT; ++j) { hid[j][0] = nxt[j][0]; hid[j][1] = nxt[j][1]; }
        FSTAMP(12);   // FFN stage compute
        stage_end(sg);
        FSTAMP(13);   // FFN stage barrier
    }
    {
        stage_fetch(sg);
        OPV hb[NT][1];
#pragma unroll
        for (int j = 0; j < NT; ++j) hb[j][0] = OP_PACK(OP_GELU(hid[j][0]), OP_GELU(hid[j][1]));
        gemm5<NT, kTiles, 1, 0>(x, hb, sg);
        stage_end(sg);
    }
    FSTAMP(14);   // FFN epilogue stage
    constexpr int kLnW = ENCL ? PV_LN2_W : PV_LN3_W, kLnB = ENCL ? PV_LN2_B : PV_LN3_B;
#pragma unroll 1
    for (int j = 0; j < NT; ++j) {
        layer_norm_rows<true>(x[0], pv + kLnW, pv + kLnB, g);
        rotate_tiles<NT>(x);
    }
    FSTAMP(15);   // norm3
    if constexpr (MODE == 0) {   // xs.append(x): packed operands of the skip linear that pops them
        uint4* sk = skipbuf + (size_t)blk * (20 * 4 * 64);
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c)
            {
                sk[((tile0 + 4 * j) * 4 + c) * 64 + lane] = __builtin_bit_cast(uint4, OP_PACK(x[j][2 * c], x[j][2 * c + 1]));
            }
    }
    if constexpr (TAP) {
        if (tap_out && blockIdx.x == 0) store_tap<NT>(tap_out, blk, x, tile0, g, r);
    }
}

}  // namespace
}  // namespace amuse
