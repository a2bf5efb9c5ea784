// fp32x (AMUSE_PREC_F32X) variant of the 8-wave persistent sampling kernel: the T-step loop of
// PretrainedLPDM_v1.diffusion_backward (reference models/latent_diffusion/infer_ldm.py:137-161; Denoiser.forward
// denoiser.py:135-204; encoder blocks cross_attention.py:41-64,259-272; diffusers scheduler step) with the role split of
// k_sampler8.hip and the split-fp16 arithmetic of amuse_dev.hpp: every GEMM operand is hi + lo in fp16 (weights split on
// the host, activations in registers) and a product is three v_mfma_f32_16x16x32_f16 accumulated in fp32; softmax,
// LayerNorm, erf GELU and the scheduler update are fp32, as in the 4-wave parity kernels (k_sampler.hip).
//
// Why 8 waves here too: the mode moves 7.6 MB of weights per step through the CU's 64 B/clk load path (119 k cycles) and the
// 4-wave kernel, one wave per SIMD, pays that time ON TOP of its barrier / LDS / VALU chain - a wave blocked at load issue
// does nothing else (measured 4-wave fp32x: 96 us per step wherever the issue is placed; k_sampler.hip).  With two waves
// per SIMD in alternating roles the group that is off the critical path issues the loads:
//
//   wave w8 = 4 s + h.
//   Group A (s = 0): head h end to end (in_proj, attention, out_proj split-K) and FFN quarters 0,1 of head h's slice; in
//     both combines it publishes its partial and then waits - and fetches (its FFN half during the out_proj combine, the
//     next block's leading 32 units during the linear2 combine).
//   Group B (s = 1): reduces + normalises both combines (wave h: feature tiles 2h, 2h+1), FFN quarters 2,3; fetches its FFN
//     half while the A waves run attention.
//   A unit pair (hi, lo) takes two ring slots, so a 32-slot ring holds HALF of what a window of k_sampler8.hip fetches:
//   the second half of every group is re-armed at consumption, behind the MFMAs of the first (A: q,k k-pairs 2,3 behind
//   k-pairs 0,1 and out_proj behind v; B: linear2 behind linear1 - the A waves fetch their linear2 as one burst behind their
//   GELUs instead, leaving the path to the reducers first: ffn_half).
//   U-Net skip linears, residual stream exchange, final LayerNorm + scheduler update: as k_sampler8.hip, with the residual
//   stream travelling between waves as split operands (8 x 1 KiB: hi on the diagonal slots (c, c), lo on (4 + c, 4 + c)).
//   The skip stack as split operands (32 KiB) and all nine blocks' small parameters (61.5 KiB) do not fit the LDS together:
//   the parameters are streamed instead - the current and the next block's in two 7 KiB slots, the next one fetched by LDS-DMA
//   from the A waves' waiting time in the out_proj combine (combine_publish_c1) - and the skip stack stays in LDS.
//
// LDS (134,144 B): combine matrix A8 [8 rows][8 tiles][64] f32x4 | row statistics | small parameters (2 slots + tail) | skip
// stack | static token rows | latent | double-buffered time token.
#include "amuse_dev.hpp"
#include "amuse_kernels.hpp"

namespace amuse {

namespace {

constexpr int kR8 = kRing8;
constexpr int kA8Bytes = 8 * kTiles * 64 * 16;                 // 65,536
constexpr int kStat8Off = kA8Bytes;                            // [4 reducers][16 rows] float2 (1 KiB reserved)
// small parameters: the CURRENT and the NEXT block's (two slots of 7 KiB, filled by LDS-DMA a block ahead) + skip-linear biases +
// final LayerNorm.  (All nine blocks resident cost 61.5 KiB - the room the skip stack needs as split operands.)
constexpr int kPvSlotFloats = 7 * 256;                         // kEncPv = 1664 floats rounded up to whole 1 KiB DMA pieces
static_assert(kPvSlotFloats >= kEncPv && kPvSlotFloats <= PV_BLOCK, "a slot is filled by whole pieces read from the block's own region");
constexpr int kPv8Off = kStat8Off + 8 * 16 * 8;
constexpr int kPvTail8Off = kPv8Off + 2 * kPvSlotFloats * 4;   // [4][128] skip-linear biases | final LayerNorm weight, bias
constexpr int kSkip8Off = kPvTail8Off + (4 * kD + 2 * kD) * 4; // [4 levels][4 pairs][hi, lo][64] uint4: the U-Net skip stack, split operands
constexpr int kTokRows8Off = kSkip8Off + 4 * 8 * 64 * 16;      // [8 tiles][64] f32x4
constexpr int kLat8Off = kTokRows8Off + kTiles * 64 * 16;      // [8 tiles][64] f32x4: the latent (rows tok == 0)
constexpr int kTT8Off = kLat8Off + kTiles * 64 * 16;           // [2][128] float
static_assert(kTT8Off + 2 * kD * 4 == kSample8xLdsBytes, "LDS layout");

using Ring = WRing<kR8>;

// Issue split of an A wave's 32 units inside the out_proj (C1) and linear2 (C2) combines: N0 before the first barrier,
// N1 / N2 after the first / second barrier, the rest after the gather.
constexpr int kXC1N1 = 12, kXC1N2 = 12, kXC2N1 = 8;
// units a B wave issues during the A waves' attention phase (all 32 of its linear1)
constexpr int kXBEarly = 32;
// s_setprio of the B waves inside their FFN half (they are the reducers of the combine behind it): 73.4 -> 72.6 us per step
constexpr int kXBPrio = 3;
// The A waves fetch linear2's units as ONE burst behind both GELUs instead of re-arming at consumption (ffn_half, DEFER).  The FFN's
// GELU takes libm erff: same-box wall clock of this kernel, us per step: libm 71.4, the branch-free fit (amuse_dev.hpp erf_bf) 75.3,
// Abramowitz-Stegun on rcp / exp2 79.7 (docs/history.md; the 3 : 1 FFN split of tools/probes/fp32x_ffn_3_1_split ranked them the other way round).

__device__ __forceinline__ f32x4* a8_slot(char* lds, int row, int col, int lane) {
    return reinterpret_cast<f32x4*>(lds) + (row * kTiles + col) * 64 + lane;
}
// split operand c (feature tiles 2c, 2c+1 of the residual stream): hi in diagonal slot (c, c), lo in (4 + c, 4 + c) -
// partials never use the diagonal (part_row below)
__device__ __forceinline__ uint4* xs_hi_slot(char* lds, int c, int lane) { return reinterpret_cast<uint4*>(a8_slot(lds, c, c, lane)); }
__device__ __forceinline__ uint4* xs_lo_slot(char* lds, int c, int lane) { return reinterpret_cast<uint4*>(a8_slot(lds, 4 + c, 4 + c, lane)); }
__device__ __forceinline__ void publish_xs(char* lds, int c, int lane, const F16Pair& p) {
    *xs_hi_slot(lds, c, lane) = __builtin_bit_cast(uint4, p.hi);
    *xs_lo_slot(lds, c, lane) = __builtin_bit_cast(uint4, p.lo);
}
__device__ __forceinline__ F16Pair gather_xs(char* lds, int c, int lane) {
    F16Pair p;
    p.hi = __builtin_bit_cast(f16x8, *xs_hi_slot(lds, c, lane));
    p.lo = __builtin_bit_cast(f16x8, *xs_lo_slot(lds, c, lane));
    return p;
}

// acc[o] += W_o . x over NP k-pairs of split operands; unit pair (c, o) sits in ring slots PH + 2 (c NO + o), + 1
template <int NO, int NP, bool SWAP, int PH, bool REARM>
__device__ __forceinline__ void gemm_xs(f32x4 (&acc)[NO], const F16Pair* xs, Ring& rg) {
#pragma unroll
    for (int c = 0; c < NP; ++c) {
        f16x8 wh[NO], wl[NO];
#pragma unroll
        for (int o = 0; o < NO; ++o) {
            const int s0 = (PH + 2 * (c * NO + o)) % kR8, s1 = (PH + 2 * (c * NO + o) + 1) % kR8;
            wh[o] = __builtin_bit_cast(f16x8, rg.s[s0]);
            wl[o] = __builtin_bit_cast(f16x8, rg.s[s1]);
            if constexpr (REARM) {
                rg.s[s0] = ldw(rg.next);
                rg.s[s1] = ldw_pos(rg.next + 64, 1);   // (the lo unit of the pair)
                rg.next += 128;
            }
        }
#pragma unroll
        for (int o = 0; o < NO; ++o) acc[o] = SWAP ? mfma_f16(xs[c].hi, wl[o], acc[o]) : mfma_f16(wl[o], xs[c].hi, acc[o]);
#pragma unroll
        for (int o = 0; o < NO; ++o) acc[o] = SWAP ? mfma_f16(xs[c].lo, wh[o], acc[o]) : mfma_f16(wh[o], xs[c].lo, acc[o]);
#pragma unroll
        for (int o = 0; o < NO; ++o) acc[o] = SWAP ? mfma_f16(xs[c].hi, wh[o], acc[o]) : mfma_f16(wh[o], xs[c].hi, acc[o]);
    }
}

// ---- split-K combine, B waves reduce (k_sampler8.hip combine_red): NP = 4 partials from the A waves (out_proj) or 8
// from all waves (linear2).  Row of A8 holding writer w's partial of tile t - never the diagonal, and the tile's
// reducer (wave 4 + t/2) keeps its own partial in registers:
__device__ __forceinline__ constexpr int part_row(int w, int t) {
    const int red = 4 + (t >> 1);
    const int k = w - (w > red ? 1 : 0);
    return k + (k >= t ? 1 : 0);
}
template <int W, int NP>
__device__ __forceinline__ void combine_red(f32x4 (&part)[kTiles], f32x4 (&xo)[2], F16Pair (&xs)[4], const float* bias,
                                            const float* gamma, const float* beta, char* lds, int lane) {
    float2* stats = reinterpret_cast<float2*>(lds + kStat8Off);
    const int g = lane >> 4, r = lane & 15;
    constexpr int T0 = 2 * W;
    if constexpr (NP == 8) {
#pragma unroll
        for (int t = 0; t < kTiles; ++t)
            if (t != T0 && t != T0 + 1) *a8_slot(lds, part_row(4 + W, t), t, lane) = part[t];
    }
    f32x4 bi[2], ga[2], be[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        bi[i] = ld4(bias + 16 * (T0 + i) + 4 * g);
        ga[i] = ld4(gamma + 16 * (T0 + i) + 4 * g);
        be[i] = ld4(beta + 16 * (T0 + i) + 4 * g);
    }
    __syncthreads();
    f32x4 y[2];
    {
        f32x4 p[NP][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int w = 0; w < NP; ++w)
                if (w != 4 + W) p[w][i] = *a8_slot(lds, part_row(w, T0 + i), T0 + i, lane);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            f32x4 sum;
            if constexpr (NP == 8) {
                p[4 + W][i] = part[T0 + i];
                sum = ((p[0][i] + p[1][i]) + (p[2][i] + p[3][i])) + ((p[4][i] + p[5][i]) + (p[6][i] + p[7][i]));
            } else {
                sum = ((p[0][i] + p[1][i]) + p[2][i]) + p[3][i];
            }
            y[i] = xo[i] + (sum + bi[i]);
        }
    }
    float s = ((y[0][0] + y[0][1]) + (y[0][2] + y[0][3])) + ((y[1][0] + y[1][1]) + (y[1][2] + y[1][3]));
    s = allreduce_g_sum(s);
    const float mw = s * (1.0f / 32.0f);
    float m2 = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const float d = y[i][m] - mw;
            m2 += d * d;
        }
    m2 = allreduce_g_sum(m2);
    if (g == 0) stats[W * 16 + r] = float2{mw, m2};
    __syncthreads();
    const float2 s0 = stats[r], s1 = stats[16 + r], s2 = stats[32 + r], s3 = stats[48 + r];
    const float mean = ((s0.x + s1.x) + (s2.x + s3.x)) * 0.25f;
    const float d0 = s0.x - mean, d1 = s1.x - mean, d2 = s2.x - mean, d3 = s3.x - mean;
    const float M2 = ((s0.y + s1.y) + (s2.y + s3.y)) + 32.0f * ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
    const float rstd = 1.0f / sqrtf(M2 * (1.0f / kD) + 1e-5f);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int m = 0; m < 4; ++m) y[i][m] = (y[i][m] - mean) * rstd * ga[i][m] + be[i][m];
    xo[0] = y[0];
    xo[1] = y[1];
    // publish the two tiles as ONE split operand (k-tile pair W of every following GEMM)
    xs[W] = split_f16(y[0], y[1]);
    publish_xs(lds, W, lane, xs[W]);
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if (c != W) xs[c] = gather_xs(lds, c, lane);
    // keeps SimplifyCFG from sinking the four cases' register-array stores into one block behind a pointer PHI
    asm volatile("; combine_red case %0" ::"n"(W));
}
template <int NP>
__device__ __forceinline__ void combine_reduce(f32x4 (&part)[kTiles], f32x4 (&xo)[2], F16Pair (&xs)[4], const float* bias,
                                               const float* gamma, const float* beta, char* lds, int h, int lane) {
    if (h == 0) combine_red<0, NP>(part, xo, xs, bias, gamma, beta, lds, lane);
    else if (h == 1) combine_red<1, NP>(part, xo, xs, bias, gamma, beta, lds, lane);
    else if (h == 2) combine_red<2, NP>(part, xo, xs, bias, gamma, beta, lds, lane);
    else combine_red<3, NP>(part, xo, xs, bias, gamma, beta, lds, lane);
}

// Slot of the combine matrix through which A wave h hands the skip-input half of the next output block's skip linear
// (feature tile t = 2h + i) to B wave h: off the diagonal, written after the reducers have read their partials.
__device__ __forceinline__ f32x4* u_slot(char* lds, int t, int lane) { return a8_slot(lds, t == 6 ? 5 : 6, t, lane); }
// where B wave h publishes its two tiles of a skip linear's result: rows 7 (hi) and 5 (lo), columns 0..3 - not the
// diagonal (lagging waves may still be gathering the linear2 combine's operands), not u_slot's cells, not a row the next
// out_proj combine's partials use (rows 0..4)
__device__ __forceinline__ uint4* sk_hi_slot(char* lds, int c, int lane) { return reinterpret_cast<uint4*>(a8_slot(lds, 7, c, lane)); }
__device__ __forceinline__ uint4* sk_lo_slot(char* lds, int c, int lane) { return reinterpret_cast<uint4*>(a8_slot(lds, 5, c, lane)); }

// LDS-DMA: 64 lanes x 16 B from per-lane global addresses to LDS [dst, dst + 1 KiB), lane-linear (k_vae_fused.hip glds16).  Inline
// asm: hipcc does not count it in its s_waitcnt bookkeeping - its own waits can only become longer, never too short (vmcnt
// retires in order) - and does not drain it at barriers.  Completion before the data is read: see combine_publish_c1.
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

// The A waves' side of the out_proj combine: publish the partial, then the barriers and the gather - with the issue of
// this wave's first 32 FFN units (linear1 of both quarters) in between, into the whole (empty) ring.  Two more jobs ride here,
// where these waves only wait:
//  * in the block that follows a push (push = true) wave h stores pair h of the block's INPUT operands - the U-Net skip level -
//    into the LDS skip stack (every wave holds all four pairs; the pop, four or more blocks later, reads all eight units);
//  * the NEXT block's small parameters travel global -> LDS by DMA into the slot the previous block used (pieces 2h, 2h + 1 of 7).
//    They are older than every ring load this wave issues from here on, and the wave consumes those loads in its FFN half
//    before it reaches the linear2 combine - so the pieces have landed before that combine's barriers, which is when any wave
//    first reads the slot (vmcnt retires in order).
template <int N1, int N2>
__device__ __forceinline__ void combine_publish_c1(const f32x4 (&part)[kTiles], F16Pair (&xs)[4], char* lds, int h, int lane,
                                                   Ring& rg, bool push, uint4* skip_dst, const float* pv_next_src, unsigned pv_next_dst) {
#pragma unroll
    for (int t = 0; t < kTiles; ++t) *a8_slot(lds, h + (h >= t ? 1 : 0), t, lane) = part[t];  // part_row(h, t), h < 4
    if (push) {   // (a compile-time index per case: xs[h] with a runtime h would put the whole array in scratch)
        if (h == 0) { skip_dst[0 * 64] = __builtin_bit_cast(uint4, xs[0].hi); skip_dst[1 * 64] = __builtin_bit_cast(uint4, xs[0].lo); }
        else if (h == 1) { skip_dst[2 * 64] = __builtin_bit_cast(uint4, xs[1].hi); skip_dst[3 * 64] = __builtin_bit_cast(uint4, xs[1].lo); }
        else if (h == 2) { skip_dst[4 * 64] = __builtin_bit_cast(uint4, xs[2].hi); skip_dst[5 * 64] = __builtin_bit_cast(uint4, xs[2].lo); }
        else { skip_dst[6 * 64] = __builtin_bit_cast(uint4, xs[3].hi); skip_dst[7 * 64] = __builtin_bit_cast(uint4, xs[3].lo); }
    }
    __syncthreads();
    {
        const uint4* src = reinterpret_cast<const uint4*>(pv_next_src) + (2 * h) * 64 + lane;
        const unsigned dst = __builtin_amdgcn_readfirstlane(pv_next_dst + (2 * h) * 1024);
        glds16(src, dst);
        if (h < 3) glds16(src + 64, dst + 1024);
    }
    ring_issue<N1, kR8, 0>(rg);
    __syncthreads();   // (the reducers' row-statistics exchange)
    ring_issue<N2, kR8, N1>(rg);
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; ++c) xs[c] = gather_xs(lds, c, lane);
    ring_issue<32 - N1 - N2, kR8, N1 + N2>(rg);
}

// The A waves' side of the linear2 combine: fetch the next block's leading 32 units - 16 "lead" units into slots 0..15,
// then q, k for k-pairs 0,1 into slots 16..31.  In front of an ordinary block the lead is v.  In front of an OUTPUT
// block (cross_attention.py:58-61: x = Linear(cat(x, skips.pop()))) it is this wave's units of the skip linear's skip-input
// half: in its waiting time the wave computes u = W[:, 128:] . skip for feature tiles 2h, 2h+1 - the half of the skip linear
// that does not depend on the current block's result - from the popped level of the LDS skip stack, and hands it to B wave h through
// u_slot (after the second barrier: the reducers are done with the partials); v then follows during the skip linear.
// One body for both cases, the extra work behind a branch that touches no ring slot it does not own (k_sampler8.hip).
template <int N1>
__device__ __forceinline__ void combine_publish_c2(const f32x4 (&part)[kTiles], F16Pair (&xs)[4], char* lds, int h, int lane,
                                                   Ring& rg, bool skip_u, const uint4* skip_src) {
#pragma unroll
    for (int t = 0; t < kTiles; ++t) *a8_slot(lds, h + (h >= t ? 1 : 0), t, lane) = part[t];
    // the popped level (all eight split-operand units: the u GEMM runs over the full K), then the lead units
    F16Pair sk[4];
    if (skip_u) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            sk[c].hi = __builtin_bit_cast(f16x8, skip_src[(2 * c) * 64]);
            sk[c].lo = __builtin_bit_cast(f16x8, skip_src[(2 * c + 1) * 64]);
        }
    }
    ring_issue<16, kR8, 0>(rg);
    __syncthreads();
    ring_issue<N1, kR8, 16>(rg);
    __syncthreads();
    if (skip_u) {
        f32x4 u[2] = {splat4(0.f), splat4(0.f)};
        gemm_xs<2, 4, false, 0, false>(u, sk, rg);
        *u_slot(lds, 2 * h, lane) = u[0];
        *u_slot(lds, 2 * h + 1, lane) = u[1];
    }
    ring_issue<16 - N1, kR8, 16 + N1>(rg);
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; ++c) xs[c] = gather_xs(lds, c, lane);
}

__device__ __forceinline__ void attention_head8x(const f32x4 (&q)[2], const f32x4 (&k)[2], const f32x4 (&v)[2],
                                                 const bool (&kvalid)[4], f32x4 (&o)[2]) {
    // S^T[j][i] = sum_d K[j][d] Q[i][d]  ->  lane (g, i) holds S[i][4 g + m]   (k_sampler.hip attention_head, PREC_F16X2)
    const F16Pair ks = split_f16(k[0], k[1]), qs = split_f16(q[0], q[1]);
    f32x4 st = mfma_f16(ks.lo, qs.hi, splat4(0.f));
    st = mfma_f16(ks.hi, qs.lo, st);
    st = mfma_f16(ks.hi, qs.hi, st);
    float mx = -INFINITY;
#pragma unroll
    for (int m = 0; m < 4; ++m) mx = kvalid[m] ? fmaxf(mx, st[m]) : mx;
    mx = allreduce_g_max(mx);
    f32x4 p;
    float sum = 0.f;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const float e = expf(st[m] - mx);
        p[m] = kvalid[m] ? e : 0.f;
        sum += p[m];
    }
    sum = allreduce_g_sum(sum);
#pragma unroll
    for (int m = 0; m < 4; ++m) p[m] = p[m] / sum;
    const F16Pair ps = split_f16(p, splat4(0.f));
#pragma unroll
    for (int td = 0; td < 2; ++td) {
        const F16Pair vs = split_f16(v[td], splat4(0.f));
        o[td] = mfma_f16(vs.lo, ps.hi, splat4(0.f));
        o[td] = mfma_f16(vs.hi, ps.lo, o[td]);
        o[td] = mfma_f16(vs.hi, ps.hi, o[td]);
    }
}

// optional phase timeline: s_memtime stamps by lane 0 of every wave of workgroup 0 during ONE step ([8][96] u64)
struct Prof8 {
    unsigned long long* out;
    int idx;
    bool on;
};
template <bool PROF>
__device__ __forceinline__ void stamp8(Prof8& pf) {
    if constexpr (PROF) {
        if (pf.on) pf.out[pf.idx++] = __builtin_readcyclecounter();
    }
}

// exact-erf GELU on one FFN quarter (two hidden tiles); linear1's bias is already in the accumulators (ffn_half)
__device__ __forceinline__ void gelu_pair(f32x4 (&hq)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            hq[i][m] = gelu_erf(hq[i][m]);        // libm erff (one divergent branch per element)
        }
}

// this wave's two FFN quarters (Q0, Q0 + 1 of head h's slice): linear1 for 2 hidden tiles each -> bias + GELU -> linear2
// split-K contribution of those 32 features.  Ring on entry: F1a in slots 0..15, F1b in 16..31 (EARLY < 32: the B waves'
// last units are issued only now).  linear2's 32 units have to come through the load path inside this phase, for both
// groups (256 KiB per CU = 4.1 k cycles), and the B waves - the reducers of the combine that follows - are the critical ones:
//   DEFER = false (B): linear2's units are re-armed behind linear1's MFMAs, in the same slots;
//   DEFER = true (A): the wave leaves the path to the B waves first - both linear1 GEMMs and both GELUs without a load, then its
//     32 units as one burst.  (With both groups re-arming at consumption the A waves' loads went out first and the B waves sat
//     3 k cycles in their first GEMM: 9.5 k cycles per FFN phase, the A waves idle for 5 k of them in the combine behind it.)
template <int Q0, int EARLY, bool DEFER, bool PROF>
__device__ __forceinline__ void ffn_half(f32x4 (&part)[kTiles], const F16Pair (&xs)[4], Ring& rg, const float* pv, int h, int g, Prof8& pf) {
    const float* b1 = pv + PV_L1_B + 16 * (kTiles * h + 2 * Q0) + 4 * g;
    f32x4 ha[2] = {ld4(b1), ld4(b1 + 16)}, hb[2] = {ld4(b1 + 32), ld4(b1 + 48)};
    if constexpr (EARLY < 32) ring_issue<32 - EARLY, kR8, EARLY>(rg);
    gemm_xs<2, 4, false, 0, !DEFER>(ha, xs, rg);    // F1a; (B) slots 0..15 <- F2a
    stamp8<PROF>(pf);
    gemm_xs<2, 4, false, 16, !DEFER>(hb, xs, rg);   // F1b; (B) slots 16..31 <- F2b
    stamp8<PROF>(pf);
    gelu_pair(ha);
    stamp8<PROF>(pf);
    if constexpr (DEFER) {
        gelu_pair(hb);
        ring_issue<32, kR8, 0>(rg);
        stamp8<PROF>(pf);
        const F16Pair hsa = split_f16(ha[0], ha[1]), hsb = split_f16(hb[0], hb[1]);
        gemm_xs<kTiles, 1, false, 0, false>(part, &hsa, rg);
        stamp8<PROF>(pf);
        gemm_xs<kTiles, 1, false, 16, false>(part, &hsb, rg);
    } else {
        {
            const F16Pair hs = split_f16(ha[0], ha[1]);
            gemm_xs<kTiles, 1, false, 0, false>(part, &hs, rg);
        }
        stamp8<PROF>(pf);
        gelu_pair(hb);
        stamp8<PROF>(pf);
        const F16Pair hs = split_f16(hb[0], hb[1]);
        gemm_xs<kTiles, 1, false, 16, false>(part, &hs, rg);
    }
}

// One TransformerEncoderLayer.forward_post (cross_attention.py:259-272), A / B role split; the two roles are separate
// instantiations of the whole step loop (k_sampler8.hip).
// xs: the residual stream as four split operands (every wave); xo: this B wave's two feature tiles in fp32.
template <bool ROLEA, bool PROF>
__device__ __forceinline__ void encoder_block8x(F16Pair (&xs)[4], f32x4 (&xo)[2], Ring& rg, const float* pv,
                                                const bool (&kvalid)[4], char* lds, int h, int lane, bool push, uint4* skip_dst,
                                                bool next_has_skip, const uint4* skip_src, const float* pv_next_src, unsigned pv_next_dst,
                                                Prof8& pf) {
    const int g = lane >> 4, r = lane & 15;
    f32x4 part[kTiles];
    if constexpr (ROLEA) {
        // ---- ring on entry: lead = v (slots 0..15), q,k for k-pairs 0,1 (slots 16..31).  Two groups are re-armed at
        // consumption: q,k for k-pairs 2,3 behind k-pairs 0,1 and out_proj behind v - with v's MFMAs between the issue of the
        // former and its use (in the order q,k | q,k | v two L2 latencies were exposed one after the other: 4.4 k cycles)
        f32x4 b_qk[4];
        float b_v[2];
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            b_qk[o] = ld4(pv + PV_IN_B + 16 * (2 * h + o) + 4 * g);
            b_qk[2 + o] = ld4(pv + PV_IN_B + kD + 16 * (2 * h + o) + 4 * g);
            b_v[o] = pv[PV_IN_B + 2 * kD + 16 * (2 * h + o) + r];
        }
        f32x4 qk[4], v[2];
#pragma unroll
        for (int o = 0; o < 4; ++o) qk[o] = splat4(0.f);
        v[0] = v[1] = splat4(0.f);
        gemm_xs<4, 2, false, 16, true>(qk, xs, rg);        // k-pairs 0,1; slots 16..31 <- q,k for k-pairs 2,3
        gemm_xs<2, 4, true, 0, true>(v, xs, rg);           // v; slots 0..15 <- out_proj
        gemm_xs<4, 2, false, 16, false>(qk, xs + 2, rg);   // k-pairs 2,3
        const float scaling = 0.17677669529663687f;  // sqrt(1/32): q * scaling (F.multi_head_attention_forward)
        f32x4 q[2] = {(qk[0] + b_qk[0]) * scaling, (qk[1] + b_qk[1]) * scaling};
        f32x4 k[2] = {qk[2] + b_qk[2], qk[3] + b_qk[3]};
        v[0] += splat4(b_v[0]);
        v[1] += splat4(b_v[1]);
        f32x4 o[2];
        attention_head8x(q, k, v, kvalid, o);
#pragma unroll
        for (int t = 0; t < kTiles; ++t) part[t] = splat4(0.f);
        {
            const F16Pair os = split_f16(o[0], o[1]);
            gemm_xs<kTiles, 1, false, 0, false>(part, &os, rg);   // out_proj, k-slice of head h
        }
        stamp8<PROF>(pf);  // 1: in_proj + attention + out_proj partial
        // ---- out_proj combine (B reduces): meanwhile fetch linear1 of this wave's FFN half (and store a pushed skip level)
        combine_publish_c1<kXC1N1, kXC1N2>(part, xs, lds, h, lane, rg, push, skip_dst, pv_next_src, pv_next_dst);
        stamp8<PROF>(pf);  // 2: combine 1
#pragma unroll
        for (int t = 0; t < kTiles; ++t) part[t] = splat4(0.f);
        ffn_half<0, 32, true, PROF>(part, xs, rg, pv, h, g, pf);
        stamp8<PROF>(pf);  // 3: FFN
        // ---- linear2 combine (B reduces): meanwhile fetch the next block's leading units
        combine_publish_c2<kXC2N1>(part, xs, lds, h, lane, rg, next_has_skip, skip_src);
    } else {
        // ---- ring empty on entry: fetch linear1 of this wave's FFN half while the A waves run attention
        ring_issue<kXBEarly, kR8, 0>(rg);
        stamp8<PROF>(pf);
#pragma unroll
        for (int t = 0; t < kTiles; ++t) part[t] = splat4(0.f);
        combine_reduce<4>(part, xo, xs, pv + PV_OUT_B, pv + PV_LN1_W, pv + PV_LN1_B, lds, h, lane);
        stamp8<PROF>(pf);
        __builtin_amdgcn_s_setprio(kXBPrio);
        ffn_half<2, kXBEarly, false, PROF>(part, xs, rg, pv, h, g, pf);
        __builtin_amdgcn_s_setprio(0);
        stamp8<PROF>(pf);  // 3: FFN
        if (next_has_skip) ring_issue<16, kR8, 0>(rg);  // the x half of this wave's two output tiles of the next block's skip linear
        combine_reduce<8>(part, xo, xs, pv + PV_L2_B, pv + PV_LN2_W, pv + PV_LN2_B, lds, h, lane);
    }
    stamp8<PROF>(pf);  // 4: combine 2
}

// debugging taps: one feature tile of the [16 x 128] residual stream
__device__ __forceinline__ void store_tap_tile(float* tap, int slot, int t, const f32x4& v, int g, int r) {
    st4(tap + ((size_t)slot * 16 + r) * kD + 16 * t + 4 * g, v);
}

// per-lane constants of the tile (row-lane layout: lane (g, r) holds row r); re-derived where needed (k_sampler8.hip)
struct Lane8 {
    int lane, g, r, cl, tok;
    long clip;
    bool valid, is_lat;
};
__device__ __forceinline__ Lane8 lane_info(const SampleArgs& a, int lane) {
    asm volatile("" : "+v"(lane));
    Lane8 L;
    L.lane = lane;
    L.g = lane >> 4;
    L.r = lane & 15;
    const int S = a.S, R = S * a.G;
    L.cl = L.r / S;
    L.tok = L.r - L.cl * S;
    L.clip = (long)blockIdx.x * a.G + L.cl;
    L.valid = (L.r < R) && (L.clip < (long)a.B);
    L.is_lat = L.valid && L.tok == 0;
    return L;
}

// The whole T-step loop of one role.  Both roles execute the same sequence of workgroup barriers.
template <bool ROLEA, bool PROF>
__device__ __forceinline__ void role_loop8x(const SampleArgs& a, char* smem, int w8, const Lane8& L0) {
    const float* pvl = reinterpret_cast<const float*>(smem + kPv8Off);
    const f32x4* tokrows = reinterpret_cast<const f32x4*>(smem + kTokRows8Off);
    f32x4* latl = reinterpret_cast<f32x4*>(smem + kLat8Off);
    float* ttl = reinterpret_cast<float*>(smem + kTT8Off);
    float2* stats = reinterpret_cast<float2*>(smem + kStat8Off);
    const float* pv_skip = reinterpret_cast<const float*>(smem + kPvTail8Off);
    const float* pv_final = pv_skip + 4 * kD;
    const int lane = L0.lane, g = L0.g, r = L0.r, h = w8 & 3;
    const int S = a.S, R = S * a.G;
    // attention key mask for this lane's query row: keys j = 4 g + m of the SAME clip; padding rows attend to
    // themselves only (keeps them finite, they never touch valid rows)
    bool kvalid[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int j = 4 * g + m;
        kvalid[m] = L0.valid ? (j < R && (j / S) == L0.cl) : (j == r);
    }
    const bool tap = !ROLEA && a.tap_out != nullptr && blockIdx.x == 0;  // the B waves tap their own fp32 tiles
    const uint32_t wbase_units = ROLEA ? (uint32_t)w8 * (a.wave_units_a + kR8)
                                       : 4u * (a.wave_units_a + kR8) + (uint32_t)(w8 - 4) * a.wave_units_b;
    const uint4* wbase = a.wstream + (size_t)wbase_units * 64 + lane;
    uint4* skipl = reinterpret_cast<uint4*>(smem + kSkip8Off) + lane;   // [level][8 units][64 lanes]
    const unsigned pv_lds0 = lds_addr(smem + kPv8Off);
    int gblk = 0;   // blocks since the launch: block gblk's parameters sit in slot gblk & 1
    // A waves enter every block with their leading 32 units in the ring (the stream's tail repeats its head for the wrap
    // at a step boundary); B waves enter with an empty ring
    Ring rg;
    ring_fill(rg, wbase);   // (B: never consumed)
    Prof8 pf{a.prof_out ? a.prof_out + (size_t)w8 * 96 : nullptr, 0, false};
#pragma unroll 1
    for (int step = 0; step < a.T; ++step) {
        // ---- token assembly (denoiser.py:174,180-181): every wave builds the four split operands, a B wave also
        // its own two tiles in fp32
        F16Pair xs[4];
        f32x4 xo[2] = {splat4(0.f), splat4(0.f)};
        {
            const Lane8 L = lane_info(a, lane);
            auto assemble = [&](int t) -> f32x4 {
                const f32x4 sv = tokrows[t * 64 + lane];
                // unconditional loads (a divergent branch around them costs registers)
                f32x4 tt = ld4(ttl + (step & 1) * kD + 16 * t + 4 * g);
                if (a.time_tok_clip) tt = ld4(a.time_tok_clip + (size_t)(L.valid ? L.clip : 0) * kD + 16 * t + 4 * g);
                return !L.valid ? splat4(0.f) : (L.tok == 0 ? latl[t * 64 + lane] + sv : (L.tok == 1 ? tt : sv));
            };
#pragma unroll
            for (int c = 0; c < 4; ++c) xs[c] = split_f16(assemble(2 * c), assemble(2 * c + 1));
            if constexpr (!ROLEA) {
                xo[0] = assemble(2 * h);
                xo[1] = assemble(2 * h + 1);
            }
        }
        if (tap && step == 0) {
            store_tap_tile(a.tap_out, 0, 2 * h, xo[0], g, r);
            store_tap_tile(a.tap_out, 0, 2 * h + 1, xo[1], g, r);
        }
        // next step's time token -> the other LDS buffer (read a whole step and many barriers later)
        if (!ROLEA && w8 == 4 && lane < 32 && step + 1 < a.T)
            st4(ttl + ((step + 1) & 1) * kD + 4 * lane, ld4(a.time_tok + (size_t)(step + 1) * kD + 4 * lane));
        if constexpr (PROF) {
            pf.on = a.prof_out != nullptr && blockIdx.x == 0 && lane == 0 && step == a.prof_step;
            pf.idx = 0;
        }
        stamp8<PROF>(pf);  // step start
        rg.next = ROLEA ? wbase + kR8 * 64 : wbase;  // (A: the ring already holds units 0..31 of this step)
        // ---- SkipTransformerEncoder.forward (cross_attention.py:41-64)
#pragma unroll 1
        for (int blk = 0; blk < kLayers; ++blk) {
            if (blk >= 5) {
                // x = Linear(cat(x, skips.pop())) (cross_attention.py:58-61), split over OUTPUT tiles: B wave h computes its
                // own feature tiles 2h, 2h+1 = bias + u + W[:, :128] . x, where u = W[:, 128:] . skip was prepared by A wave h
                // during the previous linear2 combine (combine_publish_c2); it keeps them as its fp32 tiles and publishes
                // the split operand: one barrier and no partial sums.  The A waves fetch v into the lead slots.
                if constexpr (ROLEA) {
                    ring_issue<16, kR8, 0>(rg);
                } else {
                    const float* bs = pv_skip + (blk - 5) * kD + 32 * h + 4 * g;
                    f32x4 acc[2] = {ld4(bs) + *u_slot(smem, 2 * h, lane), ld4(bs + 16) + *u_slot(smem, 2 * h + 1, lane)};
                    gemm_xs<2, 4, false, 0, false>(acc, xs, rg);
                    xo[0] = acc[0];
                    xo[1] = acc[1];
                    const F16Pair pr = split_f16(acc[0], acc[1]);
                    *sk_hi_slot(smem, h, lane) = __builtin_bit_cast(uint4, pr.hi);
                    *sk_lo_slot(smem, h, lane) = __builtin_bit_cast(uint4, pr.lo);
                }
                __syncthreads();
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    xs[c].hi = __builtin_bit_cast(f16x8, *sk_hi_slot(smem, c, lane));
                    xs[c].lo = __builtin_bit_cast(f16x8, *sk_lo_slot(smem, c, lane));
                }
            }
            stamp8<PROF>(pf);  // 0: block start (after the skip linear, if any)
            // U-Net wiring: the inputs of blocks 1..4 are the outputs of input blocks 0..3 = skip levels 0..3; output block
            // blk (5..8) pops level 8 - blk, fetched during the linear2 combine of block blk - 1
            const int nblk = blk + 1 == kLayers ? 0 : blk + 1;   // (behind the last step's last block: a fetch nobody reads)
            encoder_block8x<ROLEA, PROF>(xs, xo, rg, pvl + (gblk & 1) * kPvSlotFloats, kvalid, smem, h, lane, blk >= 1 && blk <= 4,
                                         skipl + (blk - 1) * 8 * 64, blk >= 4 && blk < kLayers - 1, skipl + (7 - blk) * 8 * 64,
                                         a.pvec + nblk * PV_BLOCK, pv_lds0 + ((gblk + 1) & 1) * (kPvSlotFloats * 4), pf);
            ++gblk;
            if (tap && step == 0) {
                store_tap_tile(a.tap_out, 1 + blk, 2 * h, xo[0], g, r);
                store_tap_tile(a.tap_out, 1 + blk, 2 * h + 1, xo[1], g, r);
            }
        }
        // ---- final LayerNorm (SkipTransformerEncoder.norm) + scheduler.step (diffusers 0.17.1 DDIM / DDPM;
        // amuse_hip.h amuse_schedule) on the B waves' own tiles; the latent lives in LDS
        if constexpr (!ROLEA) {
            float sm = ((xo[0][0] + xo[0][1]) + (xo[0][2] + xo[0][3])) + ((xo[1][0] + xo[1][1]) + (xo[1][2] + xo[1][3]));
            sm = allreduce_g_sum(sm);
            const float mw = sm * (1.0f / 32.0f);
            float m2 = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const float d = xo[i][m] - mw;
                    m2 += d * d;
                }
            m2 = allreduce_g_sum(m2);
            if (g == 0) stats[h * 16 + r] = float2{mw, m2};
        }
        __syncthreads();
        if constexpr (!ROLEA) {
            const Lane8 L = lane_info(a, lane);
            const float2 s0 = stats[L.r], s1 = stats[16 + L.r], s2 = stats[32 + L.r], s3 = stats[48 + L.r];
            const float mean = ((s0.x + s1.x) + (s2.x + s3.x)) * 0.25f;
            const float d0 = s0.x - mean, d1 = s1.x - mean, d2 = s2.x - mean, d3 = s3.x - mean;
            const float M2 = ((s0.y + s1.y) + (s2.y + s3.y)) + 32.0f * ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
            const float rstd = 1.0f / sqrtf(M2 * (1.0f / kD) + 1e-5f);
            const float* cf = a.coef + (size_t)step * 8;
            const float sb = cf[0], sa = cf[1], c0 = cf[2], cx = cf[3], ce = cf[4], sg = cf[5], clipv = cf[6];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int t = 2 * h + i, f = 16 * t + 4 * L.g;
                const f32x4 ga = ld4(pv_final + f), be = ld4(pv_final + kD + f);
                f32x4 e;
#pragma unroll
                for (int m = 0; m < 4; ++m) e[m] = (xo[i][m] - mean) * rstd * ga[m] + be[m];
                if (tap && step == 0) store_tap_tile(a.tap_out, 10, t, e, L.g, L.r);
                if (a.eps_out && L.is_lat && step == a.T - 1) st4(a.eps_out + (size_t)L.clip * kD + f, e);
                if (!a.no_update) {
                    // ancestral noise of the latent rows: counter (global clip, step, feature group) - the values
                    // amuse_counter_normal exposes
                    f32x4 z = splat4(0.f);
                    if (sg != 0.f && L.is_lat)
                        z = a.step_noise ? ld4(a.step_noise + ((size_t)step * a.B + L.clip) * kD + f)
                                         : counter_normal4(a.seed, a.clip0 + (uint64_t)L.clip, (uint32_t)step, (uint32_t)(4 * t + L.g), 1u);
                    f32x4 l = latl[t * 64 + L.lane];
                    {
// each product and sum rounded on its own, like the scheduler's tensor ops (see k_sampler.hip)
#pragma clang fp contract(off)
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        const float xl = l[m];
                        const float num = __fsub_rn(xl, __fmul_rn(sb, e[m]));
                        float x0 = __fdiv_rn(num, sa);
                        if (clipv > 0.f) x0 = fminf(fmaxf(x0, -clipv), clipv);
                        float nx = __fmul_rn(c0, x0);
                        if (cx != 0.f) nx = __fadd_rn(nx, __fmul_rn(cx, xl));
                        if (ce != 0.f) nx = __fadd_rn(nx, __fmul_rn(ce, e[m]));
                        if (sg != 0.f) nx = __fadd_rn(nx, __fmul_rn(sg, z[m]));
                        l[m] = nx;
                    }
                    }
                    latl[t * 64 + L.lane] = l;
                    if (a.traj_out && L.is_lat) st4(a.traj_out + ((size_t)step * a.B + L.clip) * kD + f, l);
                }
            }
        }
        stamp8<PROF>(pf);  // scheduler update done (B waves) / reached the step barrier
        __syncthreads();  // the updated latent is visible to every wave's token assembly
    }
}

template <bool PROF>
__global__ __launch_bounds__(512) void k_sample8x(SampleArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* pvl = reinterpret_cast<float*>(smem + kPv8Off);
    f32x4* tokrows = reinterpret_cast<f32x4*>(smem + kTokRows8Off);
    f32x4* latl = reinterpret_cast<f32x4*>(smem + kLat8Off);
    float* ttl = reinterpret_cast<float*>(smem + kTT8Off);
    for (int i = threadIdx.x; i < kEncPv / 4; i += 512) st4(pvl + 4 * i, ld4(a.pvec + 4 * i));   // block 0 -> slot 0
    for (int i = threadIdx.x; i < (4 * kD + 2 * kD) / 4; i += 512)
        st4(reinterpret_cast<float*>(smem + kPvTail8Off) + 4 * i, ld4(a.pvec + PV_SKIP_B + 4 * i));
    const int w8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int S = a.S;
    const Lane8 L = lane_info(a, threadIdx.x & 63);
    // static token rows (pe[0] under the latent rows, condition tokens; denoiser.py:174,180-181) and the initial
    // latent -> LDS
    if (w8 == 0) {
#pragma unroll
        for (int t = 0; t < kTiles; ++t) {
            const int f = 16 * t + 4 * L.g;
            f32x4 sv = splat4(0.f), l0 = splat4(0.f);
            if (L.valid) {
                if (L.tok == 0) sv = ld4(a.pe0 + f);
                else if (L.tok >= 2) sv = ld4(a.cond_tok + ((size_t)L.clip * (S - 2) + (L.tok - 2)) * kD + f);
            }
            if (L.is_lat)
                l0 = a.x_init ? ld4(a.x_init + (size_t)L.clip * kD + f)
                              : counter_normal4(a.seed, a.clip0 + (uint64_t)L.clip, 0u, (uint32_t)(4 * t + L.g), 0u);
            tokrows[t * 64 + L.lane] = sv;
            latl[t * 64 + L.lane] = l0;
        }
    }
    if (w8 == 4 && L.lane < 32 && !a.time_tok_clip) st4(ttl + 4 * L.lane, ld4(a.time_tok + 4 * L.lane));
    __syncthreads();
    if (w8 < 4) role_loop8x<true, PROF>(a, smem, w8, L);
    else role_loop8x<false, PROF>(a, smem, w8, L);
    if (L.is_lat && w8 == 0 && a.latents_out) {
#pragma unroll
        for (int t = 0; t < kTiles; ++t) st4(a.latents_out + (size_t)L.clip * kD + 16 * t + 4 * L.g, latl[t * 64 + L.lane]);
    }
}

}  // namespace

hipError_t launch_sample8x(const SampleArgs& a, hipStream_t stream) {
    const int tiles = (a.B + a.G - 1) / a.G;
    static DeviceOnce once;
    int dev_;
    if (!once.done(&dev_)) {
        for (const void* k : {reinterpret_cast<const void*>(&k_sample8x<false>), reinterpret_cast<const void*>(&k_sample8x<true>)}) {
            hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, kSample8xLdsBytes);
            if (e != hipSuccess) return e;
        }
        once.set(dev_);
    }
    if (a.prof_out) hipLaunchKernelGGL(k_sample8x<true>, dim3(tiles), dim3(512), kSample8xLdsBytes, stream, a);
    else hipLaunchKernelGGL(k_sample8x<false>, dim3(tiles), dim3(512), kSample8xLdsBytes, stream, a);
    return hipGetLastError();
}

}  // namespace amuse
