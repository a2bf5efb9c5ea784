// bf16 throughput variant of the persistent sampling kernel (k_sampler.hip has the design notes and the fp32
// parity kernel): the same T-step loop of PretrainedLPDM_v1.diffusion_backward (reference
// models/latent_diffusion/infer_ldm.py:137-161; Denoiser.forward denoiser.py:135-204; encoder blocks
// cross_attention.py:41-64,259-272; diffusers scheduler step) with EIGHT wavefronts per workgroup instead of four.
//
// Why: a wave's global_load blocks at issue while the CU's 64 B/clk vector-memory path is busy, so in the 4-wave
// kernel the weight fetch (3.8 MB per step and CU) and the MFMA / LDS / barrier chain of the SAME waves add up
// instead of overlapping.  Here the two halves of a block belong to different waves, and each
// group fetches its weights while the OTHER group is on the critical path:
//
//   wave w8 = 4 s + h.
//   Group A (s = 0): head h end to end (in_proj, attention, out_proj split-K) and FFN quarters 0,1 of head h's
//     slice.  It never reduces: in both combines it publishes its partial and then only waits for the result -
//     that is where it fetches its weights (the FFN half during the out_proj combine, the next block's attention
//     weights during the linear2 combine).
//   Group B (s = 1): reduces + normalises BOTH combines (wave h: feature tiles 2h, 2h+1) and computes FFN quarters
//     2,3.  It has nothing to do while the A waves run attention - that is where it fetches its FFN half.
//   So no wave issues a weight load while it is on the critical path (bar the B waves' 8-unit skip-linear groups), and the
//   FFN's MFMA + GELU work is spread over two waves per SIMD.
//   U-Net skip linears x = Linear(cat(x, skip)): split over OUTPUT tiles - B wave h computes its own feature tiles
//     2h, 2h+1 from the x half of the weights plus u = W[:, 128:] . skip, which A wave h prepared in its waiting time of
//     the previous linear2 combine; one barrier, no partial sums.  The skip stack is kept as packed bf16 MFMA
//     operands (what the 4-wave kernel's cvt_pk produces from its fp32 copy - same bits).
//   The residual stream travels between waves as packed bf16 MFMA operands (4 x 1 KiB per tile - what every GEMM
//   consumes); only the reducer of a feature tile keeps it in fp32 (the B waves, two tiles each, in registers).
//   Final LayerNorm + scheduler update in the B waves on their own tiles; latent in LDS.
//
// LDS (163,328 B): combine matrix A8 [8 rows][8 tiles][64] f32x4 - fp32 partials off the diagonal, the published
// bf16 operands ON it (slots (c, c), c < 4), so the gather of one combine never aliases the partial writes of the
// next - | row statistics | bf16 skip stack | small parameters | static token rows | latent | double-buffered time
// token.
#include "amuse_dev.hpp"
#include "amuse_kernels.hpp"

// This file is compiled twice: as is (bf16 operands: k_sample8 / launch_sample8) and through k_sampler8h.hip with AMUSE_OP_F16
// defined (fp16 operands, AMUSE_PREC_F16: k_sample8h / launch_sample8h) - the same instruction stream on v_mfma_f32_16x16x32_f16
// with v_cvt_pk_f16_f32 and a GELU polynomial one degree higher; the stream holds fp16 weights.
#ifdef AMUSE_OP_F16
#define OPV f16x8
#define OP_PACK pack_f16
#define OP_MFMA mfma_f16
#define OP_PREC PREC_F16
#define OP_GELU gelu_poly4h
#define OP_KERNEL k_sample8h
#define OP_LAUNCH launch_sample8h
#else
#define OPV bf16x8
#define OP_PACK pack_bf16
#define OP_MFMA mfma_bf16
#define OP_PREC PREC_BF16
#define OP_GELU gelu_poly4
#define OP_KERNEL k_sample8
#define OP_LAUNCH launch_sample8
#endif

namespace amuse {

namespace {

constexpr int kR8 = kRing8;
constexpr int kA8Bytes = 8 * kTiles * 64 * 16;                 // 65,536
constexpr int kStat8Off = kA8Bytes;                            // [4 reducers][16 rows] float2 (1 KiB reserved)
constexpr int kSkip8Off = kStat8Off + 8 * 16 * 8;              // [4 levels][4 pairs][64] uint4
constexpr int kPv8Off = kSkip8Off + 4 * 4 * 64 * 16;
constexpr int kPv8Floats = kLayers * kEncPv + 4 * kD + 2 * kD;
constexpr int kTokRows8Off = kPv8Off + kPv8Floats * 4;         // [8 tiles][64] f32x4
constexpr int kLat8Off = kTokRows8Off + kTiles * 64 * 16;      // [8 tiles][64] f32x4: the latent (rows tok == 0)
constexpr int kTT8Off = kLat8Off + kTiles * 64 * 16;           // [2][128] float
static_assert(kTT8Off + 2 * kD * 4 == kSample8LdsBytes, "LDS layout");

using Ring = WRing<kR8>;

// Issue split of an A wave's 32 units inside the out_proj (C1) and linear2 (C2) combines: N0 before the first barrier
// (C2 only - there the A waves arrive early), N1 / N2 after the first / second barrier, the rest after the gather.
// Values from sweeps on MI355X (rounds 1-4, docs/history.md 4.1b); all splits of the same family land within 2 %.
constexpr int kC1N1 = 12, kC1N2 = 12;
constexpr int kC2N0 = 16, kC2N1 = 12;
// units a B wave issues during the A waves' attention phase; the rest of its 32 follow behind its first FFN MFMAs
constexpr int kBEarly = 20;
// VALU instructions scheduled behind each MFMA while one FFN quarter's GELU overlaps the other quarter's GEMM (ffn_half)
constexpr int kFfnValuPerMfma = 7;

__device__ __forceinline__ f32x4* a8_slot(char* lds, int row, int col, int lane) {
    return reinterpret_cast<f32x4*>(lds) + (row * kTiles + col) * 64 + lane;
}
// packed bf16 operand c (feature tiles 2c, 2c+1 of the residual stream) is published in diagonal slot (c, c)
__device__ __forceinline__ uint4* xb_slot(char* lds, int c, int lane) {
    return reinterpret_cast<uint4*>(a8_slot(lds, c, c, lane));
}
// acc[o] += W_o . x over the 128 features held as four packed operands; same unit order as gemm_ring (bf16)
template <int NO, bool SWAP, int PH>
__device__ __forceinline__ void gemm_xb(f32x4 (&acc)[NO], const OPV (&xb)[4], const Ring& rg) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int o = 0; o < NO; ++o) {
            const OPV wf = __builtin_bit_cast(OPV, rg.s[(PH + c * NO + o) % kR8]);
            acc[o] = SWAP ? OP_MFMA(xb[c], wf, acc[o]) : OP_MFMA(wf, xb[c], acc[o]);
        }
}

// ---- split-K combine, B waves reduce.  NP = 4: partials from the A waves only (out_proj); NP = 8: from all waves
// (linear2).  B wave h reduces feature tiles 2h, 2h+1: residual + bias, LayerNorm (row statistics of the four
// 32-feature slices merged with Chan's formula), publishes them on the diagonal of A8; every wave gathers.
// Row of A8 holding writer w's partial of tile t - never the diagonal, and the tile's reducer (wave 4 + t/2) keeps its
// own partial in registers:
__device__ __forceinline__ constexpr int part_row(int w, int t) {
    const int red = 4 + (t >> 1);
    const int k = w - (w > red ? 1 : 0);
    return k + (k >= t ? 1 : 0);
}
// LN = false (U-Net skip linear): tiles = sum + bias - no residual, no LayerNorm, and one barrier less.
template <int W, int NP, bool FAST, bool LN>
__device__ __forceinline__ void combine_red(f32x4 (&part)[kTiles], f32x4 (&xo)[2], OPV (&xb)[4], const float* bias,
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
        if constexpr (LN) {
            ga[i] = ld4(gamma + 16 * (T0 + i) + 4 * g);
            be[i] = ld4(beta + 16 * (T0 + i) + 4 * g);
        }
    }
    __syncthreads();
    f32x4 y[2];
    {
        f32x4 p[NP][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int w = 0; w < NP; ++w) {
                if (w != 4 + W) p[w][i] = *a8_slot(lds, part_row(w, T0 + i), T0 + i, lane);
            }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            f32x4 sum;
            if constexpr (NP == 8) {
                p[4 + W][i] = part[T0 + i];
                sum = ((p[0][i] + p[1][i]) + (p[2][i] + p[3][i])) + ((p[4][i] + p[5][i]) + (p[6][i] + p[7][i]));
            } else {
                sum = ((p[0][i] + p[1][i]) + p[2][i]) + p[3][i];
            }
            y[i] = LN ? xo[i] + (sum + bi[i]) : sum + bi[i];
        }
    }
    if constexpr (LN) {
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
    const float var = M2 * (1.0f / kD) + 1e-5f;
    const float rstd = FAST ? __builtin_amdgcn_rsqf(var) : 1.0f / sqrtf(var);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int m = 0; m < 4; ++m) y[i][m] = (y[i][m] - mean) * rstd * ga[i][m] + be[i][m];
    }
    xo[0] = y[0];
    xo[1] = y[1];
    // publish the two tiles as ONE packed bf16 operand (k-tile pair W of every following GEMM)
    xb[W] = OP_PACK(y[0], y[1]);
    *xb_slot(lds, W, lane) = __builtin_bit_cast(uint4, xb[W]);
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if (c != W) xb[c] = __builtin_bit_cast(OPV, *xb_slot(lds, c, lane));
    // keeps SimplifyCFG from sinking the four cases' register-array stores into one block behind a pointer PHI (which
    // pins the arrays in scratch): an immediate operand cannot be merged
    asm volatile("; combine_red case %0" ::"n"(W));
}
template <int NP, bool FAST, bool LN = true>
__device__ __forceinline__ void combine_reduce(f32x4 (&part)[kTiles], f32x4 (&xo)[2], OPV (&xb)[4],
                                               const float* bias, const float* gamma, const float* beta, char* lds,
                                               int h, int lane) {
    if (h == 0) combine_red<0, NP, FAST, LN>(part, xo, xb, bias, gamma, beta, lds, lane);
    else if (h == 1) combine_red<1, NP, FAST, LN>(part, xo, xb, bias, gamma, beta, lds, lane);
    else if (h == 2) combine_red<2, NP, FAST, LN>(part, xo, xb, bias, gamma, beta, lds, lane);
    else combine_red<3, NP, FAST, LN>(part, xo, xb, bias, gamma, beta, lds, lane);
}
// The A waves' side: publish the partial, then the barriers and the gather - with the issue of N1 + N2 + N3
// weight-stream units into ring slots IPH0.. in between.  These waves are off the critical path here, so their
// blocking global_load issue costs nothing as long as it fits the reducers' phases.
// N0 units go out BEFORE the first barrier: free where the A waves arrive early (the linear2 combine - their FFN half
// is the shorter one), on the critical path where they arrive last (the out_proj combine: N0 = 0).
template <int N0, int N1, int N2, int N3, int IPH0, bool LN = true>
__device__ __forceinline__ void combine_publish(const f32x4 (&part)[kTiles], OPV (&xb)[4], char* lds, int h,
                                                int lane, Ring& rg) {
#pragma unroll
    for (int t = 0; t < kTiles; ++t)
        *a8_slot(lds, h + (h >= t ? 1 : 0), t, lane) = part[t];  // part_row(h, t), h < 4
    ring_issue<N0, kR8, IPH0 % kR8>(rg);
    __syncthreads();
    ring_issue<N1, kR8, (IPH0 + N0) % kR8>(rg);
    if constexpr (LN) __syncthreads();   // (the reducers' row-statistics exchange)
    ring_issue<N2, kR8, (IPH0 + N0 + N1) % kR8>(rg);
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; ++c) xb[c] = __builtin_bit_cast(OPV, *xb_slot(lds, c, lane));
    ring_issue<N3, kR8, (IPH0 + N0 + N1 + N2) % kR8>(rg);
}

// Slot of the combine matrix through which A wave h hands the skip-input half of the next output block's skip linear
// (feature tile t = 2h + i) to B wave h: off the diagonal, written after the reducers have read their partials.
__device__ __forceinline__ f32x4* u_slot(char* lds, int t, int lane) { return a8_slot(lds, t == 6 ? 5 : 6, t, lane); }

// The A waves' side of the linear2 combine: as combine_publish, fetching the next block's group of 32 units - its
// first 8 units into slots 24..31, then q, k, v into slots 0..23.  In front of an ordinary block the leading 8 are out_proj.
// In front of an OUTPUT block (cross_attention.py:58-61: x = Linear(cat(x, skips.pop()))) they are this wave's units of
// the skip linear, and in its waiting time the wave computes u = W[:, 128:] . skip for feature tiles 2h, 2h+1 - the half
// of the skip linear that does not depend on the current block's result - and hands it to B wave h through u_slot
// (after the second barrier: the reducers are done with the partials).  out_proj then follows during the skip linear.
// One body for both cases, the extra work behind a branch that touches no ring slot: separate instantiations of the
// fetch make hipcc copy the ring registers at the merge - behind an s_waitcnt vmcnt that stalls the wave until its
// loads have landed, in front of the last barrier.
template <int N0, int N1, int N2>
__device__ __forceinline__ void combine_publish_c2(const f32x4 (&part)[kTiles], OPV (&xb)[4], char* lds, int h,
                                                   int lane, Ring& rg, bool skip_u, const uint4* skip_ops) {
    static_assert(N0 >= 8 && N0 + N1 + N2 == 32, "the leading 8 units go out before the first barrier");
#pragma unroll
    for (int t = 0; t < kTiles; ++t)
        *a8_slot(lds, h + (h >= t ? 1 : 0), t, lane) = part[t];
    ring_issue<N0, kR8, 24>(rg);
    __syncthreads();
    ring_issue<N1, kR8, (24 + N0) % kR8>(rg);
    __syncthreads();
    if (skip_u) {
        OPV sk[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) sk[p] = __builtin_bit_cast(OPV, skip_ops[p * 64 + lane]);
        f32x4 u[2] = {splat4(0.f), splat4(0.f)};
        gemm_xb<2, false, 24>(u, sk, rg);
        *u_slot(lds, 2 * h, lane) = u[0];
        *u_slot(lds, 2 * h + 1, lane) = u[1];
    }
    ring_issue<N2, kR8, (24 + N0 + N1) % kR8>(rg);
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; ++c) xb[c] = __builtin_bit_cast(OPV, *xb_slot(lds, c, lane));
}

__device__ __forceinline__ void attention_head8(const f32x4 (&q)[2], const f32x4 (&k)[2], const f32x4 (&v)[2],
                                                const bool (&kvalid)[4], f32x4 (&o)[2]) {
    // S^T[j][i] = sum_d K[j][d] Q[i][d]  ->  lane (g, i) holds S[i][4 g + m]   (k_sampler.hip attention_head)
    f32x4 st = OP_MFMA(OP_PACK(k[0], k[1]), OP_PACK(q[0], q[1]), splat4(0.f));
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
        o[td] = OP_MFMA(OP_PACK(v[td], splat4(0.f)), OP_PACK(p, splat4(0.f)), splat4(0.f));
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

// GELU (amuse_dev.hpp OP_GELU: the result is an MFMA operand, i.e. rounded to bf16 next) on one FFN quarter (two
// hidden tiles); linear1's bias is already in the accumulators (ffn_half)
__device__ __forceinline__ void gelu_pair(f32x4 (&hq)[2]) {
    hq[0] = OP_GELU(hq[0]);
    hq[1] = OP_GELU(hq[1]);
}

// this wave's two FFN quarters (Q0, Q0 + 1 of head h's slice): linear1 for 2 hidden tiles each -> bias + GELU ->
// linear2 split-K contribution of those 32 features, software-pipelined by one quarter.  Ring: F1a F1b F2a F2b in
// slots 0..31, nothing re-armed.
// LATE8: the last 8 units (F2b) are issued only now, behind the first GEMMs' MFMAs (B waves: their fetch window, the A
// waves' attention phase, is a little too short for all 32)
template <int Q0, bool LATE8>
__device__ __forceinline__ void ffn_half(f32x4 (&part)[kTiles], const OPV (&xb)[4], Ring& rg, const float* pv,
                                         int h, int g) {
    constexpr int P = OP_PREC;
    // accumulators start at linear1's bias (this lane's 4 features of each hidden tile)
    const float* b1 = pv + PV_L1_B + 16 * (kTiles * h + 2 * Q0) + 4 * g;
    f32x4 ha[2] = {ld4(b1), ld4(b1 + 16)}, hb[2] = {ld4(b1 + 32), ld4(b1 + 48)};
    gemm_xb<2, false, 0>(ha, xb, rg);
    if constexpr (LATE8) ring_issue<32 - kBEarly, kR8, kBEarly>(rg);
    __builtin_amdgcn_sched_barrier(0);
    // An in-order wave that issues its 8 MFMAs back to back waits out the matrix pipe (16 cycles each) before its
    // first VALU instruction; interleaved 1 : 6 the GELU of one quarter runs in the shadow of the other quarter's
    // MFMAs (the two are independent).
    gemm_xb<2, false, 8>(hb, xb, rg);
    gelu_pair(ha);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, kFfnValuPerMfma, 0);   // 7 VALU
    }
    __builtin_amdgcn_sched_barrier(0);
    gemm_ring<P, kTiles, 2, false, kR8, 16, false>(part, ha, rg);
    gelu_pair(hb);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, kFfnValuPerMfma, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    gemm_ring<P, kTiles, 2, false, kR8, 24, false>(part, hb, rg);
}

// One TransformerEncoderLayer.forward_post (cross_attention.py:259-272), A / B role split.  The two roles are
// separate instantiations (and the whole step loop is instantiated per role, OP_KERNEL below): sharing one body
// behind a runtime branch makes hipcc's register allocator spill hundreds of VGPRs at the merges.
// xb: the residual stream as four packed bf16 operands (every wave); xo: this B wave's two feature tiles in fp32.
template <bool ROLEA, bool PROF>
__device__ __forceinline__ void encoder_block8(OPV (&xb)[4], f32x4 (&xo)[2], Ring& rg, const float* pv,
                                               const bool (&kvalid)[4], char* lds, int h, int lane, bool next_has_skip,
                                               const uint4* skip_next, Prof8& pf) {
    constexpr int P = OP_PREC;
    const int g = lane >> 4, r = lane & 15;
    f32x4 part[kTiles];
    if constexpr (ROLEA) {
        // ---- ring on entry: in_proj q,k (slots 0..15), v (16..23), out_proj (24..31)
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
        gemm_xb<4, false, 0>(qk, xb, rg);
        gemm_xb<2, true, 16>(v, xb, rg);
        const float scaling = 0.17677669529663687f;  // sqrt(1/32): q * scaling (F.multi_head_attention_forward)
        f32x4 q[2] = {(qk[0] + b_qk[0]) * scaling, (qk[1] + b_qk[1]) * scaling};
        f32x4 k[2] = {qk[2] + b_qk[2], qk[3] + b_qk[3]};
        v[0] += splat4(b_v[0]);
        v[1] += splat4(b_v[1]);
        f32x4 o[2];
        attention_head8(q, k, v, kvalid, o);
#pragma unroll
        for (int t = 0; t < kTiles; ++t) part[t] = splat4(0.f);
        gemm_ring<P, kTiles, 2, false, kR8, 24, false>(part, o, rg);
        stamp8<PROF>(pf);  // 1: in_proj + attention + out_proj partial
        // ---- out_proj combine (B reduces): meanwhile fetch this wave's FFN half
        combine_publish<0, kC1N1, kC1N2, 32 - kC1N1 - kC1N2, 0>(part, xb, lds, h, lane, rg);
        stamp8<PROF>(pf);  // 2: combine 1
#pragma unroll
        for (int t = 0; t < kTiles; ++t) part[t] = splat4(0.f);
        ffn_half<0, false>(part, xb, rg, pv, h, g);
        stamp8<PROF>(pf);  // 3: FFN
        // ---- linear2 combine (B reduces): meanwhile fetch the next block's attention weights
        combine_publish_c2<kC2N0, kC2N1, 32 - kC2N0 - kC2N1>(part, xb, lds, h, lane, rg, next_has_skip, skip_next);
    } else {
        // ---- ring empty on entry: fetch this wave's FFN half while the A waves run attention
        ring_issue<kBEarly, kR8, 0>(rg);
        stamp8<PROF>(pf);
#pragma unroll
        for (int t = 0; t < kTiles; ++t) part[t] = splat4(0.f);
        combine_reduce<4, true>(part, xo, xb, pv + PV_OUT_B, pv + PV_LN1_W, pv + PV_LN1_B, lds, h, lane);
        stamp8<PROF>(pf);
        ffn_half<2, (kBEarly < 32)>(part, xb, rg, pv, h, g);
        stamp8<PROF>(pf);  // 3: FFN
        if (next_has_skip) ring_issue<8, kR8, 0>(rg);  // the x half of this wave's two output tiles of the next block's skip linear
        combine_reduce<8, true>(part, xo, xb, pv + PV_L2_B, pv + PV_LN2_W, pv + PV_LN2_B, lds, h, lane);
    }
    stamp8<PROF>(pf);  // 4: combine 2
}

// debugging taps: one feature tile of the [16 x 128] residual stream
__device__ __forceinline__ void store_tap_tile(float* tap, int slot, int t, const f32x4& v, int g, int r) {
    st4(tap + ((size_t)slot * 16 + r) * kD + 16 * t + 4 * g, v);
}

// per-lane constants of the tile (row-lane layout: lane (g, r) holds row r)
struct Lane8 {
    int lane, g, r, cl, tok;
    long clip;
    bool valid, is_lat;
};
// Cheap to derive, expensive to keep: held across the block loop these constants get spilled to scratch (the ring
// leaves no slack), so the step loop re-derives them where it needs them from an opaque copy of the lane id.
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
__device__ __forceinline__ void role_loop8(const SampleArgs& a, char* smem, int w8, const Lane8& L0) {
    uint4* skipbf = reinterpret_cast<uint4*>(smem + kSkip8Off);
    const float* pvl = reinterpret_cast<const float*>(smem + kPv8Off);
    const f32x4* tokrows = reinterpret_cast<const f32x4*>(smem + kTokRows8Off);
    f32x4* latl = reinterpret_cast<f32x4*>(smem + kLat8Off);
    float* ttl = reinterpret_cast<float*>(smem + kTT8Off);
    float2* stats = reinterpret_cast<float2*>(smem + kStat8Off);
    const float* pv_skip = pvl + kLayers * kEncPv;
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
    // A waves enter every block with their 32 units in the ring (the stream's tail repeats its head for the wrap
    // at a step boundary); B waves enter with an empty ring (their fill is never consumed)
    Ring rg;
    if constexpr (ROLEA) {   // unit i of a block's group of 32 lives in slot (24 + i) % 32 (combine_publish_c2)
#pragma unroll
        for (int i = 0; i < kR8; ++i) rg.s[(24 + i) % kR8] = ldw(wbase + i * 64);
        rg.next = wbase + kR8 * 64;
    } else {
        ring_fill(rg, wbase);
    }
    Prof8 pf{a.prof_out ? a.prof_out + (size_t)w8 * 96 : nullptr, 0, false};
#pragma unroll 1
    for (int step = 0; step < a.T; ++step) {
        // ---- token assembly (denoiser.py:174,180-181): every wave builds the four packed operands, a B wave also
        // its own two tiles in fp32
        OPV xb[4];
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
            for (int c = 0; c < 4; ++c) xb[c] = OP_PACK(assemble(2 * c), assemble(2 * c + 1));
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
                // x = Linear(cat(x, skips.pop())) (cross_attention.py:58-61), split over OUTPUT tiles: B wave h computes
                // its own feature tiles 2h, 2h+1 = bias + u + W[:, :128] . x, where u = W[:, 128:] . skip was prepared by
                // A wave h during the previous linear2 combine (combine_publish_skipu).  It keeps them as its fp32 tiles
                // and publishes the packed operand: one barrier and no partial sums (the split-K version of this
                // 256 -> 128 linear cost 1.8 k cycles, mostly the 8-way reduction).  The A waves fetch out_proj.
                // Published in row 7 of the combine matrix: not the diagonal, which lagging waves may still be
                // gathering from the linear2 combine, and not a row the next out_proj combine's partials use.
                if constexpr (ROLEA) {
                    ring_issue<8, kR8, 24>(rg);
                } else {
                    const float* bs = pv_skip + (blk - 5) * kD + 32 * h + 4 * g;
                    f32x4 acc[2] = {ld4(bs) + *u_slot(smem, 2 * h, lane), ld4(bs + 16) + *u_slot(smem, 2 * h + 1, lane)};
                    gemm_xb<2, false, 0>(acc, xb, rg);
                    xo[0] = acc[0];
                    xo[1] = acc[1];
                    *reinterpret_cast<uint4*>(a8_slot(smem, 7, h, lane)) = __builtin_bit_cast(uint4, OP_PACK(acc[0], acc[1]));
                }
                __syncthreads();
#pragma unroll
                for (int c = 0; c < 4; ++c) xb[c] = __builtin_bit_cast(OPV, *reinterpret_cast<const uint4*>(a8_slot(smem, 7, c, lane)));
            }
            stamp8<PROF>(pf);  // 0: block start (after the skip linear, if any)
            encoder_block8<ROLEA, PROF>(xb, xo, rg, pvl + blk * kEncPv, kvalid, smem, h, lane, blk >= 4 && blk < kLayers - 1,
                                        skipbf + (7 - blk) * 4 * 64, pf);
            if (!ROLEA && blk < 4 && w8 == 4) {
#pragma unroll
                for (int p = 0; p < 4; ++p) skipbf[(blk * 4 + p) * 64 + lane] = __builtin_bit_cast(uint4, xb[p]);
            }
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
            const float rstd = __builtin_amdgcn_rsqf(M2 * (1.0f / kD) + 1e-5f);
            const float* cf = a.coef + (size_t)step * 8;
            const float sb = cf[0], sa = cf[1], c0 = cf[2], cx = cf[3], ce = cf[4], sg = cf[5], clipv = cf[6];
            const float inv_sa = 1.0f / sa;
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
                        float x0 = num * inv_sa;
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
__global__ __launch_bounds__(512) void OP_KERNEL(SampleArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* pvl = reinterpret_cast<float*>(smem + kPv8Off);
    f32x4* tokrows = reinterpret_cast<f32x4*>(smem + kTokRows8Off);
    f32x4* latl = reinterpret_cast<f32x4*>(smem + kLat8Off);
    float* ttl = reinterpret_cast<float*>(smem + kTT8Off);
    for (int i = threadIdx.x; i < kLayers * kEncPv / 4; i += 512) {
        const int blk = (4 * i) / kEncPv, off = 4 * i - blk * kEncPv;
        st4(pvl + 4 * i, ld4(a.pvec + blk * PV_BLOCK + off));
    }
    for (int i = threadIdx.x; i < (4 * kD + 2 * kD) / 4; i += 512) st4(pvl + kLayers * kEncPv + 4 * i, ld4(a.pvec + PV_SKIP_B + 4 * i));
    const int w8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int S = a.S;
    const Lane8 L = lane_info(a, threadIdx.x & 63);
    // static token rows (pe[0] under the latent rows, condition tokens; denoiser.py:174,180-181) and the initial
    // latent -> LDS (registers are the scarce resource at 2 waves / SIMD; wave 0 owns the latent's update)
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
    if (w8 < 4) role_loop8<true, PROF>(a, smem, w8, L);
    else role_loop8<false, PROF>(a, smem, w8, L);
    if (L.is_lat && w8 == 0 && a.latents_out) {
#pragma unroll
        for (int t = 0; t < kTiles; ++t) st4(a.latents_out + (size_t)L.clip * kD + 16 * t + 4 * L.g, latl[t * 64 + L.lane]);
    }
}

}  // namespace

hipError_t OP_LAUNCH(const SampleArgs& a, hipStream_t stream) {
    const int tiles = (a.B + a.G - 1) / a.G;
    static DeviceOnce once;
    int dev_;
    if (!once.done(&dev_)) {
        for (const void* k : {reinterpret_cast<const void*>(&OP_KERNEL<false>), reinterpret_cast<const void*>(&OP_KERNEL<true>)}) {
            hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, kSample8LdsBytes);
            if (e != hipSuccess) return e;
        }
        once.set(dev_);
    }
    if (a.prof_out) hipLaunchKernelGGL(OP_KERNEL<true>, dim3(tiles), dim3(512), kSample8LdsBytes, stream, a);
    else hipLaunchKernelGGL(OP_KERNEL<false>, dim3(tiles), dim3(512), kSample8LdsBytes, stream, a);
    return hipGetLastError();
}

}  // namespace amuse
