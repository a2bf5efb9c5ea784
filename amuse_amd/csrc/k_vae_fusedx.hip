// MotionPrior.decode in the PARITY arithmetic (AMUSE_PREC_F32X: split-fp16 operands, three v_mfma_f32_16x16x32_f16 per product, fp32 everything else) as ONE
// persistent workgroup per clip - the shape of k_vae_fused.hip (reference models/latent_diffusion/vae.py:216-278, cross_attention.py:66-125,297-345,
// infer_ldm.py:168-173) with the arithmetic of the staged fp32x pair k_vae_rows8x / k_vae_attn_x, which moves q, k, v, o and the residual stream through memory between
// its 17 launches (3.7 GB per 256-clip decode).
//
// What differs from the 16-bit kernel is the register budget: a row tile's fp32 residual stream (32 registers) AND its split operand image (32 more) for the three tiles
// of a wave do not leave room for accumulators in 256 registers.  So
//   * q, k, v of a head are projected from the residual registers with the operand split ON THE FLY per k-pair (8 transient registers per tile);
//   * a head's attention output goes to an L2-resident scratch (the staged path's attn_o array) and out_proj runs behind the head loop, re-reading it one k-pair per
//     weight stage - every GEMM then accumulates in exactly the order of the staged kernels (k-pair outer, hi.lo + lo.hi + hi.hi), so the result is bitwise theirs;
//   * for the FFN and the skip linear - where the operand image must survive the accumulation into the residual registers - the three-tile waves park the images of
//     two tiles in LDS (the K / V image region is free in those phases: 4 waves x 2 tiles x 8 KiB) and read them back as fragments; the third tile's stays in registers;
//   * the U-Net skip stack is the staged path's fp32 array, written and read back by the same lanes.
// LDS: K_h / V_h^T hi and lo fragment images 80 KiB | weight ring 3 x 16 KiB | block parameters 2 x 8 KiB | cross-attention constants 5 KiB = 149 KiB.
// The weight stream is in unit PAIRS (hi | lo, 2 KiB): a 16 KiB LDS stage = 8 pairs; the stage protocol is k_vae_fused.hip's (amuse_fused.hpp Stager).
#define AMUSE_OP_F16
#include "amuse_fused.hpp"

namespace amuse {
namespace {

constexpr int kXKvBytes = 4 * kKeyRows * 64;            // K hi | K lo | V^T hi | V^T lo: 20 KiB each
constexpr int kXOffKv = 0;
constexpr int kXOffW = kXOffKv + kXKvBytes;
constexpr int kXOffPv = kXOffW + kWBufs * kStageBytes;
constexpr int kXOffCa = kXOffPv + 2 * kPvSlot;
constexpr int kXLdsBytes = kXOffCa + kCaBytes;           // 152,576 B
static_assert(kXLdsBytes <= 160 * 1024, "LDS");
// erf of the FFN activation: the branch-free fit (amuse_dev.hpp erf_bf: max |error| 7.9e-8).  libm erff and Abramowitz-Stegun 7.1.26 on the hardware rcp / exp2 time the same
// (1.65 ms per 256-clip decode); without any erf the decode takes 1.54 instead of 1.76 ms (no hoist): 0.22 ms of exposed VALU time (profiles/r05_fusedx_decode.txt).
constexpr int kParkTile = 8192;                          // one parked operand image: [4 k-pairs][hi | lo][64 lanes x 16 B]

// one product of split operands, weights as the A operand: acc += Wl.xh + Wh.xl + Wh.xh (the term order of k_vae_rows8x)
__device__ __forceinline__ f32x4 mfma3(f16x8 wh, f16x8 wl, const F16Pair& x, f32x4 acc) {
    acc = mfma_f16(wl, x.hi, acc);
    acc = mfma_f16(wh, x.lo, acc);
    return mfma_f16(wh, x.hi, acc);
}
// the same product with the operands swapped: the accumulator holds the TRANSPOSED tile (rows along the registers) - V^T for the PV product
__device__ __forceinline__ f32x4 mfma3t(f16x8 wh, f16x8 wl, const F16Pair& x, f32x4 acc) {
    acc = mfma_f16(x.hi, wl, acc);
    acc = mfma_f16(x.lo, wh, acc);
    return mfma_f16(x.hi, wh, acc);
}
// the 8 unit pairs of the current LDS stage: f(i, hi, lo); pair i + 1 is read before pair i's MFMAs
// Who copies the weight stream: two pieces of every 16-piece stage per wave, as in the 16-bit kernels.  (Leaving the copies to the four TWO-tile waves - the three-tile waves
// are the critical path of every stage - is bitwise the same and flat: 1.67 ms per 256-clip decode either way, profiles/r05_fusedx_decode.txt.)
constexpr int kP = 2;   // pieces per stage and wave
template <int NT>
__device__ __forceinline__ void stage_fetch_x(Stager& s) {
    const unsigned d = __builtin_amdgcn_readfirstlane(s.dst0 + s.widx * kStageBytes);
#pragma unroll
    for (int i = 0; i < kP; ++i) glds16(s.src + i * 64, d + i * 1024);
    s.src += kStage * 64;
    s.widx = s.widx == kWBufs - 1 ? 0 : s.widx + 1;
}
template <int NT>
__device__ __forceinline__ void stage_end_x(Stager& s) {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(kP) : "memory");
    s.ridx = s.ridx == kWBufs - 1 ? 0 : s.ridx + 1;
}
template <int NT, bool FETCH = true, class F>
__device__ __forceinline__ void for_pairs(Stager& s, F&& f) {
    if constexpr (FETCH) stage_fetch_x<NT>(s);
    f16x8 h = wfrag(s, 0), l = wfrag(s, 1);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const f16x8 ch = h, cl = l;
        if (i + 1 < 8) { h = wfrag(s, 2 * i + 2); l = wfrag(s, 2 * i + 3); }
        f(i, ch, cl);
    }
}

// operand images of a wave's NT tiles for a phase that accumulates into the residual registers: tiles [0, PARK) in LDS, the rest in registers
template <int NT>
struct Images {
    static constexpr int PARK = NT == 3 ? 2 : 0;
    F16Pair reg[NT - PARK][4];
    char* lds;   // this wave's parking area + lane * 16
    __device__ __forceinline__ void build(const f32x4 (&x)[NT][kTiles]) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const F16Pair p = split_f16(x[j][2 * c], x[j][2 * c + 1]);
                if (j < PARK) {
                    *reinterpret_cast<uint4*>(lds + j * kParkTile + (2 * c) * 1024) = __builtin_bit_cast(uint4, p.hi);
                    *reinterpret_cast<uint4*>(lds + j * kParkTile + (2 * c + 1) * 1024) = __builtin_bit_cast(uint4, p.lo);
                } else {
                    reg[j - PARK][c] = p;
                }
            }
    }
    __device__ __forceinline__ F16Pair get(int j, int c) const {
        if (j < PARK) {
            F16Pair p;
            p.hi = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(lds + j * kParkTile + (2 * c) * 1024));
            p.lo = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(lds + j * kParkTile + (2 * c + 1) * 1024));
            return p;
        }
        return reg[j - PARK][c];
    }
};

// softmax(q K^T) V of ONE 16-query tile against the keys of the current head: K / V^T hi and lo fragment images in LDS (bank-conflict-free slot order, amuse_fused.hpp
// frag_slot), one pair of key tiles (32 keys) per trip, merged online in log2 units.  The loop is VALU-bound (the kernel's SIMDs spend 54 % of their time issuing VALU
// instructions, 32 % on the matrix pipe - profiles/r05_fusedx_pmc.txt), so it is built like the 16-bit kernels' `attend`: scores leave the MFMAs RELATIVE to the row's
// running maximum (C operand = -m_run), the maximum's butterfly and the rescale run only in trips that move it (wave-uniform ballot), the row sums ride the matrix pipe
// on a ones fragment (of the split P the PV product uses), v_max3 without NaN canonicalisation (-fno-honor-nans).
__device__ __forceinline__ void attend_x(const uint4* Kh, const uint4* Kl, const uint4* Vh, const uint4* Vl, const F16Pair& qs, f32x4 (&o)[2], int len, int g, int r) {
    const int fs = frag_slot(g, r);
    o[0] = o[1] = splat4(0.f);
    float m_run = 0.f;
    f32x4 os = splat4(0.f);
    const f16x8 ones = __builtin_bit_cast(f16x8, r == 0 ? uint4{OP_ONE2, OP_ONE2, OP_ONE2, OP_ONE2} : uint4{0u, 0u, 0u, 0u});
    auto pair = [&](int jp, auto first, auto masked) {
        constexpr bool FIRST = decltype(first)::value, MASKED = decltype(masked)::value;
        f16x8 kh[2], kl[2], vh[2], vl[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            kh[u] = __builtin_bit_cast(f16x8, Kh[(2 * jp + u) * 64 + fs]);
            kl[u] = __builtin_bit_cast(f16x8, Kl[(2 * jp + u) * 64 + fs]);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            vh[u] = __builtin_bit_cast(f16x8, Vh[(2 * jp + u) * 64 + fs]);
            vl[u] = __builtin_bit_cast(f16x8, Vl[(2 * jp + u) * 64 + fs]);
        }
        f32x4 st[2];
        const f32x4 c0 = splat4(-m_run);
#pragma unroll
        for (int u = 0; u < 2; ++u) {   // lane (g, i): S[i][32 jp + 16 u + 4 g + m] - m_run
            st[u] = mfma_f16(kl[u], qs.hi, c0);
            st[u] = mfma_f16(kh[u], qs.lo, st[u]);
            st[u] = mfma_f16(kh[u], qs.hi, st[u]);
        }
        if constexpr (MASKED) {
            int lim = len - 32 * jp - 4 * g;   // element (u, m) is valid iff 16 u + m < lim
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int m = 0; m < 4; ++m) st[u][m] = (16 * u + m < lim) ? st[u][m] : -INFINITY;
        }
        float mx = max3(max3(st[0][0], st[0][1], st[0][2]), max3(st[0][3], st[1][0], st[1][1]), fmaxf(st[1][2], st[1][3]));
        if (FIRST || __builtin_amdgcn_ballot_w64(mx > kAttnTau) != 0) {   // (wave-uniform) some row's maximum moves by more than 2^kAttnTau (lazy rescaling, amuse_dev.hpp)
            mx = allreduce_g_max(mx);
            const float d = FIRST ? mx : fmaxf(mx, 0.f);
            st[0] -= splat4(d);
            st[1] -= splat4(d);
            if constexpr (!FIRST) {
                const float alpha = __builtin_amdgcn_exp2f(-d);
                os *= alpha;
                o[0] *= alpha;
                o[1] *= alpha;
            }
            m_run += d;
        }
        f32x4 p[2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int m = 0; m < 4; ++m) p[u][m] = __builtin_amdgcn_exp2f(st[u][m]);
        const F16Pair pp = split_f16(p[0], p[1]);
#pragma unroll
        for (int td = 0; td < 2; ++td) {   // O^T[d][i] += sum_key V[key][d] P[i][key]
            o[td] = mfma_f16(vl[td], pp.hi, o[td]);
            o[td] = mfma_f16(vh[td], pp.lo, o[td]);
            o[td] = mfma_f16(vh[td], pp.hi, o[td]);
        }
        os = mfma_f16(ones, pp.lo, os);
        os = mfma_f16(ones, pp.hi, os);
    };
    const int full = min(len / 32, kPairs);   // pairs entirely below len (len >= 1: pair 0 holds a valid key)
    if (full > 0) pair(0, std::true_type{}, std::false_type{});
    else pair(0, std::true_type{}, std::true_type{});
#pragma unroll 1
    for (int jp = 1; jp < full; ++jp) pair(jp, std::false_type{}, std::false_type{});
#pragma unroll 1
    for (int jp = max(full, 1); jp < kPairs; ++jp) pair(jp, std::false_type{}, std::true_type{});
    const float l = allreduce_g_sum(os[0]);
    o[0] = o[0] / l;
    o[1] = o[1] / l;
}

// What a block half needs to know about its clip and launch comes from the kernel's argument block through these accessors (kept in the kernarg segment and re-read
// where needed: as a struct of values they would sit in registers for the whole network).  The decoder (S = 300 rows per clip, a compile-time constant) and one step of
// the pose-space Denoiser (S = 302..304) share the halves.
template <class A> struct Geo;
template <> struct Geo<VaeFusedXArgs> {
    static __device__ __forceinline__ int S(const VaeFusedXArgs&) { return kFrames; }
    static __device__ __forceinline__ float* tap(const VaeFusedXArgs& a) { return a.tap_out; }
    static __device__ __forceinline__ const float* c1(const VaeFusedXArgs& a) { return a.c1; }
    static __device__ __forceinline__ float* c1_out(const VaeFusedXArgs& a) { return a.c1_out; }
};
template <> struct Geo<DenFusedXArgs> {
    static __device__ __forceinline__ int S(const DenFusedXArgs& a) { return a.S; }
    static __device__ __forceinline__ float* tap(const DenFusedXArgs&) { return nullptr; }
    static __device__ __forceinline__ const float* c1(const DenFusedXArgs&) { return nullptr; }
    static __device__ __forceinline__ float* c1_out(const DenFusedXArgs&) { return nullptr; }
};

// The ATTENTION half of a block: (MODE 2: the skip linear in front of an output block,) q / k / v of each head from the residual registers, K / V images, attention, the
// outputs to (a.obuf + row0 * kD).  The residual registers are not changed (MODE 2: replaced by the skip linear's result).  MODE 0: input block, 1: middle block, 2: output block.
template <int NT, int MODE, class A>
__device__ __forceinline__ void attn_half_x(f32x4 (&x)[NT][kTiles], Stager& sg, const A& a, int b, int blk, int tile0, const float* pv, const float* pv_next_src, unsigned pv_next_dst,
                                            char* smem, int len, int wave, int lane) {
    const int g = lane >> 4, r = lane & 15;
    const int S = Geo<A>::S(a);
    const size_t nrows = (size_t)a.B * S, row0 = (size_t)b * S;
    [[maybe_unused]] const bool prof_on = blockIdx.x == 0 && threadIdx.x == 0 && (blk == 1 || blk == 6);
    FSTAMP(1);   // block start
    char* kv = smem + kXOffKv;
    Images<NT> im;
    im.lds = kv + (wave & 3) * (2 * kParkTile) + lane * 16;
    if constexpr (MODE == 2) {
        // x = linear_blocks[blk - 5](cat(x, xs.pop()))   (cross_attention.py:118-120): eight stages, the x half (k-pairs 0..3), then the popped skip
        im.build(x);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the parked images are wave-private: no barrier)
        const float* bias = a.pvec + PV_SKIP_B + (blk - 5) * kD;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int t = 0; t < kTiles; ++t) x[j][t] = ld4(bias + 16 * t + 4 * g);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            F16Pair xc[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) xc[j] = im.get(j, c);
            for_pairs<NT>(sg, [&](int o, f16x8 wh, f16x8 wl) {
#pragma unroll
                for (int j = 0; j < NT; ++j) x[j][o] = mfma3(wh, wl, xc[j], x[j][o]);
            });
            stage_end_x<NT>(sg);
        }
        const float* sk = a.skip + (size_t)(8 - blk) * nrows * kD;
#pragma unroll 1
        for (int c = 0; c < 4; ++c) {
            F16Pair xc[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int frame = 16 * (tile0 + 4 * j) + r;
                const float* src = sk + (row0 + min(frame, S - 1)) * kD + 32 * c + 4 * g;
                const bool ok = frame < S;
                xc[j] = split_f16(ok ? ld4(src) : splat4(0.f), ok ? ld4(src + 16) : splat4(0.f));
            }
            for_pairs<NT>(sg, [&](int o, f16x8 wh, f16x8 wl) {
#pragma unroll
                for (int j = 0; j < NT; ++j) x[j][o] = mfma3(wh, wl, xc[j], x[j][o]);
            });
            stage_end_x<NT>(sg);
        }
    }
    FSTAMP(2);   // skip linear done
    {
        uint4* Kh = reinterpret_cast<uint4*>(kv);
        uint4* Kl = Kh + kKeyRows * 4;
        uint4* Vh = Kl + kKeyRows * 4;
        uint4* Vl = Vh + kPairs * 2 * 16 * 4;
        float* obuf = (a.obuf + row0 * kD);
        constexpr float kScaling = 0.17677669529663687f, kLog2e = 1.44269504088896340736f;
#pragma unroll 1
        for (int h = 0; h < kHeads; ++h) {
            // ---- stage A (two LDS stages): k, v of this head -> hi / lo fragment images
            {
                f32x4 kk[NT][2], vv[NT][2];
                const f32x4 bk0 = ld4(pv + PV_IN_B + kD + 32 * h + 4 * g), bk1 = ld4(pv + PV_IN_B + kD + 32 * h + 16 + 4 * g);
                const float bv0 = pv[PV_IN_B + 2 * kD + 32 * h + r], bv1 = pv[PV_IN_B + 2 * kD + 32 * h + 16 + r];
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    kk[j][0] = bk0; kk[j][1] = bk1;
                    vv[j][0] = splat4(bv0); vv[j][1] = splat4(bv1);
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    F16Pair xc[NT];
                    for_pairs<NT>(sg, [&](int i, f16x8 wh, f16x8 wl) {   // per k-pair c: k tiles (2), v tiles (2)
                        const int c = 2 * s2 + (i >> 2), t = i & 3;
                        if (t == 0) {
#pragma unroll
                            for (int j = 0; j < NT; ++j) xc[j] = split_f16(x[j][2 * c], x[j][2 * c + 1]);
                        }
#pragma unroll
                        for (int j = 0; j < NT; ++j) {
                            if (t < 2) kk[j][t] = mfma3(wh, wl, xc[j], kk[j][t]);
                            else vv[j][t - 2] = mfma3t(wh, wl, xc[j], vv[j][t - 2]);
                        }
                    });
                    if (s2 == 0) stage_end_x<NT>(sg);
                }
                FSTAMP(3);   // k, v computed
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int tile = tile0 + 4 * j;
                    const bool ok = 16 * tile + r < S;
                    const F16Pair ks = split_f16(ok ? kk[j][0] : splat4(0.f), ok ? kk[j][1] : splat4(0.f));
                    Kh[tile * 64 + frag_slot(g, r)] = __builtin_bit_cast(uint4, ks.hi);
                    Kl[tile * 64 + frag_slot(g, r)] = __builtin_bit_cast(uint4, ks.lo);
                    // V^T: lane (g, d) holds V[16 tile + 4 g + m][16 td + d]; the image slot ((pair, td), d, g) takes keys {32 jp + 4 g + e} | {32 jp + 16 + 4 g + e}
#pragma unroll
                    for (int td = 0; td < 2; ++td) {
                        f32x4 v = vv[j][td];
#pragma unroll
                        for (int m = 0; m < 4; ++m) v[m] = (16 * tile + 4 * g + m < S) ? v[m] : 0.f;
                        const F16Pair vs = split_f16(v, v);   // (both halves the same four values: the low one is written)
                        const int slot = ((tile >> 1) * 2 + td) * 64 + frag_slot(g, r);
                        const uint4 hi4 = __builtin_bit_cast(uint4, vs.hi), lo4 = __builtin_bit_cast(uint4, vs.lo);
                        *reinterpret_cast<uint2*>(reinterpret_cast<char*>(Vh + slot) + (tile & 1) * 8) = uint2{hi4.x, hi4.y};
                        *reinterpret_cast<uint2*>(reinterpret_cast<char*>(Vl + slot) + (tile & 1) * 8) = uint2{lo4.x, lo4.y};
                    }
                }
                FSTAMP(4);   // images written
                stage_end_x<NT>(sg);
                FSTAMP(5);   // barrier of stage A passed
            }
            // ---- stage B (one LDS stage): q of this head; attention, a tile at a time; the output to the scratch
            if (h == 0 && pv_next_src) {   // next block's small parameters -> the other LDS slot
                const unsigned d = __builtin_amdgcn_readfirstlane(pv_next_dst + wave * 1024);
                glds16(reinterpret_cast<const uint4*>(pv_next_src) + wave * 64 + lane, d);
            }
            F16Pair qs[NT];
            {
                f32x4 q[NT][2];
                const f32x4 bq0 = ld4(pv + PV_IN_B + 32 * h + 4 * g), bq1 = ld4(pv + PV_IN_B + 32 * h + 16 + 4 * g);
#pragma unroll
                for (int j = 0; j < NT; ++j) { q[j][0] = bq0; q[j][1] = bq1; }
                F16Pair xc[NT];
                for_pairs<NT>(sg, [&](int i, f16x8 wh, f16x8 wl) {
                    const int c = i >> 1, o = i & 1;
                    if (o == 0) {
#pragma unroll
                        for (int j = 0; j < NT; ++j) xc[j] = split_f16(x[j][2 * c], x[j][2 * c + 1]);
                    }
#pragma unroll
                    for (int j = 0; j < NT; ++j) q[j][o] = mfma3(wh, wl, xc[j], q[j][o]);
                });
#pragma unroll
                for (int j = 0; j < NT; ++j) qs[j] = split_f16((q[j][0] * kScaling) * kLog2e, (q[j][1] * kScaling) * kLog2e);
            }
            FSTAMP(6);   // q
#pragma unroll 1
            for (int j = 0; j < NT; ++j) {   // runtime loop: the attention's only copy in the instruction stream; the tiles rotate through slot 0
                f32x4 o[2];
                attend_x(Kh, Kl, Vh, Vl, qs[0], o, len, g, r);
                const int frame = 16 * (tile0 + 4 * j) + r;
                if (frame < S) {
                    float* dst = obuf + (size_t)frame * kD + 32 * h + 4 * g;
                    st4(dst, o[0]);
                    st4(dst + 16, o[1]);
                }
                const F16Pair first = qs[0];
#pragma unroll
                for (int jj = 0; jj + 1 < NT; ++jj) qs[jj] = qs[jj + 1];
                qs[NT - 1] = first;
            }
            FSTAMP(7);   // attention of the wave's tiles, outputs stored
            stage_end_x<NT>(sg);
            FSTAMP(8);   // barrier of stage B passed
        }
    }
}

// The ROW half of a block: out_proj of the outputs in (a.obuf + row0 * kD) (brought back by counted LDS-DMA), norm1, (decoder layers: the cross-attention constant, norm2,) FFN, the last
// norm, (MODE 0: the skip push).  ENCL: a TransformerEncoderLayer (cross_attention.py:259-272: two norms) instead of the decoder layer with its one-token memory.
// hoist (decode, block 0 of a full-length clip): the result behind norm1 is the weight set's constant c1 - no out_proj, no norm1.
template <int NT, int MODE, bool ENCL, class A>
__device__ __forceinline__ void row_half_x(f32x4 (&x)[NT][kTiles], Stager& sg, const A& a, int blk, int tile0, int b, const float* pv, const float* cal, char* smem, int wave,
                                           int lane, const bool hoist = false) {
    const int g = lane >> 4, r = lane & 15;
    const int S = Geo<A>::S(a);
    const size_t nrows = (size_t)a.B * S, row0 = (size_t)b * S;
    [[maybe_unused]] const bool prof_on = blockIdx.x == 0 && threadIdx.x == 0 && (blk == 1 || blk == 6);
    char* kv = smem + kXOffKv;
    Images<NT> im;
    im.lds = kv + (wave & 3) * (2 * kParkTile) + lane * 16;
    if (hoist) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int frame = 16 * (tile0 + 4 * j) + r;
#pragma unroll
            for (int t = 0; t < kTiles; ++t) x[j][t] = frame < S ? ld4(Geo<A>::c1(a) + (size_t)frame * kD + 16 * t + 4 * g) : splat4(0.f);
        }
    } else {
        float* obuf = (a.obuf + row0 * kD);
        // ---- out_proj (four LDS stages, k-pair = head): the attention outputs come back from the scratch one k-pair ahead of their stage
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int t = 0; t < kTiles; ++t) x[j][t] += ld4(pv + PV_OUT_B + 16 * t + 4 * g);
        // Plain loads here would be waited for by every stage end (vmcnt retires in order and the stage protocol waits for all but the two newest operations): a full
        // L2 round trip per stage, 7 % of the block.  So the outputs come back by LDS-DMA into a wave-private double buffer in the (now dead) K / V image region - lane
        // (g, r) fetches the 16 bytes it would have loaded, one k-pair (head) = 2 NT pieces - issued BEHIND the stage's weight fetch and counted exactly.
        const unsigned obase = lds_addr(smem) + kXOffKv + (wave < 4 ? wave * 12288 : 49152 + (wave - 4) * 8192);
        const char* obuf_l = smem + kXOffKv + (wave < 4 ? wave * 12288 : 49152 + (wave - 4) * 8192) + lane * 16;
        const float* osrc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) osrc[j] = obuf + (size_t)min(16 * (tile0 + 4 * j) + r, S - 1) * kD + 4 * g;
        auto fetch_o = [&](int c) {
            const unsigned d = __builtin_amdgcn_readfirstlane(obase + (c & 1) * (NT * 2048));
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                glds16(reinterpret_cast<const uint4*>(osrc[j] + 32 * c), d + (2 * j) * 1024);
                glds16(reinterpret_cast<const uint4*>(osrc[j] + 32 * c + 16), d + (2 * j + 1) * 1024);
            }
        };
        fetch_o(0);
#pragma unroll 1
        for (int c = 0; c < 4; ++c) {
            stage_fetch_x<NT>(sg);
            if (c + 1 < 4) {
                fetch_o(c + 1);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kP + 2 * NT) : "memory");   // this k-pair's outputs are in (and every older weight stage)
            } else {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kP) : "memory");
            }
            F16Pair xc[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const char* src = obuf_l + (c & 1) * (NT * 2048) + (2 * j) * 1024;
                xc[j] = split_f16(*reinterpret_cast<const f32x4*>(src), *reinterpret_cast<const f32x4*>(src + 1024));
            }
            for_pairs<NT, false>(sg, [&](int o, f16x8 wh, f16x8 wl) {
#pragma unroll
                for (int j = 0; j < NT; ++j) x[j][o] = mfma3(wh, wl, xc[j], x[j][o]);
            });
            if (c + 1 < 4) {   // (stage_end with the next k-pair's pieces allowed in flight behind this stage's weight fetch)
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(kP + 2 * NT) : "memory");
                sg.ridx = sg.ridx == kWBufs - 1 ? 0 : sg.ridx + 1;
            } else {
                stage_end_x<NT>(sg);
            }
            FSTAMP(9);   // an out_proj stage
        }
    }
    // ---------------- norm1, the one-token cross-attention (a per-clip constant), norm2  (cross_attention.py:331-337)
    {
        const float* ca = cal + blk * kD;
#pragma unroll 1
        for (int j = 0; j < NT; ++j) {
            if (!hoist) layer_norm_rows<false>(x[0], pv + PV_LN1_W, pv + PV_LN1_B, g);
            if (Geo<A>::c1_out(a) && blk == 0 && b == 0) {   // what the hoist loads: block 0 behind norm1 (one-clip launch of the library)
                const int frame = 16 * (tile0 + 4 * j) + r;
                if (frame < S) {
#pragma unroll
                    for (int t = 0; t < kTiles; ++t) st4(Geo<A>::c1_out(a) + (size_t)frame * kD + 16 * t + 4 * g, x[0][t]);
                }
            }
            if constexpr (!ENCL) {
#pragma unroll
                for (int t = 0; t < kTiles; ++t) x[0][t] += ld4(ca + 16 * t + 4 * g);
                layer_norm_rows<false>(x[0], pv + PV_LN2_W, pv + PV_LN2_B, g);
            }
            rotate_tiles<NT>(x);
        }
    }
    FSTAMP(10);   // norm1, cross-attention constant, norm2
    // ---------------- FFN (cross_attention.py:338-340): x = norm3(x + linear2(gelu(linear1(x)))), 16 chunks of 32 hidden features, two LDS stages each
    im.build(x);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int t = 0; t < kTiles; ++t) x[j][t] += ld4(pv + PV_L2_B + 16 * t + 4 * g);
    // Stages: linear1(0), 15 x [linear1(ch + 1), linear2(ch)], linear2(15).  linear1 runs one chunk ahead, so the erf-GELU of chunk ch (VALU) and linear1's MFMAs of chunk
    // ch + 1 share a stage - and the two waves of a SIMD, in lock step since the last barrier, take them in OPPOSITE order (three-tile waves: MFMAs first; two-tile waves:
    // GELU first), so that one's matrix-pipe time covers the other's VALU time instead of adding to it.
    f32x4 hid[NT][2];
    auto lin1 = [&](f32x4 (&acc)[NT][2], int ch) {
        const f32x4 b0 = ld4(pv + PV_L1_B + 32 * ch + 4 * g), b1 = ld4(pv + PV_L1_B + 32 * ch + 16 + 4 * g);
#pragma unroll
        for (int j = 0; j < NT; ++j) { acc[j][0] = b0; acc[j][1] = b1; }
        F16Pair xc[NT];
        for_pairs<NT>(sg, [&](int i, f16x8 wh, f16x8 wl) {
            const int c = i >> 1, o = i & 1;
            if (o == 0) {
#pragma unroll
                for (int j = 0; j < NT; ++j) xc[j] = im.get(j, c);
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                acc[j][o] = mfma3(wh, wl, xc[j], acc[j][o]);
            }
        });
    };
    auto gelu_split = [&](F16Pair (&hs)[NT]) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int m = 0; m < 4; ++m) hid[j][i][m] = gelu_erf_bf(hid[j][i][m]);
            hs[j] = split_f16(hid[j][0], hid[j][1]);
        }
    };
    lin1(hid, 0);
    stage_end_x<NT>(sg);
#pragma unroll 1
    for (int ch = 0; ch < 16; ++ch) {
        F16Pair hs[NT];
        if (ch < 15) {
            f32x4 nxt[NT][2];
            FSTAMP(11);
            if constexpr (NT == 3) {
                lin1(nxt, ch + 1);
                __builtin_amdgcn_sched_barrier(0);
                gelu_split(hs);
            } else {
                gelu_split(hs);
                __builtin_amdgcn_sched_barrier(0);
                lin1(nxt, ch + 1);
            }
            FSTAMP(12);   // linear1 of the next chunk + GELU of this one
            stage_end_x<NT>(sg);
            FSTAMP(13);
#pragma unroll
            for (int j = 0; j < NT; ++j) { hid[j][0] = nxt[j][0]; hid[j][1] = nxt[j][1]; }
        } else {
            gelu_split(hs);
        }
        for_pairs<NT>(sg, [&](int o, f16x8 wh, f16x8 wl) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                x[j][o] = mfma3(wh, wl, hs[j], x[j][o]);
            }
        });
        FSTAMP(14);   // linear2 of the chunk
        stage_end_x<NT>(sg);
        FSTAMP(15);
    }
#pragma unroll 1
    for (int j = 0; j < NT; ++j) {
        layer_norm_rows<false>(x[0], pv + (ENCL ? PV_LN2_W : PV_LN3_W), pv + (ENCL ? PV_LN2_B : PV_LN3_B), g);   // (an encoder layer's norm2)
        rotate_tiles<NT>(x);
    }
    FSTAMP(16);   // norm3
    if constexpr (MODE == 0) {   // xs.append(x)
        float* sk = a.skip + (size_t)blk * nrows * kD;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int frame = 16 * (tile0 + 4 * j) + r;
            if (frame < S) {
#pragma unroll
                for (int t = 0; t < kTiles; ++t) st4(sk + (row0 + frame) * kD + 16 * t + 4 * g, x[j][t]);
            }
        }
    }
    if (Geo<A>::tap(a) && b == 0) store_tap<NT>(Geo<A>::tap(a), blk, x, tile0, g, r);   // (decode only)
}

template <int NT>
__device__ __forceinline__ void decode_tiles_x(const VaeFusedXArgs& a, char* smem, Stager& sg, int tile0, int b, int len, int wave, int lane) {
    const int g = lane >> 4, r = lane & 15;
    const float* pvl = reinterpret_cast<const float*>(smem + kXOffPv);
    const float* cal = reinterpret_cast<const float*>(smem + kXOffCa);
    const unsigned lds0 = lds_addr(smem);
    f32x4 x[NT][kTiles];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int frame = 16 * (tile0 + 4 * j) + r;
#pragma unroll
        for (int t = 0; t < kTiles; ++t) x[j][t] = frame < kFrames ? ld4(a.pe + (size_t)frame * kD + 16 * t + 4 * g) : splat4(0.f);
    }
    asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // parameters, constants and stage 0 are in
    const bool hoist = a.c1 != nullptr && len == kFrames;
#pragma unroll 1
    for (int blk = 0; blk < 4; ++blk) {
        const float* pv = pvl + (blk & 1) * (kPvSlot / 4);
        const bool h0 = blk == 0 && hoist;
        if (h0) {   // (the next block's small parameters: otherwise issued inside the head loop)
            const unsigned d = __builtin_amdgcn_readfirstlane(lds0 + kXOffPv + kPvSlot + wave * 1024);
            glds16(reinterpret_cast<const uint4*>(a.pvec + PV_BLOCK) + wave * 64 + lane, d);
        } else {
            attn_half_x<NT, 0>(x, sg, a, b, blk, tile0, pv, a.pvec + (blk + 1) * PV_BLOCK, lds0 + kXOffPv + ((blk + 1) & 1) * kPvSlot, smem, len, wave, lane);
        }
        row_half_x<NT, 0, false>(x, sg, a, blk, tile0, b, pv, cal, smem, wave, lane, h0);
    }
    attn_half_x<NT, 1>(x, sg, a, b, 4, tile0, pvl, a.pvec + 5 * PV_BLOCK, lds0 + kXOffPv + kPvSlot, smem, len, wave, lane);
    row_half_x<NT, 1, false>(x, sg, a, 4, tile0, b, pvl, cal, smem, wave, lane);
#pragma unroll 1
    for (int blk = 5; blk < kLayers; ++blk) {
        const float* pv = pvl + (blk & 1) * (kPvSlot / 4);
        attn_half_x<NT, 2>(x, sg, a, b, blk, tile0, pv, blk + 1 < kLayers ? a.pvec + (blk + 1) * PV_BLOCK : nullptr, lds0 + kXOffPv + ((blk + 1) & 1) * kPvSlot, smem, len, wave, lane);
        row_half_x<NT, 2, false>(x, sg, a, blk, tile0, b, pv, cal, smem, wave, lane);
    }
    // ---------------- decoder.norm -> final_layer (333 outputs in 24 tiles, four quarters of 6: three LDS stages each) -> rotation epilogue
#pragma unroll 1
    for (int j = 0; j < NT; ++j) {
        layer_norm_rows<false>(x[0], a.pvec + PV_FINAL_W, a.pvec + PV_FINAL_B, g);
        rotate_tiles<NT>(x);
    }
    if (a.tap_out && b == 0) store_tap<NT>(a.tap_out, 9, x, tile0, g, r);
    F16Pair xs[NT][4];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) xs[j][c] = split_f16(x[j][2 * c], x[j][2 * c + 1]);
    float* fst = reinterpret_cast<float*>(smem + kXOffKv) + wave * 16 * kQStride;   // one staging tile per wave (the K / V images are dead)
#pragma unroll 1
    for (int quarter = 0; quarter < 4; ++quarter) {
        f32x4 f[NT][6];
#pragma unroll
        for (int o = 0; o < 6; ++o) {
            const f32x4 bi = ld4(a.final_bias + 16 * (6 * quarter + o) + 4 * g);
#pragma unroll
            for (int j = 0; j < NT; ++j) f[j][o] = bi;
        }
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3) {
            for_pairs<NT>(sg, [&](int i, f16x8 wh, f16x8 wl) {
                const int lin = 8 * s3 + i, c = lin / 6, o = lin - 6 * c;
#pragma unroll
                for (int j = 0; j < NT; ++j) f[j][o] = mfma3(wh, wl, xs[j][c], f[j][o]);
            });
            stage_end_x<NT>(sg);
        }
        const int f0 = 96 * quarter, nfe = quarter == 3 ? kFeats - 288 : 96, njo = quarter == 3 ? kJoints - 48 : 16;
#pragma unroll 1
        for (int j = 0; j < NT; ++j) {   // one tile at a time through the wave's staging tile (wave-private: no barrier); the tiles rotate through slot 0
            const int tile = tile0 + 4 * j;
            const int frame = 16 * tile + r;
            const bool keep = frame < kFrames && frame < len;   // output[~mask.T] = 0 (vae.py:274)
            const int rows_here = min(16, kFrames - 16 * tile);   // <= 0 for the padding tile
            const size_t row0 = (size_t)b * kFrames + 16 * tile;
#pragma unroll
            for (int o = 0; o < 6; ++o) st4(fst + r * kQStride + 16 * o + 4 * g, keep ? f[0][o] : splat4(0.f));
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
#pragma unroll
            for (int o = 0; o < 6; ++o) {
                const f32x4 first = f[0][o];
#pragma unroll
                for (int jj = 0; jj + 1 < NT; ++jj) f[jj][o] = f[jj + 1][o];
                f[NT - 1][o] = first;
            }
        }
    }
}

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_vae_fusedx(VaeFusedXArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x;
    const int len = a.lengths ? a.lengths[b] : kFrames;
    const unsigned lds0 = lds_addr(smem);
    // block 0's parameters, the clip's cross-attention constants, then the first two weight stages
    glds16(reinterpret_cast<const uint4*>(a.pvec) + wave * 64 + lane, lds0 + kXOffPv + wave * 1024);
    if (wave < 5) glds16(reinterpret_cast<const uint4*>(a.ca + (size_t)b * kLayers * kD) + wave * 64 + lane, lds0 + kXOffCa + wave * 1024);
    Stager sg;
    // (a full-length clip with the block-0 constant at hand starts behind block 0's sixteen attention stages: decoder_block_x, hoist)
    const size_t skip_units = (a.c1 && len == kFrames) ? (size_t)16 * kStage : 0;
    const int piece0 = 2 * wave;
    sg.src = a.wstream + (skip_units + (size_t)piece0) * 64 + lane;
    sg.dst0 = lds0 + kXOffW + piece0 * 1024;
    sg.ring = smem + kXOffW + lane * 16;
    sg.widx = 0;
    sg.ridx = 0;
    if (wave < 4) { stage_fetch_x<3>(sg); stage_fetch_x<3>(sg); }
    else { stage_fetch_x<2>(sg); stage_fetch_x<2>(sg); }
    if (wave < 4) decode_tiles_x<3>(a, smem, sg, wave, b, len, wave, lane);
    else decode_tiles_x<2>(a, smem, sg, wave + 8, b, len, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the two surplus fetches must not outlive the workgroup's LDS
}

// ---- one step of the pose-space Denoiser (DenFusedXArgs, amuse_kernels.hpp): encoder blocks, S = 302..304 rows per clip, no key mask (denoiser.py:177-187)
// x_t rows are 1332 B apart, so a lane's four-float groups are only 4-byte aligned: unaligned dwordx4 loads (global memory takes them); the k-pair that reaches past
// feature 332 is read element-wise under the bound.
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
template <int NT>
__device__ __forceinline__ void load_xt_pair(f32x4 (&v)[NT][2], const float* xb, int tile0, int r, int g, int npre, int c) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int fi = 16 * (tile0 + 4 * j) + r - npre;
        const bool fv = fi >= 0 && fi < kFrames;
        const float* row = xb + (size_t)(fv ? fi : 0) * kFeats + 4 * g;
#pragma unroll
        for (int h = 0; h < 2; ++h) {   // features 32 c + 16 h + 4 g + m
            const int f = 32 * c + 16 * h;
            f32x4 t;
            if (f + 15 < kFeats) {
                t = *reinterpret_cast<const f32x4u*>(row + f);
            } else {
#pragma unroll
                for (int m = 0; m < 4; ++m) t[m] = (f + 4 * g + m < kFeats) ? row[f + m] : 0.f;
            }
            v[j][h] = fv ? t : splat4(0.f);
        }
    }
}

// ENCM: MotionPrior.encode instead of a Denoiser step (DenFusedXArgs: encode)
template <int NT, bool ENCM>
__device__ __forceinline__ void den_tiles_x(const DenFusedXArgs& a, char* smem, Stager& sg, int tile0, int b, int wave, int lane) {
    const int g = lane >> 4, r = lane & 15;
    const float* pvl = reinterpret_cast<const float*>(smem + kXOffPv);
    const unsigned lds0 = lds_addr(smem);
    const int S = a.S, npre = a.npre;
    const float* xin = a.x_in + (size_t)b * kFrames * kFeats;
    // ---------------- pose_embd (denoiser.py:178): eleven stages of one k-pair x eight output tiles, the operands a stage ahead
    f32x4 x[NT][kTiles];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int t = 0; t < kTiles; ++t) x[j][t] = ld4(a.emb_bias + 16 * t + 4 * g);
    f32x4 cur[NT][2];
    load_xt_pair<NT>(cur, xin, tile0, r, g, npre, 0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // block 0's parameters and weight stages 0, 1 are in
#pragma unroll 1
    for (int c = 0; c < 11; ++c) {
        F16Pair xc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) xc[j] = split_f16(cur[j][0], cur[j][1]);
        if (c + 1 < 11) load_xt_pair<NT>(cur, xin, tile0, r, g, npre, c + 1);
        for_pairs<NT>(sg, [&](int o, f16x8 wh, f16x8 wl) {
#pragma unroll
            for (int j = 0; j < NT; ++j) x[j][o] = mfma3(wh, wl, xc[j], x[j][o]);
        });
        stage_end_x<NT>(sg);
    }
    // xseq = cat(emb_latent, pose_embd(sample)) + query_pos (denoiser.py:180-181); the tokens arrive with their positions added
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int frame = 16 * (tile0 + 4 * j) + r;
        if constexpr (ENCM) {   // xseq = cat(global_motion_token, skel_embedding(features)) + query_pos_encoder.pe[:302]  (vae.py:171-188)
#pragma unroll
            for (int t = 0; t < kTiles; ++t) {
                const int c = 16 * t + 4 * g;
                const f32x4 e = frame < npre ? ld4(a.ttok + (frame & 1) * kD + c) : x[j][t];
                x[j][t] = frame < S ? e + ld4(a.pe + (size_t)frame * kD + c) : splat4(0.f);
            }
        } else {
            const float* pt = frame == 0 ? a.ttok + (size_t)b * a.ttok_stride : a.ctok + ((size_t)b * (npre - 1) + (frame < npre ? frame - 1 : 0)) * kD;
#pragma unroll
            for (int t = 0; t < kTiles; ++t) {
                const int c = 16 * t + 4 * g;
                x[j][t] = frame >= S ? splat4(0.f) : (frame < npre ? ld4(pt + c) : x[j][t] + ld4(a.pe + (size_t)frame * kD + c));
            }
        }
    }
    // key mask: the Denoiser passes none; encode masks the padded frames, its two distribution tokens are always visible (vae.py:176-181)
    const int klen = ENCM ? (a.lengths ? a.lengths[b] : kFrames) + npre : S;
#pragma unroll 1
    for (int blk = 0; blk < 4; ++blk) {
        const float* pv = pvl + (blk & 1) * (kPvSlot / 4);
        attn_half_x<NT, 0>(x, sg, a, b, blk, tile0, pv, a.pvec + (blk + 1) * PV_BLOCK, lds0 + kXOffPv + ((blk + 1) & 1) * kPvSlot, smem, klen, wave, lane);
        row_half_x<NT, 0, true>(x, sg, a, blk, tile0, b, pv, pvl, smem, wave, lane);
    }
    attn_half_x<NT, 1>(x, sg, a, b, 4, tile0, pvl, a.pvec + 5 * PV_BLOCK, lds0 + kXOffPv + kPvSlot, smem, klen, wave, lane);
    row_half_x<NT, 1, true>(x, sg, a, 4, tile0, b, pvl, pvl, smem, wave, lane);
#pragma unroll 1
    for (int blk = 5; blk < kLayers; ++blk) {
        const float* pv = pvl + (blk & 1) * (kPvSlot / 4);
        attn_half_x<NT, 2>(x, sg, a, b, blk, tile0, pv, blk + 1 < kLayers ? a.pvec + (blk + 1) * PV_BLOCK : nullptr, lds0 + kXOffPv + ((blk + 1) & 1) * kPvSlot, smem, klen, wave, lane);
        row_half_x<NT, 2, true>(x, sg, a, blk, tile0, b, pv, pvl, smem, wave, lane);
    }
    if constexpr (ENCM) {   // encoder.norm of the distribution rows (mu | logvar, vae.py:196-203): rows 0, 1 of the clip's tile 0
        if (tile0 == 0) {
            layer_norm_rows<false>(x[0], a.pvec + PV_FINAL_W, a.pvec + PV_FINAL_B, g);
            if (r < 2) {
#pragma unroll
                for (int t = 0; t < kTiles; ++t) st4(a.eps_out + ((size_t)b * 2 + r) * kD + 16 * t + 4 * g, x[0][t]);
            }
        }
        return;
    }
    // ---------------- encoder.norm -> pose_proj (333 outputs in 24 tiles, four quarters of 6: three LDS stages each) -> mask -> eps_hat / scheduler update (k_vae.hip's last
    // stage: the parity modes' arithmetic - exact division, no contraction), every element read and written by the wave that owns its row tile (x_out may alias x_in)
#pragma unroll 1
    for (int j = 0; j < NT; ++j) {
        layer_norm_rows<false>(x[0], a.pvec + PV_FINAL_W, a.pvec + PV_FINAL_B, g);
        rotate_tiles<NT>(x);
    }
    F16Pair xs[NT][4];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) xs[j][c] = split_f16(x[j][2 * c], x[j][2 * c + 1]);
    float* fst = reinterpret_cast<float*>(smem + kXOffKv) + wave * 16 * kQStride;   // one staging tile per wave (the K / V images are dead)
    const int len = (a.lengths ? a.lengths[b] : kFrames) + npre;   // (eps rows of masked frames are zeroed: denoiser.py:187)
    const float* cf = a.coef;
    const float sb = cf ? cf[0] : 0.f, sa = cf ? cf[1] : 1.f, c0 = cf ? cf[2] : 0.f, cxx = cf ? cf[3] : 0.f, ce = cf ? cf[4] : 0.f, sgm = cf ? cf[5] : 0.f, clipv = cf ? cf[6] : 0.f;
    const float* nz = a.step_noise ? a.step_noise + (size_t)b * kFrames * kFeats : nullptr;
    float* eo = a.eps_out ? a.eps_out + (size_t)b * kFrames * kFeats : nullptr;
    float* xo = (cf && a.x_out) ? a.x_out + (size_t)b * kFrames * kFeats : nullptr;
#pragma unroll 1
    for (int quarter = 0; quarter < 4; ++quarter) {
        f32x4 f[NT][6];
#pragma unroll
        for (int o = 0; o < 6; ++o) {
            const f32x4 bi = ld4(a.final_bias + 16 * (6 * quarter + o) + 4 * g);
#pragma unroll
            for (int j = 0; j < NT; ++j) f[j][o] = bi;
        }
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3) {
            for_pairs<NT>(sg, [&](int i, f16x8 wh, f16x8 wl) {
                const int lin = 8 * s3 + i, c = lin / 6, o = lin - 6 * c;
#pragma unroll
                for (int j = 0; j < NT; ++j) f[j][o] = mfma3(wh, wl, xs[j][c], f[j][o]);
            });
            stage_end_x<NT>(sg);
        }
        const int f0 = 96 * quarter, nfe = quarter == 3 ? kFeats - 288 : 96;
#pragma unroll 1
        for (int j = 0; j < NT; ++j) {   // one tile at a time through the wave's staging tile (wave-private: no barrier); the tiles rotate through slot 0
            const int tile = tile0 + 4 * j;
            const int frame = 16 * tile + r;
            const bool keep = frame < S && frame < len;          // sample[~mask.T] = 0 (denoiser.py:187)
            const int rows_here = min(16, S - 16 * tile);          // <= 0 for a tile beyond the sequence
#pragma unroll
            for (int o = 0; o < 6; ++o) st4(fst + r * kQStride + 16 * o + 4 * g, keep ? f[0][o] : splat4(0.f));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            for (int i = lane; i < rows_here * nfe; i += 64) {
                const int rr = i / nfe, c = i - rr * nfe, fr = 16 * tile + rr - npre;
                if (fr < 0) continue;                              // (a condition-token row: its output is dropped, denoiser.py:184)
                const float e = fst[rr * kQStride + c];
                const size_t el = (size_t)fr * kFeats + f0 + c;
                if (eo) eo[el] = e;
                if (xo) {
#pragma clang fp contract(off)
                    const float xl = xin[el];
                    const float num = __fsub_rn(xl, __fmul_rn(sb, e));
                    float x0 = __fdiv_rn(num, sa);
                    if (clipv > 0.f) x0 = fminf(fmaxf(x0, -clipv), clipv);
                    float nx = __fmul_rn(c0, x0);
                    if (cxx != 0.f) nx = __fadd_rn(nx, __fmul_rn(cxx, xl));
                    if (ce != 0.f) nx = __fadd_rn(nx, __fmul_rn(ce, e));
                    if (sgm != 0.f) {
                        const float z = nz ? nz[el] : counter_normal4(a.seed, a.clip0 + (uint64_t)b, (uint32_t)a.step, (uint32_t)(el >> 2), 1u)[el & 3];
                        nx = __fadd_rn(nx, __fmul_rn(sgm, z));
                    }
                    xo[el] = nx;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int o = 0; o < 6; ++o) {
                const f32x4 first = f[0][o];
#pragma unroll
                for (int jj = 0; jj + 1 < NT; ++jj) f[jj][o] = f[jj + 1][o];
                f[NT - 1][o] = first;
            }
        }
    }
}

template <bool ENCM>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_den_fusedx(DenFusedXArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x;
    const unsigned lds0 = lds_addr(smem);
    glds16(reinterpret_cast<const uint4*>(a.pvec) + wave * 64 + lane, lds0 + kXOffPv + wave * 1024);   // block 0's parameters
    Stager sg;
    const int piece0 = 2 * wave;
    sg.src = a.wstream + (size_t)piece0 * 64 + lane;
    sg.dst0 = lds0 + kXOffW + piece0 * 1024;
    sg.ring = smem + kXOffW + lane * 16;
    sg.widx = 0;
    sg.ridx = 0;
    if (wave < 4) { stage_fetch_x<3>(sg); stage_fetch_x<3>(sg); }
    else { stage_fetch_x<2>(sg); stage_fetch_x<2>(sg); }
    if (wave < 4) den_tiles_x<3, ENCM>(a, smem, sg, wave, b, wave, lane);
    else den_tiles_x<2, ENCM>(a, smem, sg, wave + 8, b, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the two surplus fetches must not outlive the workgroup's LDS
}

}  // namespace

hipError_t launch_vae_fusedx(const VaeFusedXArgs& a, hipStream_t stream) {
    static DeviceOnce once;
    int dev_;
    if (!once.done(&dev_)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_vae_fusedx), hipFuncAttributeMaxDynamicSharedMemorySize, kXLdsBytes);
        if (e != hipSuccess) return e;
        once.set(dev_);
    }
#if AMUSE_FPROF
    {
        int zero = 0;
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fprof_n), &zero, sizeof(int));
    }
#endif
    hipLaunchKernelGGL(k_vae_fusedx, dim3(a.B), dim3(512), kXLdsBytes, stream, a);
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

hipError_t launch_den_fusedx(const DenFusedXArgs& a, hipStream_t stream) {
    static DeviceOnce once;
    int dev_;
    if (!once.done(&dev_)) {
        for (const void* k : {reinterpret_cast<const void*>(&k_den_fusedx<false>), reinterpret_cast<const void*>(&k_den_fusedx<true>)}) {
            hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, kXLdsBytes);
            if (e != hipSuccess) return e;
        }
        once.set(dev_);
    }
    if (a.encode) hipLaunchKernelGGL(k_den_fusedx<true>, dim3(a.B), dim3(512), kXLdsBytes, stream, a);
    else hipLaunchKernelGGL(k_den_fusedx<false>, dim3(a.B), dim3(512), kXLdsBytes, stream, a);
    return hipGetLastError();
}

}  // namespace amuse
