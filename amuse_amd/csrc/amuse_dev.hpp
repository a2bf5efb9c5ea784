// Device-side building blocks shared by the sampling kernel and the VAE-decode kernels (gfx950 only).
//
// Data layout ("row-lane" layout) - the key design decision of this library:
//   a tile of 16 token rows x F features lives in the registers of ONE wavefront as f32x4 x[F/16];
//   lane l = (g = l >> 4, r = l & 15) holds, for feature tile t and m = 0..3,
//       x[t][m] = X[row r][feature 16 t + 4 g + m].
//   This is exactly the C/D fragment layout of v_mfma_f32_16x16x4_f32 / v_mfma_f32_16x16x32_bf16
//   when the WEIGHT matrix is the A operand (M = output features) and the activations are the B
//   operand (N = rows).  Because the host packs every weight matrix with the K index permuted to
//   match (k-slot (g, m) of k-tile t <-> feature 16 t + 4 g + m), the accumulator registers of one
//   GEMM are, unchanged (fp32) or after one cvt_pk (bf16), the B-operand registers of the next:
//   activations never round-trip through LDS for a layout change.
//
// Weight streams: each of the 4 waves of a workgroup consumes its own contiguous stream of 1 KiB
// "units" (64 lanes x 16 B), packed on the host in the exact order the kernel issues them.
//   fp32 unit  (o, t)      : lane (g,i) -> { W[16 o + i][16 t + 4 g + m] }, m = 0..3
//   bf16 unit  (o, t0|t1)  : lane (g,i) -> { W[16 o + i][16 t0 + 4 g + e] e<4 , W[..][16 t1 + 4 g + e-4] }
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace amuse {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int PREC_F32 = 0;
constexpr int PREC_BF16 = 1;
// "fp32x": fp32-class results at the 16-bit MFMA rate.  Weights and activations are split into two fp16 pieces each
// (x = hi + lo with hi = rn16(x), lo = rn16(x - hi): 22 significand bits) and a product is three MFMAs,
// Wh.xh + Wh.xl + Wl.xh, accumulated in fp32 (the dropped Wl.xl term is 2^-22 relative).  Everything outside the GEMMs
// (softmax, LayerNorm, erf GELU, scheduler update) is the PREC_F32 code.  Same bytes per weight as fp32.
constexpr int PREC_F16X2 = 2;
// "fp16": the throughput mode with fp16 instead of bf16 operands - the bf16 kernels' instruction stream (one
// v_mfma_f32_16x16x32_f16 per product, 2 bytes per weight), 11 significand bits instead of 8: an eighth of the bf16 mode's drift
// against fp32 at the same speed.  Range as for PREC_F16X2 (|operands| < 65504; subnormals kept by the MFMA).
constexpr int PREC_F16 = 3;

constexpr int kD = 128;       // d_model
constexpr int kTiles = 8;     // kD / 16
constexpr int kHeads = 4;
constexpr int kFF = 512;
constexpr int kLayers = 9;
constexpr int kCond = 256;
constexpr int kFrames = 300;
constexpr int kJoints = 55;
constexpr int kFeats = 333;
constexpr int kFeatTiles = 24;  // 333 -> 21 tiles, padded to 24 (6 per wave)

// per-block small-parameter vector (fp32), offsets in floats
constexpr int PV_IN_B = 0;        // [384] self_attn.in_proj_bias
constexpr int PV_OUT_B = 384;     // [128] self_attn.out_proj.bias
constexpr int PV_L1_B = 512;      // [512] linear1.bias
constexpr int PV_L2_B = 1024;     // [128] linear2.bias
constexpr int PV_LN1_W = 1152, PV_LN1_B = 1280;
constexpr int PV_LN2_W = 1408, PV_LN2_B = 1536;
constexpr int PV_LN3_W = 1664, PV_LN3_B = 1792;  // decoder blocks only
constexpr int PV_BLOCK = 1920;
// after the 9 blocks: 4 x [128] skip-linear bias, then final LayerNorm weight, bias
constexpr int PV_SKIP_B = kLayers * PV_BLOCK;
constexpr int PV_FINAL_W = PV_SKIP_B + 4 * 128;
constexpr int PV_FINAL_B = PV_FINAL_W + 128;
constexpr int PV_TOTAL = PV_FINAL_B + 128;

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ f32x4 splat4(float v) { return f32x4{v, v, v, v}; }

__device__ __forceinline__ f32x4 mfma_f32(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma_bf16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma_f16(f16x8 a, f16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
// Two feature tiles as one split MFMA operand (k-tile pair): hi = rn16(x) (v_cvt_pk_f16_f32, round-to-nearest-even),
// lo = rn16(x - hi) - the subtraction is exact in fp32.
struct F16Pair {
    f16x8 hi, lo;
};
__device__ __forceinline__ F16Pair split_f16(f32x4 a, f32x4 b) {
    const f32x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    F16Pair p;
    p.hi = __builtin_convertvector(v, f16x8);
    p.lo = __builtin_convertvector(v - __builtin_convertvector(p.hi, f32x8), f16x8);
    return p;
}
__device__ __forceinline__ bf16x8 pack_bf16(f32x4 lo, f32x4 hi) {
    f32x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_convertvector(v, bf16x8);  // v_cvt_pk_bf16_f32, round-to-nearest-even
}

__device__ __forceinline__ f16x8 pack_f16(f32x4 lo, f32x4 hi) {
    f32x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_convertvector(v, f16x8);  // v_cvt_pk_f16_f32, round-to-nearest-even
}
// operand format of the one-piece 16-bit modes (PREC_BF16, PREC_F16): vector type, pack, MFMA
template <int PREC> struct Op16;
template <> struct Op16<PREC_BF16> {
    typedef bf16x8 vec;
    static __device__ __forceinline__ vec pack(f32x4 lo, f32x4 hi) { return pack_bf16(lo, hi); }
    static __device__ __forceinline__ f32x4 mfma(vec a, vec b, f32x4 c) { return mfma_bf16(a, b, c); }
};
template <> struct Op16<PREC_F16> {
    typedef f16x8 vec;
    static __device__ __forceinline__ vec pack(f32x4 lo, f32x4 hi) { return pack_f16(lo, hi); }
    static __device__ __forceinline__ f32x4 mfma(vec a, vec b, f32x4 c) { return mfma_f16(a, b, c); }
};
constexpr bool is_op16(int prec) { return prec == PREC_BF16 || prec == PREC_F16; }

// acc[o] (+)= W_o . x   over NK k-tiles held in x[], consuming the wave's weight stream `w`
// (already offset by lane).  SWAP = false: row-lane result (weights = A operand).
// SWAP = true: feature-lane result (activations = A): lane (g, f) holds rows 4 g + m of feature f.
// Stream order: k-tile (fp32) / k-tile pair (bf16) outer, output tile inner.
template <int PREC, int NO, int NK, bool SWAP>
__device__ __forceinline__ const uint4* gemm_tiles(f32x4 (&acc)[NO], const f32x4 (&x)[NK],
                                                   const uint4* __restrict__ w) {
    if constexpr (PREC == PREC_F32) {
#pragma unroll
        for (int t = 0; t < NK; ++t) {
            f32x4 wf[NO];
#pragma unroll
            for (int o = 0; o < NO; ++o) wf[o] = *reinterpret_cast<const f32x4*>(w + (t * NO + o) * 64);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
#pragma unroll
                for (int o = 0; o < NO; ++o)
                    acc[o] = SWAP ? mfma_f32(x[t][m], wf[o][m], acc[o]) : mfma_f32(wf[o][m], x[t][m], acc[o]);
            }
        }
        return w + NK * NO * 64;
    } else {
        static_assert(NK % 2 == 0 && is_op16(PREC), "16-bit units cover k-tile pairs");
        typedef Op16<PREC> Op;
#pragma unroll
        for (int c = 0; c < NK / 2; ++c) {
            const typename Op::vec xb = Op::pack(x[2 * c], x[2 * c + 1]);
#pragma unroll
            for (int o = 0; o < NO; ++o) {
                const typename Op::vec wf = *reinterpret_cast<const typename Op::vec*>(w + (c * NO + o) * 64);
                acc[o] = SWAP ? Op::mfma(xb, wf, acc[o]) : Op::mfma(wf, xb, acc[o]);
            }
        }
        return w + (NK / 2) * NO * 64;
    }
}

// ---- software-pipelined variant: the wave's weight stream flows through a ring of R register slots
// (R x 1 KiB in flight per wave).  Every unit is consumed from its slot and the slot is immediately
// re-armed with the unit R positions further down the stream, so loads run a full revolution ahead of
// their use - across GEMM stages, barriers, blocks and denoising steps (the stream is sequential and
// its first R units are replicated after its end, so the wrap at a step boundary needs no special case).
// hipcc left to itself serialises this kernel's loads (global_load -> s_waitcnt vmcnt(0) -> v_mfma,
// one L2 round trip per KiB); the ring is what turns the loop from latency- into bandwidth-bound.
// Stream loads are plain loads: non-temporal ones (whole stream, or only the `lo` units of a split-fp16 stream so that the `hi` half alone
// competes for an XCD's 4 MB L2) measured SLOWER (profiles/r04_fp32x_lo_nt_ab.txt) - a non-temporal line is not kept for the XCD's other 31
// CUs, which then fetch it over the fabric themselves.
__device__ __forceinline__ uint4 ldw(const uint4* p) {
    return *p;
}
__device__ __forceinline__ uint4 ldw_pos(const uint4* p, int pos) {
    return ldw(p);
}
// Lazy rescaling in the online-softmax loops whose scores leave the MFMA relative to the running maximum (amuse_fused.hpp attend, k_vae_fusedx.hip attend_x, k_audio.hip):
// a chunk only moves a tile's running maxima when some score exceeds its row's by more than kAttnTau log2 units.  Until then p = exp2(s - m_run) may reach 2^kAttnTau
// instead of 1 - harmless: bf16 / fp16 / fp32 keep their relative precision there (fp16: 2^6 is far from 65504), the row sums carry the same scale and the final division
// removes it.  With the rule of rounds 2-5 (threshold 0) the "rare" rescale ran in nearly every chunk - 16 rows x 64 fresh keys almost always hold some new row maximum
// (record statistics: probability 1 / (chunk + 1) per row) - and cost ~35 VALU instructions beside the chunk's own ~30: measured on MI355X (round 6,
// profiles/r06_attention_lazy_rescale.txt) the decoder's S = 300 attention went from 0.27 to 0.31 of the MFMA peak.  Thresholds 4 .. 12 time the same and give the same bits on
// the test data (no chunk behind the first moves a maximum by 16 x).
constexpr float kAttnTau = 6.0f;
template <int R>
struct WRing {
    uint4 s[R];
    const uint4* next;  // lane-offset address of the unit that re-arms the next consumed slot
};
template <int R>
__device__ __forceinline__ void ring_fill(WRing<R>& rg, const uint4* w) {
#pragma unroll
    for (int i = 0; i < R; ++i) rg.s[i] = ldw(w + i * 64);
    rg.next = w + R * 64;
}
// Issue the next N units of the stream into slots IPH.. (they must have been consumed already).  Decoupling
// this from consumption lets the loads be ISSUED during the VALU/LDS phases between GEMMs: a wave's
// global_load blocks at issue once the CU's 64 B/clk load path is saturated, so a GEMM that re-arms every slot
// as it consumes it runs at load-issue speed, not MFMA speed, while the path idles during the phases in between.
template <int N, int R, int IPH>
__device__ __forceinline__ void ring_issue(WRing<R>& rg) {
    if constexpr (N > 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) rg.s[(IPH + i) % R] = ldw_pos(rg.next + i * 64, IPH + i);
        rg.next += N * 64;
        __builtin_amdgcn_sched_barrier(0);
    }
}
// PREC_F16X2 with the activations already split (one split serves every GEMM that reads the same rows): NP k-tile pairs;
// per (pair c, output tile o) the stream holds two units, Wh then Wl.  The three MFMAs of a product go out term-major
// over the output tiles, so consecutive MFMAs never hit the same accumulator.  Small terms first.
template <int NO, int NP, bool SWAP, int R, int PH, bool REARM = true>
__device__ __forceinline__ void gemm_ring_s(f32x4 (&acc)[NO], const F16Pair (&xs)[NP], WRing<R>& rg) {
    // output tiles in groups of OG (at most four: eight would keep 64 registers of weight pieces live beside the ring)
    constexpr int OG = NO % 4 == 0 ? 4 : (NO % 2 == 0 ? 2 : 1);
#pragma unroll
    for (int c = 0; c < NP; ++c) {
#pragma unroll
        for (int o0 = 0; o0 < NO; o0 += OG) {
            f16x8 wh[OG], wl[OG];
#pragma unroll
            for (int i = 0; i < OG; ++i) {
                const int s0 = (PH + 2 * (c * NO + o0 + i)) % R, s1 = (PH + 2 * (c * NO + o0 + i) + 1) % R;
                wh[i] = __builtin_bit_cast(f16x8, rg.s[s0]);
                wl[i] = __builtin_bit_cast(f16x8, rg.s[s1]);
                if constexpr (REARM) {
                    rg.s[s0] = ldw(rg.next);
                    rg.s[s1] = ldw(rg.next + 64);
                    rg.next += 128;
                }
            }
#pragma unroll
            for (int i = 0; i < OG; ++i) acc[o0 + i] = SWAP ? mfma_f16(xs[c].hi, wl[i], acc[o0 + i]) : mfma_f16(wl[i], xs[c].hi, acc[o0 + i]);
#pragma unroll
            for (int i = 0; i < OG; ++i) acc[o0 + i] = SWAP ? mfma_f16(xs[c].lo, wh[i], acc[o0 + i]) : mfma_f16(wh[i], xs[c].lo, acc[o0 + i]);
#pragma unroll
            for (int i = 0; i < OG; ++i) acc[o0 + i] = SWAP ? mfma_f16(xs[c].hi, wh[i], acc[o0 + i]) : mfma_f16(wh[i], xs[c].hi, acc[o0 + i]);
        }
    }
}
template <int NK>
__device__ __forceinline__ void split_rows(F16Pair (&xs)[NK / 2], const f32x4 (&x)[NK]) {
#pragma unroll
    for (int c = 0; c < NK / 2; ++c) xs[c] = split_f16(x[2 * c], x[2 * c + 1]);
}

// PH = ring phase (slot of the first unit) at entry; the caller tracks it at compile time.
// REARM: re-arm each slot as it is consumed (true) or leave that to later ring_issue calls (false).
template <int PREC, int NO, int NK, bool SWAP, int R, int PH, bool REARM = true>
__device__ __forceinline__ void gemm_ring(f32x4 (&acc)[NO], const f32x4 (&x)[NK], WRing<R>& rg) {
    if constexpr (PREC == PREC_F16X2) {
        static_assert(NK % 2 == 0, "f16x2 units cover k-tile pairs");
        F16Pair xs[NK / 2];
        split_rows<NK>(xs, x);
        gemm_ring_s<NO, NK / 2, SWAP, R, PH, REARM>(acc, xs, rg);
    } else if constexpr (PREC == PREC_F32) {
        static_assert(NO % 2 == 0, "output tiles are processed in pairs");
#pragma unroll
        for (int t = 0; t < NK; ++t) {
#pragma unroll
            for (int o = 0; o < NO; o += 2) {
                // two units at a time, MFMAs alternating between their accumulators: a v_mfma_f32_16x16x4_f32
                // chain on ONE accumulator pays 40 cycles per instruction instead of the 32-cycle issue rate
                const int s0 = (PH + t * NO + o) % R, s1 = (PH + t * NO + o + 1) % R;
                const f32x4 w0 = __builtin_bit_cast(f32x4, rg.s[s0]);
                const f32x4 w1 = __builtin_bit_cast(f32x4, rg.s[s1]);
                if constexpr (REARM) {
                    rg.s[s0] = ldw(rg.next);
                    rg.s[s1] = ldw(rg.next + 64);
                    rg.next += 128;
                }
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    acc[o] = SWAP ? mfma_f32(x[t][m], w0[m], acc[o]) : mfma_f32(w0[m], x[t][m], acc[o]);
                    acc[o + 1] = SWAP ? mfma_f32(x[t][m], w1[m], acc[o + 1]) : mfma_f32(w1[m], x[t][m], acc[o + 1]);
                }
            }
        }
    } else {
        static_assert(NK % 2 == 0 && is_op16(PREC), "16-bit units cover k-tile pairs");
        typedef Op16<PREC> Op;
#pragma unroll
        for (int c = 0; c < NK / 2; ++c) {
            const typename Op::vec xb = Op::pack(x[2 * c], x[2 * c + 1]);
#pragma unroll
            for (int o = 0; o < NO; ++o) {
                const int slot = (PH + c * NO + o) % R;
                const uint4 u = rg.s[slot];
                if constexpr (REARM) {
                    rg.s[slot] = ldw(rg.next);
                    rg.next += 64;
                }
                const typename Op::vec wf = __builtin_bit_cast(typename Op::vec, u);
                acc[o] = SWAP ? Op::mfma(xb, wf, acc[o]) : Op::mfma(wf, xb, acc[o]);
            }
        }
    }
}

// consume N padding units (re-arm their slots, use nothing)
template <int N, int R, int PH>
__device__ __forceinline__ void ring_discard(WRing<R>& rg) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        rg.s[(PH + i) % R] = ldw(rg.next);
        rg.next += 64;
    }
}

// units (1 KiB) consumed by gemm_tiles<PREC, NO, NK>
// (f16x2: k-tile pairs x two pieces - the fp32 count)
constexpr int gemm_units(int prec, int no, int nk) { return is_op16(prec) ? no * nk / 2 : no * nk; }

// All-reduce over the four 16-lane rows of a wavefront (lanes l, l^16, l^32, l^48 - the "g" axis of the
// row-lane layout) in two VALU instructions each: v_permlane16_swap / v_permlane32_swap exchange whole rows
// between two registers, so (swap(v, v)[0] op swap(v, v)[1]) is the xor-16 resp. xor-32 butterfly without
// the LDS round trip of ds_bpermute (__shfl_xor).  Every lane ends with the same value, combined in the
// same order.
// (Written as inline asm: ROCm 7.2 hipcc folds __builtin_amdgcn_permlane16_swap(u, u)[1] into [0] - it emits
// v_add v, v3, v3 - whenever both operands carry the same value.  The s_nop covers the "VALU write ->
// v_permlane*_swap read" hazard, which hipcc does not pad inside an asm statement.)
__device__ __forceinline__ void swap_rows16(float& a, float& b) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void swap_rows32(float& a, float& b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float allreduce_g_sum(float v) {
    float a = v, b = v;
    swap_rows16(a, b);   // a = [v0 v0 v2 v2], b = [v1 v1 v3 v3] (rows of 16 lanes)
    float s = a + b, t = s;
    swap_rows32(s, t);   // s = [lo lo], t = [hi hi]
    return s + t;
}
__device__ __forceinline__ float allreduce_g_max(float v) {
    float a = v, b = v;
    swap_rows16(a, b);
    float s = fmaxf(a, b), t = s;
    swap_rows32(s, t);
    return fmaxf(s, t);
}

// LayerNorm over the 128 features of each row, row-lane layout.  eps 1e-5, biased variance
// (nn.LayerNorm).  gamma/beta are fp32 [128] (LDS or global).  FAST: v_rsq_f32 instead of 1/sqrt.
struct LnParams {
    f32x4 ga[kTiles], be[kTiles];
};
// issue the 16 parameter reads early (a phase ahead of their use) so their LDS/L2 latency is off the chain
__device__ __forceinline__ void ln_params_load(LnParams& p, const float* gamma, const float* beta, int g) {
#pragma unroll
    for (int t = 0; t < kTiles; ++t) {
        p.ga[t] = ld4(gamma + 16 * t + 4 * g);
        p.be[t] = ld4(beta + 16 * t + 4 * g);
    }
}
template <bool FAST = false>
__device__ __forceinline__ void layer_norm_rows(f32x4 (&x)[kTiles], const LnParams& p) {
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < kTiles; ++t) s += (x[t][0] + x[t][1]) + (x[t][2] + x[t][3]);
    s = allreduce_g_sum(s);
    const float mean = s * (1.0f / kD);
    float v = 0.f;
#pragma unroll
    for (int t = 0; t < kTiles; ++t) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const float d = x[t][m] - mean;
            v += d * d;
        }
    }
    v = allreduce_g_sum(v);
    const float rstd = FAST ? __builtin_amdgcn_rsqf(v * (1.0f / kD) + 1e-5f) : 1.0f / sqrtf(v * (1.0f / kD) + 1e-5f);
#pragma unroll
    for (int t = 0; t < kTiles; ++t) {
#pragma unroll
        for (int m = 0; m < 4; ++m) x[t][m] = (x[t][m] - mean) * rstd * p.ga[t][m] + p.be[t][m];
    }
}
template <bool FAST = false>
__device__ __forceinline__ void layer_norm_rows(f32x4 (&x)[kTiles], const float* gamma, const float* beta, int g) {
    LnParams p;
    ln_params_load(p, gamma, beta, g);
    layer_norm_rows<FAST>(x, p);
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// exact-erf GELU through Abramowitz-Stegun 7.1.26 (|erf error| <= 1.5e-7, i.e. fp32 rounding class) with the
// hardware rcp / exp2: ~14 VALU instead of ~60 for erff.  Used by the bf16 throughput mode.
__device__ __forceinline__ float gelu_erf_fast(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(t, 1.061405429f, -1.453152027f);
    p = fmaf(t, p, 1.421413741f);
    p = fmaf(t, p, -0.284496736f);
    p = fmaf(t, p, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-1.44269504088896340736f * z * z);
    const float erfabs = fmaf(-(p * t), e, 1.0f);
    const float hx = 0.5f * x;
    return fmaf(fabsf(hx), erfabs, hx);  // 0.5 x + 0.5 |x| erf(|x|/sqrt2) == 0.5 x (1 + erf(x/sqrt2))
}

// Branch-free fp32 erf for the fp32x kernels (libm's erff compiles to one divergent branch per element - 24 of them per FFN
// phase and wave, with live registers spilled around each): |z| < 0.875: z + z P(z^2); otherwise 1 - exp2(Q(min(|z|, 4))) with Q
// fitted to log2(erfc), both evaluated and selected.  Coefficients from tools/fit_erf.py; max |error| 7.9e-8 (1.3 ulp at 1),
// relative error <= 9.8e-8 - and the GELU built on it errs by 4.5e-7 at most over [-8, 8], exactly what torch's own fp32
// 0.5 x (1 + erf(x / sqrt2)) does against float64.
__device__ __forceinline__ float erf_bf(float z) {
    const float az = fabsf(z), t = az * az;
    float p = 8.698958845343441e-05f;
    p = fmaf(p, t, -0.000821653928142041f);
    p = fmaf(p, t, 0.005207217764109373f);
    p = fmaf(p, t, -0.026861771941184998f);
    p = fmaf(p, t, 0.11283736675977707f);
    p = fmaf(p, t, -0.37612634897232056f);
    p = fmaf(p, t, 0.12837916612625122f);
    const float r1 = fmaf(az, p, az);
    const float zc = fminf(az, 4.0f);
    float q = -1.0750341061793733e-08f;
    q = fmaf(q, zc, 2.5535098302498227e-06f);
    q = fmaf(q, zc, -6.779012619517744e-05f);
    q = fmaf(q, zc, 0.0008653029217384756f);
    q = fmaf(q, zc, -0.0068697636015713215f);
    q = fmaf(q, zc, 0.038096215575933456f);
    q = fmaf(q, zc, -0.15873675048351288f);
    q = fmaf(q, zc, -0.9116426110267639f);
    q = fmaf(q, zc, -1.6305032968521118f);
    q = fmaf(q, zc, 0.0004394356219563633f);
    const float r2 = 1.0f - __builtin_amdgcn_exp2f(q);
    return copysignf(az < 0.875f ? r1 : r2, z);
}
__device__ __forceinline__ float gelu_erf_bf(float x) { return 0.5f * x * (1.0f + erf_bf(x * 0.70710678118654752440f)); }

// GELU for values that are rounded to bf16 right away (the FFN hidden activations of the 8-wave bf16 sampling kernel):
// erf(a / sqrt2) ~ a P(a^2) with a = clamp(x, +-3 sqrt2) and P of degree 7 (minimax fit with the value at the clamp
// point pinned to 1, |erf error| <= 8.7e-5) - an odd function, so no abs / sign handling.  No transcendental:
// v_med3 + mul / fma, which hipcc packs two floats at a time (v_pk_fma_f32) - 6.5 issue slots per element against
// about 17 for gelu_erf_fast.  |GELU error| <= 1.9e-4 absolute, <= 0.33 bf16 ulp for x in [-2, 4].
// (The same polynomial on four scalar chains - v_fma_f32 instead of v_pk_fma_f32, the A/B the microarchitecture guide's "packed f32 beside
// MFMAs is an anti-lever" asks for - measured no faster: profiles/r03_k_sample8_packed_f32_ab.txt, r05_k_sample8_noslp_ab.txt.)
__device__ __forceinline__ float gelu_poly1(float x) {
    constexpr float X0 = 4.24264068711928514641f;
    const float a = __builtin_amdgcn_fmed3f(x, -X0, X0), s = a * a;
    float p = -2.152084733e-09f;
    p = fmaf(p, s, 1.840825661e-07f);
    p = fmaf(p, s, -6.815091183e-06f);
    p = fmaf(p, s, 1.449597330e-04f);
    p = fmaf(p, s, -1.993848477e-03f);
    p = fmaf(p, s, 1.900408231e-02f);
    p = fmaf(p, s, -1.319021881e-01f);
    p = fmaf(p, s, 7.975201607e-01f);
    const float hx = 0.5f * x;
    return fmaf(hx, a * p, hx);
}
__device__ __forceinline__ f32x4 gelu_poly4(f32x4 x) {
    constexpr float X0 = 4.24264068711928514641f;
    f32x4 a;
#pragma unroll
    for (int m = 0; m < 4; ++m) a[m] = __builtin_amdgcn_fmed3f(x[m], -X0, X0);
    const f32x4 s = a * a;
    f32x4 p = splat4(-2.152084733e-09f);
    p = p * s + splat4(1.840825661e-07f);
    p = p * s + splat4(-6.815091183e-06f);
    p = p * s + splat4(1.449597330e-04f);
    p = p * s + splat4(-1.993848477e-03f);
    p = p * s + splat4(1.900408231e-02f);
    p = p * s + splat4(-1.319021881e-01f);
    p = p * s + splat4(7.975201607e-01f);
    const f32x4 hx = 0.5f * x;
    return hx * (a * p) + hx;  // 0.5 x (1 + erf(x / sqrt2))
}

// The fp16 mode's FFN activation: the same clamped odd polynomial one degree higher (tools/fit_gelu_poly.py 8 3.0): |erf error| <=
// 3.0e-5, |GELU error| <= 6.3e-5 absolute (at |x| ~ 4, i.e. 1.5e-5 relative) - a tenth of an fp16 ulp of the result it is rounded to.
__device__ __forceinline__ f32x4 gelu_poly4h(f32x4 x) {
    constexpr float X0 = 4.24264068711928514641f;
    f32x4 a;
#pragma unroll
    for (int m = 0; m < 4; ++m) a[m] = __builtin_amdgcn_fmed3f(x[m], -X0, X0);
    const f32x4 s = a * a;
    f32x4 p = splat4(1.232038360e-10f);
    p = p * s + splat4(-1.159289287e-08f);
    p = p * s + splat4(4.817333092e-07f);
    p = p * s + splat4(-1.179083301e-05f);
    p = p * s + splat4(1.923068630e-04f);
    p = p * s + splat4(-2.249475103e-03f);
    p = p * s + splat4(1.973768137e-02f);
    p = p * s + splat4(-1.328561008e-01f);
    p = p * s + splat4(7.978953719e-01f);
    const f32x4 hx = 0.5f * x;
    return hx * (a * p) + hx;  // 0.5 x (1 + erf(x / sqrt2))
}

// Split-K combine across the 4 waves of a workgroup: every wave publishes its partial [16 x 128]
// tile, one barrier, every wave sums all four in the same order (so all waves hold bit-identical
// copies afterwards).  `exch` = 2 x [4 waves][8 tiles][64 lanes] f32x4, alternated by `parity`, so a
// single barrier per exchange is enough (WAR on buffer p is separated from its last readers by the
// barrier of the exchange in between).
template <int W, int NC, int R, int IPH>
__device__ __forceinline__ void exchange_combine(f32x4 (&part)[kTiles], const f32x4* buf, int lane, WRing<R>* rg) {
    // reads of the other waves' partials batched 12 at a time (hipcc otherwise emits read-3 / wait / add
    // groups, one LDS round trip each), then ((p0 + p1) + p2) + p3 with the wave's own partial from registers
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        f32x4 p[4][kTiles / 2];
#pragma unroll
        for (int t = 0; t < kTiles / 2; ++t)
#pragma unroll
            for (int w = 0; w < 4; ++w)
                if (w != W) p[w][t] = buf[(w * kTiles + h * (kTiles / 2) + t) * 64 + lane];
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (NC > 0) {
            if (h == 0) ring_issue<NC, R, IPH % R>(*rg);
            else ring_issue<NC, R, (IPH + NC) % R>(*rg);
        }
#pragma unroll
        for (int t = 0; t < kTiles / 2; ++t) {
            p[W][t] = part[h * (kTiles / 2) + t];
            part[h * (kTiles / 2) + t] = ((p[0][t] + p[1][t]) + p[2][t]) + p[3][t];
        }
    }
}
// NI weight-stream units are issued in four chunks spread over the combine (before the LDS writes, before the
// barrier, and inside each half of the read/add phase); NI = 0: plain combine.
template <int NI = 0, int R = 1, int IPH = 0>
__device__ __forceinline__ void exchange_sum(f32x4 (&part)[kTiles], f32x4* exch, int& parity, int wave, int lane,
                                             WRing<R>* rg = nullptr) {
    static_assert(NI % 4 == 0, "issue count is split in four chunks");
    constexpr int NC = NI / 4;
    f32x4* buf = exch + parity * (4 * kTiles * 64);
    if constexpr (NI > 0) ring_issue<NC, R, IPH % R>(*rg);
#pragma unroll
    for (int t = 0; t < kTiles; ++t) buf[(wave * kTiles + t) * 64 + lane] = part[t];
    if constexpr (NI > 0) ring_issue<NC, R, (IPH + NC) % R>(*rg);
    __syncthreads();
    if (wave == 0) exchange_combine<0, NC, R, IPH + 2 * NC>(part, buf, lane, rg);
    else if (wave == 1) exchange_combine<1, NC, R, IPH + 2 * NC>(part, buf, lane, rg);
    else if (wave == 2) exchange_combine<2, NC, R, IPH + 2 * NC>(part, buf, lane, rg);
    else exchange_combine<3, NC, R, IPH + 2 * NC>(part, buf, lane, rg);
    parity ^= 1;
}
constexpr int kExchBytes = 2 * 4 * kTiles * 64 * 16;  // 64 KiB

// ------------------------------------------------------------------------------------------------
// Split-K combine as reduce-scatter + (LayerNorm) + all-gather.  exchange_sum makes every wave read all
// partials and then repeat the same residual + LayerNorm on the full [16 x 128] tile - 4x redundant VALU and
// 32 LDS b128 operations per wave.  Here wave W reduces only ITS two feature tiles (6 writes, 6 reads),
// normalises them (row statistics of the four 32-feature slices are merged with Chan's parallel-variance
// formula through a 512-byte LDS table), publishes them (2 writes) and reads the other six (6 reads):
// 22 LDS operations, a quarter of the adds and of the LayerNorm arithmetic, at the price of two more barriers.
// All waves end with bit-identical x (everything shared goes through LDS, merges use one fixed order).
// LDS: A = [4 waves][8 tiles][64] f32x4 (32 KiB) | stats = [4][16] float2 (512 B) | Bq = [8][64] f32x4 (8 KiB);
// single-buffered: each buffer's last readers are separated from its next writers by a later barrier of the
// same combine.
constexpr int kCombA = 4 * kTiles * 64;                 // f32x4 elements
constexpr int kCombBytes = kCombA * 16 + 4 * 16 * 8 + kTiles * 64 * 16;   // 41,472 B

template <int W, bool DO_LN, bool FAST, int NC, int R, int IPH>
__device__ __forceinline__ void combine_rs_impl(f32x4 (&part)[kTiles], f32x4 (&x)[kTiles], bool residual,
                                                const float* bias, const float* gamma, const float* beta,
                                                char* lds, int lane, WRing<R>* rg) {
    f32x4* A = reinterpret_cast<f32x4*>(lds);
    float2* stats = reinterpret_cast<float2*>(lds + kCombA * 16);
    f32x4* Bq = reinterpret_cast<f32x4*>(lds + kCombA * 16 + 4 * 16 * 8);
    const int g = lane >> 4, r = lane & 15;
    constexpr int T0 = 2 * W;
    if constexpr (NC > 0) ring_issue<NC, R, IPH % R>(*rg);
#pragma unroll
    for (int t = 0; t < kTiles; ++t)
        if (t != T0 && t != T0 + 1) A[(W * kTiles + t) * 64 + lane] = part[t];
    f32x4 bi[2], ga[2], be[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        bi[i] = ld4(bias + 16 * (T0 + i) + 4 * g);
        if constexpr (DO_LN) {
            ga[i] = ld4(gamma + 16 * (T0 + i) + 4 * g);
            be[i] = ld4(beta + 16 * (T0 + i) + 4 * g);
        }
    }
    __syncthreads();
    if constexpr (NC > 0) ring_issue<NC, R, (IPH + NC) % R>(*rg);
    f32x4 y[2];
    {
        f32x4 p[4][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int w = 0; w < 4; ++w)
                if (w != W) p[w][i] = A[(w * kTiles + T0 + i) * 64 + lane];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            p[W][i] = part[T0 + i];
            const f32x4 sum = ((p[0][i] + p[1][i]) + p[2][i]) + p[3][i];
            y[i] = residual ? x[T0 + i] + (sum + bi[i]) : sum + bi[i];
        }
    }
    if constexpr (DO_LN) {
        // statistics of this wave's 32 features of every row (lanes g = 0..3 of a row hold 8 of them each)
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
        if constexpr (NC > 0) ring_issue<NC, R, (IPH + 2 * NC) % R>(*rg);
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
    } else {
        if constexpr (NC > 0) ring_issue<NC, R, (IPH + 2 * NC) % R>(*rg);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        Bq[(T0 + i) * 64 + lane] = y[i];
        x[T0 + i] = y[i];
    }
    __syncthreads();
    if constexpr (NC > 0) ring_issue<NC, R, (IPH + 3 * NC) % R>(*rg);
#pragma unroll
    for (int t = 0; t < kTiles; ++t)
        if (t != T0 && t != T0 + 1) x[t] = Bq[t * 64 + lane];
}
// x <- [LayerNorm]( [x +] (sum over waves of part) + bias );  NI weight-stream units issued in four chunks
template <bool DO_LN, bool FAST, int NI = 0, int R = 1, int IPH = 0>
__device__ __forceinline__ void combine_rs(f32x4 (&part)[kTiles], f32x4 (&x)[kTiles], bool residual, const float* bias,
                                           const float* gamma, const float* beta, char* lds, int wave, int lane,
                                           WRing<R>* rg = nullptr) {
    static_assert(NI % 4 == 0, "issue count is split in four chunks");
    constexpr int NC = NI / 4;
    if (wave == 0) combine_rs_impl<0, DO_LN, FAST, NC, R, IPH>(part, x, residual, bias, gamma, beta, lds, lane, rg);
    else if (wave == 1) combine_rs_impl<1, DO_LN, FAST, NC, R, IPH>(part, x, residual, bias, gamma, beta, lds, lane, rg);
    else if (wave == 2) combine_rs_impl<2, DO_LN, FAST, NC, R, IPH>(part, x, residual, bias, gamma, beta, lds, lane, rg);
    else combine_rs_impl<3, DO_LN, FAST, NC, R, IPH>(part, x, residual, bias, gamma, beta, lds, lane, rg);
}

// ---- counter-based normals: Philox4x32-10, key = seed, counter = (clip, step, feature/4, stream)
__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
        const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
        c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
__device__ __forceinline__ f32x4 counter_normal4(uint64_t seed, uint64_t clip, uint32_t step, uint32_t q, uint32_t stream) {
    uint32_t c[4] = {(uint32_t)clip, step, q, stream};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    float u[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) u[i] = (float)(c[i] >> 8) * 5.9604644775390625e-8f + 2.98023223876953125e-8f;
    // Box-Muller on the hardware transcendentals: v_log_f32 (log2), v_sqrt_f32, v_sin/v_cos_f32 (argument in
    // revolutions, so sin(2 pi u) is v_sin_f32(u): no range reduction).  r = sqrt(-2 ln u) = sqrt(-2 ln2 log2 u)
    const float r0 = __builtin_amdgcn_sqrtf(-1.38629436111989061883f * __builtin_amdgcn_logf(u[0]));
    const float r1 = __builtin_amdgcn_sqrtf(-1.38629436111989061883f * __builtin_amdgcn_logf(u[2]));
    return f32x4{r0 * __builtin_amdgcn_cosf(u[1]), r0 * __builtin_amdgcn_sinf(u[1]),
                 r1 * __builtin_amdgcn_cosf(u[3]), r1 * __builtin_amdgcn_sinf(u[3])};
}

// ---- shared by the decode kernels (k_vae.hip, k_vae_fused.hip)
// four floats <-> four bf16 (round-to-nearest-even, v_cvt_pk_bf16_f32; widening is exact)
__device__ __forceinline__ uint2 f32_to_bf16x4(f32x4 v) {
    const uint4 u = __builtin_bit_cast(uint4, pack_bf16(v, splat4(0.f)));
    return uint2{u.x, u.y};
}
__device__ __forceinline__ uint2 f32_to_f16x4(f32x4 v) {
    const uint4 u = __builtin_bit_cast(uint4, pack_f16(v, splat4(0.f)));
    return uint2{u.x, u.y};
}
__device__ __forceinline__ f32x4 bf16x4_to_f32(uint2 u) {
    return f32x4{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                 __uint_as_float(u.y & 0xffff0000u)};
}
__device__ __forceinline__ f32x4 f16x4_to_f32(uint2 u) {
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    return __builtin_convertvector(__builtin_bit_cast(f16x4, u), f32x4);   // exact
}
// the one-piece 16-bit modes' conversions and FFN activation by precision tag (PREC_BF16 / PREC_F16)
template <int PREC> __device__ __forceinline__ uint2 f32_to_x16x4(f32x4 v) {
    if constexpr (PREC == PREC_F16) return f32_to_f16x4(v); else return f32_to_bf16x4(v);
}
template <int PREC> __device__ __forceinline__ f32x4 x16x4_to_f32(uint2 u) {
    if constexpr (PREC == PREC_F16) return f16x4_to_f32(u); else return bf16x4_to_f32(u);
}
template <int PREC> __device__ __forceinline__ f32x4 gelu_poly16(f32x4 x) {
    if constexpr (PREC == PREC_F16) return gelu_poly4h(x); else return gelu_poly4(x);
}


// 6D -> rotation matrix -> quaternion -> axis-angle of one joint (infer_ldm.py:168-173)
__device__ __forceinline__ void rot6d_to_axis_angle(const float* d6, int quat_mode, float (&aa)[3]) {
    // rotation_6d_to_matrix (pytorch3d; vendored copy rotation_conversions.py:512-533)
    const float a1x = d6[0], a1y = d6[1], a1z = d6[2], a2x = d6[3], a2y = d6[4], a2z = d6[5];
    const float n1 = fmaxf(sqrtf(a1x * a1x + a1y * a1y + a1z * a1z), 1e-12f);
    const float b1x = a1x / n1, b1y = a1y / n1, b1z = a1z / n1;
    const float dt = b1x * a2x + b1y * a2y + b1z * a2z;
    float b2x = a2x - dt * b1x, b2y = a2y - dt * b1y, b2z = a2z - dt * b1z;
    const float n2 = fmaxf(sqrtf(b2x * b2x + b2y * b2y + b2z * b2z), 1e-12f);
    b2x /= n2; b2y /= n2; b2z /= n2;
    const float b3x = b1y * b2z - b1z * b2y, b3y = b1z * b2x - b1x * b2z, b3z = b1x * b2y - b1y * b2x;
    const float m00 = b1x, m01 = b1y, m02 = b1z, m10 = b2x, m11 = b2y, m12 = b2z, m20 = b3x, m21 = b3y, m22 = b3z;
    float qw, qx, qy, qz;
    if (quat_mode == 1) {  // legacy snapshot: rotation_conversions.py:97-119
        qw = 0.5f * sqrtf(fmaxf(0.f, 1.f + m00 + m11 + m22));
        qx = 0.5f * sqrtf(fmaxf(0.f, 1.f + m00 - m11 - m22));
        qy = 0.5f * sqrtf(fmaxf(0.f, 1.f - m00 + m11 - m22));
        qz = 0.5f * sqrtf(fmaxf(0.f, 1.f - m00 - m11 + m22));
        if ((qx < 0.f) != ((m21 - m12) < 0.f)) qx = -qx;
        if ((qy < 0.f) != ((m02 - m20) < 0.f)) qy = -qy;
        if ((qz < 0.f) != ((m10 - m01) < 0.f)) qz = -qz;
    } else {  // pytorch3d >= 0.5: best-conditioned of four candidates, no sign standardisation
        const float qa0 = sqrtf(fmaxf(0.f, 1.f + m00 + m11 + m22)), qa1 = sqrtf(fmaxf(0.f, 1.f + m00 - m11 - m22));
        const float qa2 = sqrtf(fmaxf(0.f, 1.f - m00 + m11 - m22)), qa3 = sqrtf(fmaxf(0.f, 1.f - m00 - m11 + m22));
        int best = 0;
        float qb = qa0;
        if (qa1 > qb) { qb = qa1; best = 1; }
        if (qa2 > qb) { qb = qa2; best = 2; }
        if (qa3 > qb) { qb = qa3; best = 3; }
        const float den = 2.0f * fmaxf(qb, 0.1f);
        if (best == 0) { qw = qa0 * qa0; qx = m21 - m12; qy = m02 - m20; qz = m10 - m01; }
        else if (best == 1) { qw = m21 - m12; qx = qa1 * qa1; qy = m10 + m01; qz = m02 + m20; }
        else if (best == 2) { qw = m02 - m20; qx = m10 + m01; qy = qa2 * qa2; qz = m12 + m21; }
        else { qw = m10 - m01; qx = m20 + m02; qy = m21 + m12; qz = qa3 * qa3; }
        qw /= den; qx /= den; qy /= den; qz /= den;
    }
    // quaternion_to_axis_angle (rotation_conversions.py:480-509)
    const float nrm = sqrtf(qx * qx + qy * qy + qz * qz);
    const float half = atan2f(nrm, qw);
    const float ang = 2.0f * half;
    const float s = (fabsf(ang) < 1e-6f) ? (0.5f - (ang * ang) / 48.0f) : (sinf(half) / ang);
    aa[0] = qx / s; aa[1] = qy / s; aa[2] = qz / s;
}


}  // namespace amuse
