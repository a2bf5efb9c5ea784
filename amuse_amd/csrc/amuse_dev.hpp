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

constexpr int PREC_F32 = 0;
constexpr int PREC_BF16 = 1;

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
__device__ __forceinline__ bf16x8 pack_bf16(f32x4 lo, f32x4 hi) {
    f32x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_convertvector(v, bf16x8);  // v_cvt_pk_bf16_f32, round-to-nearest-even
}

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
        static_assert(NK % 2 == 0, "bf16 units cover k-tile pairs");
#pragma unroll
        for (int c = 0; c < NK / 2; ++c) {
            const bf16x8 xb = pack_bf16(x[2 * c], x[2 * c + 1]);
#pragma unroll
            for (int o = 0; o < NO; ++o) {
                const bf16x8 wf = *reinterpret_cast<const bf16x8*>(w + (c * NO + o) * 64);
                acc[o] = SWAP ? mfma_bf16(xb, wf, acc[o]) : mfma_bf16(wf, xb, acc[o]);
            }
        }
        return w + (NK / 2) * NO * 64;
    }
}

// units (1 KiB) consumed by gemm_tiles<PREC, NO, NK>
constexpr int gemm_units(int prec, int no, int nk) { return prec == PREC_F32 ? no * nk : no * nk / 2; }

// LayerNorm over the 128 features of each row, row-lane layout.  eps 1e-5, biased variance
// (nn.LayerNorm).  gamma/beta are fp32 [128] in global memory.
__device__ __forceinline__ void layer_norm_rows(f32x4 (&x)[kTiles], const float* gamma, const float* beta, int g) {
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < kTiles; ++t) s += (x[t][0] + x[t][1]) + (x[t][2] + x[t][3]);
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
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
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    const float rstd = 1.0f / sqrtf(v * (1.0f / kD) + 1e-5f);
#pragma unroll
    for (int t = 0; t < kTiles; ++t) {
        const f32x4 ga = ld4(gamma + 16 * t + 4 * g), be = ld4(beta + 16 * t + 4 * g);
#pragma unroll
        for (int m = 0; m < 4; ++m) x[t][m] = (x[t][m] - mean) * rstd * ga[m] + be[m];
    }
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// Split-K combine across the 4 waves of a workgroup: every wave publishes its partial [16 x 128]
// tile, one barrier, every wave sums all four in the same order (so all waves hold bit-identical
// copies afterwards).  `exch` = 2 x [4 waves][8 tiles][64 lanes] f32x4, alternated by `parity`, so a
// single barrier per exchange is enough (WAR on buffer p is separated from its last readers by the
// barrier of the exchange in between).
__device__ __forceinline__ void exchange_sum(f32x4 (&part)[kTiles], f32x4* exch, int& parity, int wave, int lane) {
    f32x4* buf = exch + parity * (4 * kTiles * 64);
#pragma unroll
    for (int t = 0; t < kTiles; ++t) buf[(wave * kTiles + t) * 64 + lane] = part[t];
    __syncthreads();
#pragma unroll
    for (int t = 0; t < kTiles; ++t) {
        f32x4 s = buf[(0 * kTiles + t) * 64 + lane];
        s += buf[(1 * kTiles + t) * 64 + lane];
        s += buf[(2 * kTiles + t) * 64 + lane];
        s += buf[(3 * kTiles + t) * 64 + lane];
        part[t] = s;
    }
    parity ^= 1;
}
constexpr int kExchBytes = 2 * 4 * kTiles * 64 * 16;  // 64 KiB

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
    const float r0 = sqrtf(-2.0f * logf(u[0])), r1 = sqrtf(-2.0f * logf(u[2]));
    const float t0 = 6.28318530717958647692f * u[1], t1 = 6.28318530717958647692f * u[3];
    return f32x4{r0 * cosf(t0), r0 * sinf(t0), r1 * cosf(t1), r1 * sinf(t1)};
}

}  // namespace amuse
