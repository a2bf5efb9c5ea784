// Audio front-end: PretrainedLPDM_v1.process_single_seq (reference
// models/latent_diffusion/infer_ldm.py:180-193) = kaldi fbank -> pad / normalise -> 3 x ASTModel.forward
// (models/audio/audio_main_new.py:174-204: DeiT-B distilled ViT over 2 + 12 x 101 tokens) -> feature_head.
// 260 GFLOP per encoder per clip - 37 x the 1000-step sampler - almost all of it in four GEMM shapes, so this file is
// a conventional MFMA pipeline rather than the register-resident design of the sampler:
//   k_fbank        one workgroup per frame: DC removal, pre-emphasis, Hann window, 512-point FFT in LDS, 128 mel bins
//   k_im2col       16 x 16 stride-10 patches of the [128 x 1024] spectrogram as bf16 rows (K = 256)
//   k_gemm_bf16    C = A . W^T (+ fused epilogue): 128 x 128 x 64 tiles, bf16 operands through padded LDS images that
//                  are read back as ready MFMA fragments (one ds_read_b128 per operand), fp32 accumulation.
//                  Weights are the A operand, so a lane ends up with 4 consecutive FEATURES of one token row: bias,
//                  GELU, residual and the q / k / v^T split are applied in registers and stored 8 or 16 B wide.
//   k_ln_bf16      LayerNorm of the fp32 residual stream -> bf16 GEMM operand (one wave per row)
//   k_ast_attn     flash attention, S = 1214, d = 64: S^T = K.Q^T and O^T = V^T.P^T on v_mfma_f32_16x16x32_bf16 with
//                  the softmax along registers (the layout of the S = 300 decoder attention, k_vae.hip), K and V^T
//                  streamed through LDS in 64-key chunks; V is written TRANSPOSED by the qkv epilogue.
//   k_ast_pool / k_ast_head   final LayerNorm + mean over the patch tokens, feature_head (LayerNorm + Linear 768 -> 256)
// Arithmetic: bf16 GEMM / attention operands, fp32 accumulation, fp32 residual stream, LayerNorm, softmax and GELU.
#include <cstdlib>

#include "amuse_dev.hpp"
#include "amuse_audio.hpp"

namespace amuse {
namespace {

typedef unsigned short bf16raw;
__device__ __forceinline__ unsigned pack2(float a, float b) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
__device__ __forceinline__ uint2 pack4(f32x4 v) { return uint2{pack2(v[0], v[1]), pack2(v[2], v[3])}; }

// ---------------------------------------------------------------------------------------------- fbank
// grid (kAstFrames, B), 256 threads.  Frames beyond the waveform are the padding rows of infer_ldm.py:185-188.
__global__ __launch_bounds__(256) void k_fbank(const float* __restrict__ wave, int n_samples, const float* __restrict__ window,
                                               const float* __restrict__ melw_t /*[257][128]: transposed*/,
                                               const int* __restrict__ mel_range /*[128][2]: first bin, end bin of the filter*/,
                                               float norm_mean, float inv_2std,
                                               float* __restrict__ out /*[B][1024][128]*/) {
    __shared__ float re[512], im[512], red[8];
    const int f = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
    float* dst = out + ((size_t)b * kAstFrames + f) * kAstMel;
    const int n_frames = n_samples < 400 ? 0 : 1 + (n_samples - 400) / 160;
    if (f >= n_frames) {
        if (t < kAstMel) dst[t] = (0.0f - norm_mean) * inv_2std;
        return;
    }
    const float* src = wave + (size_t)b * n_samples + (size_t)f * 160;
    const float x0 = t < 400 ? src[t] : 0.f, x1 = t + 256 < 400 ? src[t + 256] : 0.f;
    float s = x0 + x1;
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((t & 63) == 0) red[t >> 6] = s;
    __syncthreads();
    const float mean = ((red[0] + red[1]) + (red[2] + red[3])) * (1.0f / 400.0f);
    re[t] = x0 - mean;
    re[t + 256] = (t + 256 < 400) ? x1 - mean : 0.f;
    __syncthreads();
    // pre-emphasis against the previous sample (the first against itself), window, bit-reversed scatter for the FFT
    float y[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int i = t + 256 * h;
        y[h] = 0.f;
        if (i < 400) y[h] = (re[i] - 0.97f * re[i > 0 ? i - 1 : 0]) * window[i];
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int i = t + 256 * h;
        const int r = __brev((unsigned)i) >> 23;   // 9-bit reversal
        re[r] = y[h];
        im[r] = 0.f;
    }
    __syncthreads();
    for (int len = 2; len <= 512; len <<= 1) {
        const int half = len >> 1, k = t & (half - 1), base = ((t - k) << 1) + k;
        float sn, cs;
        sincospif(-2.0f * (float)k / (float)len, &sn, &cs);
        const float ur = re[base], ui = im[base], vr = re[base + half], vi = im[base + half];
        const float tr = vr * cs - vi * sn, ti = vr * sn + vi * cs;
        __syncthreads();
        re[base] = ur + tr; im[base] = ui + ti;
        re[base + half] = ur - tr; im[base + half] = ui - ti;
        __syncthreads();
    }
    // power spectrum (bins 0..256) in place, then the mel filters
    const float p0 = re[t] * re[t] + im[t] * im[t];
    const float p256 = re[256] * re[256] + im[256] * im[256];
    __syncthreads();
    re[t] = p0;
    if (t == 0) re[256] = p256;
    __syncthreads();
    if (t < kAstMel) {
        // bins in ascending order, as the dense product; the filter's zero bins outside [k0, k1) add nothing.  The
        // transposed weights make a bin's 128 loads one contiguous 512 B row.
        const int k0 = mel_range[2 * t], k1 = mel_range[2 * t + 1];
        float e = 0.f;
        for (int k = k0; k < k1; ++k) e += re[k] * melw_t[k * kAstMel + t];
        dst[t] = (logf(fmaxf(e, 1.1920929e-07f)) - norm_mean) * inv_2std;
    }
}

// ---------------------------------------------------------------------------------------------- im2col
// patches[b * 1212 + fh * 101 + tw][kh * 16 + kw] = fbank[b][10 tw + kw][10 fh + kh]   (x.unsqueeze(1).transpose(2, 3)
// then Conv2d(1, 768, 16, stride 10): audio_main_new.py:180-184, 92-96)
__global__ __launch_bounds__(256) void k_im2col(const float* __restrict__ fbank, bf16raw* __restrict__ patches, int B) {
    const size_t row = (size_t)blockIdx.x;           // b * 1212 + p
    const int b = (int)(row / kAstPatches), p = (int)(row - (size_t)b * kAstPatches);
    const int fh = p / kAstT, tw = p - fh * kAstT;
    const int kh = threadIdx.x >> 4, kw = threadIdx.x & 15;
    const float v = fbank[((size_t)b * kAstFrames + 10 * tw + kw) * kAstMel + 10 * fh + kh];
    typedef __bf16 bf;
    patches[row * 256 + threadIdx.x] = __builtin_bit_cast(bf16raw, (bf)v);
}

// cls / distillation rows of the token matrix (audio_main_new.py:185-188)
__global__ __launch_bounds__(256) void k_ast_tokens(const float* __restrict__ cls, const float* __restrict__ dist,
                                                    const float* __restrict__ pos, float* __restrict__ X) {
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < 2 * kAstDim; i += 256) {
        const int r = i / kAstDim, c = i - r * kAstDim;
        X[((size_t)b * kAstTokens + r) * kAstDim + c] = (r == 0 ? cls[c] : dist[c]) + pos[(size_t)r * kAstDim + c];
    }
}

// ---------------------------------------------------------------------------------------------- GEMM
// 128 x 128 x 64 output tiles, four waves of 64 x 64.  What bounds this kernel is LDS bandwidth: with BOTH operands
// staged through LDS a k-tile costs every wave 16 fragment reads + 8 staging writes of 1 KiB against 32 MFMAs - 135 %
// of the matrix-pipe time at the measured 114 B/clk (tools/probes/lds_probe.hip; ablation in DESIGN.md 4.4).  The
// weights are static, so they take the sampler's route instead: packed once on the host into MFMA-fragment order
// (1 KiB units, amuse_audio_api.hip pack_w) and streamed global -> registers -> MFMA by each wave, one k-tile ahead.
// Only the activations go through LDS (8 reads + 4 writes per k-tile per wave).
constexpr int BM = 128, BN = 128, BK = 64;
constexpr int LDSK = BK + 8;                       // padded row (bf16 elements): 144 B, conflict-free ds_read_b128
constexpr int kGemmLds = 2 * BM * LDSK * 2;        // double-buffered A tile: 36,864 B

// DEEP: the weights run TWO k-tiles ahead of their use (three register sets) instead of one; needs K / 64 divisible by 6.
template <int EPI, bool DEEP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_gemm_bf16(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16raw* As = reinterpret_cast<bf16raw*>(smem);                    // [2][BM][LDSK]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int g = lane >> 4, j = lane & 15;
    const int wm = wave >> 1, wn = wave & 1;
    // PERSISTENT: workgroup w computes output tiles w, w + gridDim.x, ... and the k-tile pipeline runs straight across
    // output-tile boundaries, so only the first tile of a workgroup pays the load round trip in the open and every
    // epilogue overlaps the next tile's loads.  Consecutive tile indices walk the N tiles of one M tile: concurrent
    // workgroups share the A rows through L2.
    const int tiles_n = a.N / BN;
    const int n_tiles = ((a.M + BM - 1) / BM) * tiles_n;
    const int nk = a.K / BK;
    const int lr = t >> 3, lc = (t & 7) * 8;                          // this thread's 16 B: rows lr + 32 i, k lc..lc+7
    const size_t rstep = (size_t)32 * a.K;
    bf16raw* as = As + (size_t)lr * LDSK + lc;
    // activations: global -> registers two k-tiles ahead (sets a / b) -> LDS; weights: fragment units one k-tile
    // ahead (sets 0 / 1), straight into the MFMA
    uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
    uint4 w0[8], w1[8], w2[DEEP ? 8 : 1];
#define GEMM_ALOAD(A0, A1, A2, A3, ag)                                     \
    A0 = *reinterpret_cast<const uint4*>(ag);                              \
    A1 = *reinterpret_cast<const uint4*>(ag + rstep);                      \
    A2 = *reinterpret_cast<const uint4*>(ag + 2 * rstep);                  \
    A3 = *reinterpret_cast<const uint4*>(ag + 3 * rstep);
#define GEMM_ASTORE(A0, A1, A2, A3, buf)                                               \
    *reinterpret_cast<uint4*>(as + ((buf) * BM + 0) * LDSK) = A0;                      \
    *reinterpret_cast<uint4*>(as + ((buf) * BM + 32) * LDSK) = A1;                     \
    *reinterpret_cast<uint4*>(as + ((buf) * BM + 64) * LDSK) = A2;                     \
    *reinterpret_cast<uint4*>(as + ((buf) * BM + 96) * LDSK) = A3;
    // unit (x, ks) of this wave's 64-feature span: wp + (x * (K / 32) + ks) * 64 lanes; a k-tile = k-steps 2 kt, 2 kt + 1
#define GEMM_WLOAD(WS, wp, kt)                                                                     \
    _Pragma("unroll") for (int x = 0; x < 4; ++x) {                                                \
        WS[2 * x] = wp[((size_t)x * (2 * nk) + 2 * (kt)) * 64];                                    \
        WS[2 * x + 1] = wp[((size_t)x * (2 * nk) + 2 * (kt) + 1) * 64];                            \
    }
    // the same from a pointer to the k-tile's first unit
#define GEMM_WLOADQ(WS, wq)                                                                        \
    _Pragma("unroll") for (int x = 0; x < 4; ++x) {                                                \
        WS[2 * x] = (wq)[((size_t)x * (2 * nk)) * 64];                                             \
        WS[2 * x + 1] = (wq)[((size_t)x * (2 * nk) + 1) * 64];                                     \
    }
#define GEMM_COMPUTE(buf, WS)                                                                               \
    {                                                                                                       \
        const bf16raw* Ab = As + (size_t)(buf) * BM * LDSK + (size_t)(64 * wm + j) * LDSK + 8 * g;          \
        _Pragma("unroll") for (int s = 0; s < BK / 32; ++s) {                                               \
            bf16x8 af[4];                                                                                   \
            _Pragma("unroll") for (int y = 0; y < 4; ++y)                                                   \
                af[y] = *reinterpret_cast<const bf16x8*>(Ab + (size_t)(16 * y) * LDSK + 32 * s);            \
            _Pragma("unroll") for (int x = 0; x < 4; ++x)                                                   \
                _Pragma("unroll") for (int y = 0; y < 4; ++y)                                               \
                    acc[x][y] = mfma_bf16(__builtin_bit_cast(bf16x8, WS[2 * x + s]), af[y], acc[x][y]);     \
        }                                                                                                   \
    }
    int tile = blockIdx.x;
    if (tile >= n_tiles) return;
    auto a_ptr = [&](int tl) { return a.A + ((size_t)(tl / tiles_n) * BM + lr) * a.K + lc; };
    // this wave's packed weight span of output tile tl: 64-feature span index = (tl % tiles_n) * 2 + wn
    auto w_ptr = [&](int tl) {
        return reinterpret_cast<const uint4*>(a.W) + ((size_t)((tl % tiles_n) * 2 + wn) * 4 * (2 * nk)) * 64 + lane;
    };
    const bf16raw* ag = a_ptr(tile);
    const uint4* wp = w_ptr(tile);
    GEMM_ALOAD(ra0, ra1, ra2, ra3, ag)
    GEMM_ALOAD(rb0, rb1, rb2, rb3, ag + BK)
    GEMM_WLOAD(w0, wp, 0)
    if constexpr (DEEP) { GEMM_WLOAD(w1, wp, 1) }
    GEMM_ASTORE(ra0, ra1, ra2, ra3, 0)
    __syncthreads();
    while (true) {
        const int tm_idx = tile / tiles_n, tn_idx = tile - tm_idx * tiles_n;
        const size_t m0 = (size_t)tm_idx * BM;
        const int n0 = tn_idx * BN;
        const int next_tile = tile + gridDim.x;
        const bool have_next = next_tile < n_tiles;
        const bf16raw* agn = have_next ? a_ptr(next_tile) : ag;
        const uint4* wpn = have_next ? w_ptr(next_tile) : wp;
        f32x4 acc[4][4];   // [feature fragment][token tile]
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int y = 0; y < 4; ++y) acc[x][y] = splat4(0.f);
        if constexpr (DEEP) {
            // six k-tiles per trip: LDS buffers alternate (period 2), weight sets rotate (period 3).  On entry: LDS
            // buffer 0 = A k-tile kt, register set b = A k-tile kt + 1; w0 / w1 = W k-tiles kt / kt + 1, w2 free.
            // Half d computes k-tile kt + d and fetches A and W of k-tile kt + d + 2 (of the next output tile past nk).
#define GEMM_HALF(d, buf, WC, WL, RL0, RL1, RL2, RL3, RS0, RS1, RS2, RS3)                                   \
            {                                                                                               \
                const int q = kt + (d) + 2;                                                                 \
                const bool over = q >= nk;                                                                  \
                const bf16raw* an = over ? agn + (size_t)(q - nk) * BK : ag + (size_t)q * BK;               \
                const uint4* wq = over ? wpn + (size_t)2 * (q - nk) * 64 : wp + (size_t)2 * q * 64;         \
                GEMM_WLOADQ(WL, wq)                                                                         \
                GEMM_ALOAD(RL0, RL1, RL2, RL3, an)                                                          \
                __builtin_amdgcn_sched_barrier(0);   /* the loads are ISSUED here, not sunk towards their use */ \
                GEMM_COMPUTE(buf, WC)                                                                       \
                GEMM_ASTORE(RS0, RS1, RS2, RS3, 1 - (buf))                                                  \
                __syncthreads();                                                                            \
            }
#pragma unroll 1
            for (int kt = 0; kt < nk; kt += 6) {
                GEMM_HALF(0, 0, w0, w2, ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3)
                GEMM_HALF(1, 1, w1, w0, rb0, rb1, rb2, rb3, ra0, ra1, ra2, ra3)
                GEMM_HALF(2, 0, w2, w1, ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3)
                GEMM_HALF(3, 1, w0, w2, rb0, rb1, rb2, rb3, ra0, ra1, ra2, ra3)
                GEMM_HALF(4, 0, w1, w0, ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3)
                GEMM_HALF(5, 1, w2, w1, rb0, rb1, rb2, rb3, ra0, ra1, ra2, ra3)
            }
#undef GEMM_HALF
        } else {
#pragma unroll 1
        for (int kt = 0; kt < nk; kt += 2) {
            // LDS buffer 0 = A k-tile kt, register set b = A k-tile kt + 1, set a free; w0 = W k-tile kt, w1 free.
            // NO branch in this body: the next operands' addresses are SELECTED (the k-tiles after this pair belong to
            // the next output tile when `wrap`; a workgroup's very last pair re-loads its own tile and stores it unused).
            // (With the loads behind `if (wrap) / if (more)` hipcc's s_waitcnt pass lost the queue positions at the joins
            // and drained the whole queue, vmcnt(0), inside every iteration.  Removing that changed nothing measurable:
            // the loop pays about one memory round trip per half iteration either way - see DESIGN.md 4.4.)
            const bool wrap = kt + 2 >= nk;
            const bf16raw* an = wrap ? agn : ag + (kt + 2) * BK;
            const uint4* wq = wrap ? wpn : wp + (size_t)2 * (kt + 2) * 64;
            GEMM_WLOAD(w1, wp, kt + 1)
            GEMM_ALOAD(ra0, ra1, ra2, ra3, an)
            GEMM_COMPUTE(0, w0)
            GEMM_ASTORE(rb0, rb1, rb2, rb3, 1)   // buffer 1 was last read before the previous barrier
            __syncthreads();
            GEMM_ALOAD(rb0, rb1, rb2, rb3, an + BK)
            GEMM_WLOAD(w0, wq, 0)
            GEMM_COMPUTE(1, w1)
            GEMM_ASTORE(ra0, ra1, ra2, ra3, 0)
            __syncthreads();
        }
        }
    // ---- epilogue: lane (g, j): token row m0 + 64 wm + 16 y + j, features n0 + 64 wn + 32 p + 8 g .. + 7
#pragma unroll
    for (int y = 0; y < 4; ++y) {
        const size_t row = m0 + 64 * wm + 16 * y + j;
        if (row >= (size_t)a.M) continue;
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            const int n = n0 + 64 * wn + 32 * pp + 8 * g;
            f32x4 v0 = acc[2 * pp][y] + ld4(a.bias + n), v1 = acc[2 * pp + 1][y] + ld4(a.bias + n + 4);
            if constexpr (EPI == EPI_BF16) {
                const uint2 lo = pack4(v0), hi = pack4(v1);
                *reinterpret_cast<uint4*>(a.out_bf16 + row * a.N + n) = uint4{lo.x, lo.y, hi.x, hi.y};
            } else if constexpr (EPI == EPI_GELU_BF16) {
#pragma unroll
                for (int m = 0; m < 4; ++m) { v0[m] = gelu_erf_fast(v0[m]); v1[m] = gelu_erf_fast(v1[m]); }
                const uint2 lo = pack4(v0), hi = pack4(v1);
                *reinterpret_cast<uint4*>(a.out_bf16 + row * a.N + n) = uint4{lo.x, lo.y, hi.x, hi.y};
            } else if constexpr (EPI == EPI_RESID_F32) {
                float* p = a.out_f32 + row * a.N + n;
                const f32x4 r0 = ld4(p), r1 = ld4(p + 4);
                st4(p, r0 + v0);
                st4(p + 4, r1 + v1);
            } else if constexpr (EPI == EPI_F32) {
                st4(a.out_f32 + row * a.N + n, v0);
                st4(a.out_f32 + row * a.N + n + 4, v1);
            } else if constexpr (EPI == EPI_PATCH) {
                // row = b * 1212 + p  ->  token row b * 1214 + 2 + p, + pos_embed[2 + p]
                const size_t b = row / kAstPatches, p = row - b * kAstPatches;
                float* dst = a.out_f32 + (b * kAstTokens + 2 + p) * kAstDim + n;
                const float* ps = a.pos + (2 + p) * kAstDim + n;
                st4(dst, v0 + ld4(ps));
                st4(dst + 4, v1 + ld4(ps + 4));
            } else {  // EPI_QKV: q (pre-scaled by head_dim ** -0.5 = 1/8, exact in bf16) | k row-major, v transposed
                if (n < 2 * kAstDim) {
                    const float sc = n < kAstDim ? 0.125f : 1.0f;
                    const uint2 lo = pack4(v0 * sc), hi = pack4(v1 * sc);
                    *reinterpret_cast<uint4*>(a.out_bf16 + row * (2 * kAstDim) + n) = uint4{lo.x, lo.y, hi.x, hi.y};
                } else {
                    const size_t b = row / kAstTokens, tok = row - b * kAstTokens;
                    const int hd = n - 2 * kAstDim;   // h * 64 + d
                    typedef __bf16 bf;
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        a.vt[((b * kAstDim) + hd + m) * kAstKeysPad + tok] = __builtin_bit_cast(bf16raw, (bf)v0[m]);
                        a.vt[((b * kAstDim) + hd + 4 + m) * kAstKeysPad + tok] = __builtin_bit_cast(bf16raw, (bf)v1[m]);
                    }
                }
            }
        }
    }
        if (!have_next) break;
        tile = next_tile;
        ag = agn;
        wp = wpn;
    }
#undef GEMM_ALOAD
#undef GEMM_ASTORE
#undef GEMM_WLOAD
#undef GEMM_WLOADQ
#undef GEMM_COMPUTE
}

// ---------------------------------------------------------------------------------------------- LayerNorm rows
// one wave per row of 768: fp32 in -> bf16 out
__global__ __launch_bounds__(256) void k_ln_bf16(const float* __restrict__ X, const float* __restrict__ gamma,
                                                 const float* __restrict__ beta, float eps, bf16raw* __restrict__ out, int M) {
    const int lane = threadIdx.x & 63;
    const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (size_t)M) return;
    const float* x = X + row * kAstDim;
    f32x4 v[3];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        v[i] = ld4(x + 256 * i + 4 * lane);
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s * (1.0f / kAstDim);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const float d = v[i][m] - mean;
            q += d * d;
        }
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = 1.0f / sqrtf(q * (1.0f / kAstDim) + eps);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const f32x4 ga = ld4(gamma + 256 * i + 4 * lane), be = ld4(beta + 256 * i + 4 * lane);
        f32x4 y;
#pragma unroll
        for (int m = 0; m < 4; ++m) y[m] = (v[i][m] - mean) * rstd * ga[m] + be[m];
        *reinterpret_cast<uint2*>(out + row * kAstDim + 256 * i + 4 * lane) = pack4(y);
    }
}

// ---------------------------------------------------------------------------------------------- attention
constexpr int kKc = 64;                 // keys per LDS chunk
constexpr int kKS = 64 + 8;             // padded K row (bf16): 144 B
constexpr int kVS = kKc + 8;            // padded V^T row (bf16): 144 B
// grid (19 query blocks of 64, 12 heads, B); wave w owns the 16 queries 64 qb + 16 w ..
__global__ __launch_bounds__(256) void k_ast_attn(const bf16raw* __restrict__ QK /*[M][1536]*/,
                                                  const bf16raw* __restrict__ Vt /*[B][768][1216]*/,
                                                  bf16raw* __restrict__ O /*[M][768]*/) {
    __shared__ __attribute__((aligned(16))) bf16raw Ks[kKc * kKS];
    __shared__ __attribute__((aligned(16))) bf16raw Vs[64 * kVS];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int g = lane >> 4, j = lane & 15;
    const int qb = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const size_t row0 = (size_t)b * kAstTokens;
    const int q = 64 * qb + 16 * wave + j;
    const bool qv = q < kAstTokens;
    bf16x8 qf[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        uint4 u = uint4{0, 0, 0, 0};
        if (qv) u = *reinterpret_cast<const uint4*>(QK + (row0 + q) * (2 * kAstDim) + 64 * h + 32 * s + 8 * g);
        qf[s] = __builtin_bit_cast(bf16x8, u);
    }
    float m_run = -INFINITY, l_run = 0.f;
    f32x4 o[4] = {splat4(0.f), splat4(0.f), splat4(0.f), splat4(0.f)};
    constexpr float kLog2e = 1.44269504088896340736f;
    const int lr = t >> 3, lc = (t & 7) * 8;
    const bf16raw* vsrc = Vt + ((size_t)b * kAstDim + 64 * h) * kAstKeysPad;
    for (int k0 = 0; k0 < kAstKeysPad; k0 += kKc) {
        __syncthreads();   // previous chunk fully consumed
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int key = k0 + lr + 32 * i;
            uint4 u = uint4{0, 0, 0, 0};
            if (key < kAstTokens) u = *reinterpret_cast<const uint4*>(QK + (row0 + key) * (2 * kAstDim) + kAstDim + 64 * h + lc);
            *reinterpret_cast<uint4*>(Ks + (lr + 32 * i) * kKS + lc) = u;
            // V^T rows d = lr + 32 i, keys k0 + lc .. + 7
            *reinterpret_cast<uint4*>(Vs + (lr + 32 * i) * kVS + lc) =
                *reinterpret_cast<const uint4*>(vsrc + (size_t)(lr + 32 * i) * kAstKeysPad + k0 + lc);
        }
        __syncthreads();
        // One online-softmax step per 64-key chunk (four 16-key tiles): the per-step fixed costs - two cross-lane
        // reductions, the rescale of the accumulators, exp2 of the running-max shift - are paid once per 64 keys, the
        // log2(e) scaling rides in the exp2 argument's fma, and only the chunk that holds the sequence end pays for
        // key masking.  (This loop is VALU-bound: 16 exp2 + ~50 other VALU per lane against 16 MFMAs.)
        static_assert(kKc == 64, "one softmax step per chunk");
        f32x4 st[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bf16raw* kr = Ks + (16 * u + j) * kKS + 8 * g;
            st[u] = mfma_bf16(*reinterpret_cast<const bf16x8*>(kr), qf[0], splat4(0.f));
            st[u] = mfma_bf16(*reinterpret_cast<const bf16x8*>(kr + 32), qf[1], st[u]);
        }
        // lane (g, query j): S[j][key = k0 + 16 u + 4 g + m]
        const bool tail = k0 + kKc > kAstTokens;   // uniform
        if (tail) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    if (k0 + 16 * u + 4 * g + m >= kAstTokens) st[u][m] = -INFINITY;
        }
        float mx = fmaxf(fmaxf(fmaxf(st[0][0], st[0][1]), fmaxf(st[0][2], st[0][3])),
                         fmaxf(fmaxf(st[1][0], st[1][1]), fmaxf(st[1][2], st[1][3])));
        mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(st[2][0], st[2][1]), fmaxf(st[2][2], st[2][3])),
                             fmaxf(fmaxf(st[3][0], st[3][1]), fmaxf(st[3][2], st[3][3]))));
        mx = allreduce_g_max(mx);
        const float m_new = fmaxf(m_run, mx);           // raw-score domain; every chunk holds a valid key
        const float c = m_new * kLog2e;
        const float alpha = __builtin_amdgcn_exp2f(m_run * kLog2e - c);   // m_run = -inf on the first chunk -> 0
        f32x4 p[4];
        float ps = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                p[u][m] = __builtin_amdgcn_exp2f(fmaf(st[u][m], kLog2e, -c));   // masked keys: exp2(-inf) = 0
                ps += p[u][m];
            }
        ps = allreduce_g_sum(ps);
        l_run = l_run * alpha + ps;
        m_run = m_new;
#pragma unroll
        for (int td = 0; td < 4; ++td) o[td] *= alpha;
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            // k-slots (g, e): e < 4 -> tile 2 pr key 4 g + e, else tile 2 pr + 1 key 4 g + e - 4
            const bf16x8 pb = pack_bf16(p[2 * pr], p[2 * pr + 1]);
#pragma unroll
            for (int td = 0; td < 4; ++td) {
                // A operand lane (g, i = j): V^T[d = 16 td + j][same key permutation]
                const bf16raw* vr = Vs + (16 * td + j) * kVS + 32 * pr + 4 * g;
                const uint2 lo = *reinterpret_cast<const uint2*>(vr), hi = *reinterpret_cast<const uint2*>(vr + 16);
                const bf16x8 vf = __builtin_bit_cast(bf16x8, uint4{lo.x, lo.y, hi.x, hi.y});
                o[td] = mfma_bf16(vf, pb, o[td]);
            }
        }
    }
    if (qv) {
        const float inv = 1.0f / l_run;
#pragma unroll
        for (int td = 0; td < 4; ++td)
            *reinterpret_cast<uint2*>(O + (row0 + q) * kAstDim + 64 * h + 16 * td + 4 * g) = pack4(o[td] * inv);
    }
}

// ---------------------------------------------------------------------------------------------- pooling + head
// v.norm on every token, then the mean over the 1212 patch tokens (frame_based_feats) or (cls + dist) / 2
// grid (B), 256 threads = 4 waves striding over the rows; partial sums combined through LDS
// grid (B, kAstPoolSplit): workgroup y normalises and sums its slice of the rows; k_ast_head adds the slices in a fixed
// order (one workgroup per clip walking all 1212 rows serially left the pooling at 0.3 ms, whatever the batch).
__global__ __launch_bounds__(256) void k_ast_pool(const float* __restrict__ X, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, int frame_based, float* __restrict__ pooled) {
    __shared__ float part[4][kAstDim];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.x, y = blockIdx.y;
    const int r0 = frame_based ? 2 : 0, r1 = frame_based ? kAstTokens : 2;
    const int chunk = (r1 - r0 + kAstPoolSplit - 1) / kAstPoolSplit;
    const int c0 = r0 + y * chunk, c1 = min(r1, c0 + chunk);
    f32x4 acc[3] = {splat4(0.f), splat4(0.f), splat4(0.f)};
    f32x4 ga[3], be[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { ga[i] = ld4(gamma + 256 * i + 4 * lane); be[i] = ld4(beta + 256 * i + 4 * lane); }
    for (int r = c0 + wave; r < c1; r += 4) {
        const float* x = X + ((size_t)b * kAstTokens + r) * kAstDim;
        f32x4 v[3];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) { v[i] = ld4(x + 256 * i + 4 * lane); s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]); }
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float mean = s * (1.0f / kAstDim);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int m = 0; m < 4; ++m) { const float d = v[i][m] - mean; q += d * d; }
        for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
        const float rstd = 1.0f / sqrtf(q * (1.0f / kAstDim) + 1e-6f);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[i][m] += (v[i][m] - mean) * rstd * ga[i][m] + be[i][m];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) st4(&part[wave][256 * i + 4 * lane], acc[i]);
    __syncthreads();
    for (int c = threadIdx.x; c < kAstDim; c += 256)
        pooled[((size_t)b * kAstPoolSplit + y) * kAstDim + c] = (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]);
}

// feature_head: LayerNorm(768, eps 1e-5) -> Linear(768 -> 256) with bf16-rounded operands, fp32 accumulation
__global__ __launch_bounds__(256) void k_ast_head(const float* __restrict__ pooled /*[B][kAstPoolSplit][768] row sums*/,
                                                  float inv_rows, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, const bf16raw* __restrict__ W /*[256][768]*/,
                                                  const float* __restrict__ bias, float* __restrict__ out /*[B][256]*/) {
    __shared__ float h[kAstDim];
    __shared__ float red[2][4];
    const int t = threadIdx.x, b = blockIdx.x;
    const float* x = pooled + (size_t)b * kAstPoolSplit * kAstDim;
    float v[3], s = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {   // mean over the pooled rows: the slices of k_ast_pool, added in slice order
        float a = 0.f;
        for (int y = 0; y < kAstPoolSplit; ++y) a += x[y * kAstDim + t + 256 * i];
        v[i] = a * inv_rows;
        s += v[i];
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((t & 63) == 0) red[0][t >> 6] = s;
    __syncthreads();
    const float mean = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) * (1.0f / kAstDim);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) { const float d = v[i] - mean; q += d * d; }
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    if ((t & 63) == 0) red[1][t >> 6] = q;
    __syncthreads();
    const float rstd = 1.0f / sqrtf(((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) * (1.0f / kAstDim) + 1e-5f);
    typedef __bf16 bf;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int c = t + 256 * i;
        h[c] = (float)(bf)((v[i] - mean) * rstd * gamma[c] + beta[c]);   // GEMM operand rounding
    }
    __syncthreads();
    const bf16raw* w = W + (size_t)t * kAstDim;
    float acc = 0.f;
    for (int c = 0; c < kAstDim; c += 8) {
        const bf16x8 wv = *reinterpret_cast<const bf16x8*>(w + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc += (float)wv[e] * h[c + e];
    }
    out[(size_t)b * kAstFeat + t] = acc + bias[t];
}

template <int EPI>
hipError_t launch_gemm_t(const GemmArgs& a, hipStream_t s) {
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_bf16<EPI, false>), hipFuncAttributeMaxDynamicSharedMemorySize, kGemmLds);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_bf16<EPI, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kGemmLds);
        if (e != hipSuccess) return e;
        attr = true;
    }
    const int n_tiles = ((a.M + BM - 1) / BM) * (a.N / BN);
    const int resident = 2 * 256;   // two workgroups (8 waves, 2 per SIMD at this register count) per CU, 256 CUs
    const dim3 grid(n_tiles < resident ? n_tiles : resident);
    static const bool no_deep = [] { const char* e = getenv("AMUSE_GEMM_DEEP"); return e && atoi(e) == 0; }();
    if ((a.K / BK) % 6 == 0 && !no_deep) hipLaunchKernelGGL((k_gemm_bf16<EPI, true>), grid, dim3(256), kGemmLds, s, a);
    else hipLaunchKernelGGL((k_gemm_bf16<EPI, false>), grid, dim3(256), kGemmLds, s, a);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_gemm(const GemmArgs& a, int epi, hipStream_t s) {
    switch (epi) {
        case EPI_BF16: return launch_gemm_t<EPI_BF16>(a, s);
        case EPI_GELU_BF16: return launch_gemm_t<EPI_GELU_BF16>(a, s);
        case EPI_RESID_F32: return launch_gemm_t<EPI_RESID_F32>(a, s);
        case EPI_F32: return launch_gemm_t<EPI_F32>(a, s);
        case EPI_PATCH: return launch_gemm_t<EPI_PATCH>(a, s);
        default: return launch_gemm_t<EPI_QKV>(a, s);
    }
}
hipError_t launch_fbank(const float* wave, int n_samples, int B, const float* window, const float* melw, const int* mel_range, float mean, float std,
                        float* out, hipStream_t s) {
    hipLaunchKernelGGL(k_fbank, dim3(kAstFrames, B), dim3(256), 0, s, wave, n_samples, window, melw, mel_range, mean, 1.0f / (2.0f * std), out);
    return hipGetLastError();
}
hipError_t launch_im2col(const float* fbank, unsigned short* patches, int B, hipStream_t s) {
    hipLaunchKernelGGL(k_im2col, dim3(B * kAstPatches), dim3(256), 0, s, fbank, patches, B);
    return hipGetLastError();
}
hipError_t launch_ast_tokens(const float* cls, const float* dist, const float* pos, float* X, int B, hipStream_t s) {
    hipLaunchKernelGGL(k_ast_tokens, dim3(B), dim3(256), 0, s, cls, dist, pos, X);
    return hipGetLastError();
}
hipError_t launch_ln_bf16(const float* X, const float* gamma, const float* beta, float eps, unsigned short* out, int M, hipStream_t s) {
    hipLaunchKernelGGL(k_ln_bf16, dim3((M + 3) / 4), dim3(256), 0, s, X, gamma, beta, eps, out, M);
    return hipGetLastError();
}
hipError_t launch_ast_attn(const unsigned short* QK, const unsigned short* Vt, unsigned short* O, int B, hipStream_t s) {
    hipLaunchKernelGGL(k_ast_attn, dim3((kAstTokens + 63) / 64, kAstHeads, B), dim3(256), 0, s, QK, Vt, O);
    return hipGetLastError();
}
hipError_t launch_ast_pool(const float* X, const float* gamma, const float* beta, int frame_based, float* pooled, int B, hipStream_t s) {
    hipLaunchKernelGGL(k_ast_pool, dim3(B, kAstPoolSplit), dim3(256), 0, s, X, gamma, beta, frame_based, pooled);
    return hipGetLastError();
}
hipError_t launch_ast_head(const float* pooled, int frame_based, const float* gamma, const float* beta, const unsigned short* W,
                           const float* bias, float* out, int B, hipStream_t s) {
    const float inv_rows = 1.0f / (float)(frame_based ? kAstTokens - 2 : 2);
    hipLaunchKernelGGL(k_ast_head, dim3(B), dim3(256), 0, s, pooled, inv_rows, gamma, beta, W, bias, out);
    return hipGetLastError();
}

}  // namespace amuse
