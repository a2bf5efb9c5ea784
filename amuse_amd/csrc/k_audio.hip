// Audio front-end: PretrainedLPDM_v1.process_single_seq (reference
// models/latent_diffusion/infer_ldm.py:180-193) = kaldi fbank -> pad / normalise -> 3 x ASTModel.forward
// (models/audio/audio_main_new.py:174-204: DeiT-B distilled ViT over 2 + 12 x 101 tokens) -> feature_head.
// 260 GFLOP per encoder per clip - 37 x the 1000-step sampler - almost all of it in four GEMM shapes, so this file is
// a conventional MFMA pipeline rather than the register-resident design of the sampler:
//   k_fbank        one workgroup per frame: DC removal, pre-emphasis, Hann window, 512-point FFT in LDS, 128 mel bins
//   k_im2col       16 x 16 stride-10 patches of the [128 x 1024] spectrogram as bf16 rows (K = 256)
//   k_gemm_tm      (k_audio_gemm.hip) C = A . W^T (+ fused epilogue): 256 x 128 x 32 stages by LDS-DMA, fp32 accumulation.
//                  Weights are the A operand, so a lane ends up with 8 consecutive FEATURES of one token row: bias,
//                  GELU, residual and the q / k / v^T split are applied in registers.  Its operands and outputs are
//                  TILE-MAJOR (amuse_audio.hpp): so are the residual stream and every activation in this file.
//   k_ln_bf16      LayerNorm of the fp32 residual stream -> bf16 GEMM operand (one workgroup per 16-row tile row)
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
// then Conv2d(1, 768, 16, stride 10): audio_main_new.py:180-184, 92-96); tile-major (amuse_audio.hpp)
__global__ __launch_bounds__(256) void k_im2col(const float* __restrict__ fbank, bf16raw* __restrict__ patches, int B) {
    const size_t row = (size_t)blockIdx.x;           // b * 1212 + p
    const int b = (int)(row / kAstPatches), p = (int)(row - (size_t)b * kAstPatches);
    const int fh = p / kAstT, tw = p - fh * kAstT;
    const int kh = threadIdx.x >> 4, kw = threadIdx.x & 15;
    const float v = fbank[((size_t)b * kAstFrames + 10 * tw + kw) * kAstMel + 10 * fh + kh];
    typedef __bf16 bf;
    patches[tm_bf16(row, threadIdx.x, 256)] = __builtin_bit_cast(bf16raw, (bf)v);
}

// cls / distillation rows of the token matrix (audio_main_new.py:185-188)
__global__ __launch_bounds__(256) void k_ast_tokens(const float* __restrict__ cls, const float* __restrict__ dist,
                                                    const float* __restrict__ pos, float* __restrict__ X) {
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < 2 * kAstDim; i += 256) {
        const int r = i / kAstDim, c = i - r * kAstDim;
        X[tm_f32((size_t)b * kAstTokens + r, c, kAstDim)] = (r == 0 ? cls[c] : dist[c]) + pos[(size_t)r * kAstDim + c];
    }
}

// ---------------------------------------------------------------------------------------------- LayerNorm rows
// fp32 tile-major in -> bf16 tile-major out.  One workgroup per 16-row tile row; wave w owns the feature tiles 6 w .. 6 w + 5, so
// lane (g, j) holds 48 values of row j (features 32 t + 8 g .. + 7) and every load / store of a wave is one contiguous 1 KiB.
// Row statistics: lane partials -> the four g lanes of a row (two shuffles) -> the four waves through LDS, two passes (mean,
// then the centred second moment), always in the same order - a row's result does not depend on where in the batch it sits.
__global__ __launch_bounds__(256) void k_ln_bf16(const float* __restrict__ X, const float* __restrict__ gamma,
                                                 const float* __restrict__ beta, float eps, bf16raw* __restrict__ out) {
    __shared__ float red[2][4][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, j = lane & 15;
    const size_t tile0 = (size_t)blockIdx.x * (kAstDim / 32) + 6 * wave;
    const float* x = X + tile0 * 512 + lane * 4;
    f32x4 v[6][2];
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        v[t][0] = ld4(x + t * 512);
        v[t][1] = ld4(x + t * 512 + 256);
        s += ((v[t][0][0] + v[t][0][1]) + (v[t][0][2] + v[t][0][3])) + ((v[t][1][0] + v[t][1][1]) + (v[t][1][2] + v[t][1][3]));
    }
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    if (g == 0) red[0][wave][j] = s;
    __syncthreads();
    const float mean = ((red[0][0][j] + red[0][1][j]) + (red[0][2][j] + red[0][3][j])) * (1.0f / kAstDim);
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const float d = v[t][h][m] - mean;
                q += d * d;
            }
    q += __shfl_xor(q, 16);
    q += __shfl_xor(q, 32);
    if (g == 0) red[1][wave][j] = q;
    __syncthreads();
    const float rstd = 1.0f / sqrtf(((red[1][0][j] + red[1][1][j]) + (red[1][2][j] + red[1][3][j])) * (1.0f / kAstDim) + eps);
    bf16raw* o = out + tile0 * 512 + lane * 8;
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        const int f = 32 * (6 * wave + t) + 8 * g;
        uint2 pk[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x4 ga = ld4(gamma + f + 4 * h), be = ld4(beta + f + 4 * h);
            f32x4 y;
#pragma unroll
            for (int m = 0; m < 4; ++m) y[m] = (v[t][h][m] - mean) * rstd * ga[m] + be[m];
            pk[h] = pack4(y);
        }
        *reinterpret_cast<uint4*>(o + t * 512) = uint4{pk[0].x, pk[0].y, pk[1].x, pk[1].y};
    }
}

// ---------------------------------------------------------------------------------------------- row-major <-> tile-major
// one workgroup per 16-row tile row; thread = (tile-relative feature octet walk): rows >= M read as zero / are not written
__global__ __launch_bounds__(256) void k_tile_bf16(const bf16raw* __restrict__ src, bf16raw* __restrict__ dst, int M, int F) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, j = lane & 15;
    const size_t row = (size_t)blockIdx.x * 16 + j;
    for (int ft = wave; ft < F / 32; ft += 4) {
        uint4 u = uint4{0, 0, 0, 0};
        if (row < (size_t)M) u = *reinterpret_cast<const uint4*>(src + row * F + 32 * ft + 8 * g);
        *reinterpret_cast<uint4*>(dst + ((size_t)blockIdx.x * (F / 32) + ft) * 512 + lane * 8) = u;
    }
}
__global__ __launch_bounds__(256) void k_untile_bf16(const bf16raw* __restrict__ src, bf16raw* __restrict__ dst, int M, int F) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, j = lane & 15;
    const size_t row = (size_t)blockIdx.x * 16 + j;
    if (row >= (size_t)M) return;
    for (int ft = wave; ft < F / 32; ft += 4)
        *reinterpret_cast<uint4*>(dst + row * F + 32 * ft + 8 * g) =
            *reinterpret_cast<const uint4*>(src + ((size_t)blockIdx.x * (F / 32) + ft) * 512 + lane * 8);
}
__global__ __launch_bounds__(256) void k_untile_f32(const float* __restrict__ src, float* __restrict__ dst, int M, int F) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, j = lane & 15;
    const size_t row = (size_t)blockIdx.x * 16 + j;
    if (row >= (size_t)M) return;
    for (int ft = wave; ft < F / 32; ft += 4) {
        const float* t = src + ((size_t)blockIdx.x * (F / 32) + ft) * 512 + lane * 4;
        st4(dst + row * F + 32 * ft + 8 * g, ld4(t));
        st4(dst + row * F + 32 * ft + 8 * g + 4, ld4(t + 256));
    }
}

// ---------------------------------------------------------------------------------------------- attention
constexpr int kKc = 64;                 // keys per LDS chunk
constexpr int kKS = 64 + 8;             // padded K row (bf16): 144 B
constexpr int kVS = kKc + 8;            // padded V^T row (bf16): 144 B
// grid (19 query blocks of 64, 12 heads, B); wave w owns the 16 queries 64 qb + 16 w ..
__global__ __launch_bounds__(256) void k_ast_attn(const bf16raw* __restrict__ QK /*[M][1536]*/,
                                                  const bf16raw* __restrict__ Vt /*[B][768][1216]*/,
                                                  bf16raw* __restrict__ O /*tile-major [M][768]*/) {
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
            *reinterpret_cast<uint2*>(O + tm_bf16(row0 + q, 64 * h + 16 * td + 4 * g, kAstDim)) = pack4(o[td] * inv);
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
        const size_t row = (size_t)b * kAstTokens + r;
        f32x4 v[3];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) { v[i] = ld4(X + tm_f32(row, 256 * i + 4 * lane, kAstDim)); s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]); }
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

}  // namespace

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
    hipLaunchKernelGGL(k_ln_bf16, dim3((M + 15) / 16), dim3(256), 0, s, X, gamma, beta, eps, out);
    return hipGetLastError();
}
hipError_t launch_tile_bf16(const unsigned short* src, unsigned short* dst, int M, int F, hipStream_t s) {
    hipLaunchKernelGGL(k_tile_bf16, dim3((M + kGemmTM - 1) / kGemmTM * (kGemmTM / 16)), dim3(256), 0, s, src, dst, M, F);
    return hipGetLastError();
}
hipError_t launch_untile_bf16(const unsigned short* src, unsigned short* dst, int M, int F, hipStream_t s) {
    hipLaunchKernelGGL(k_untile_bf16, dim3((M + 15) / 16), dim3(256), 0, s, src, dst, M, F);
    return hipGetLastError();
}
hipError_t launch_untile_f32(const float* src, float* dst, int M, int F, hipStream_t s) {
    hipLaunchKernelGGL(k_untile_f32, dim3((M + 15) / 16), dim3(256), 0, s, src, dst, M, F);
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
